"""Round-4 vector groups of tests/golden/ref_test_vectors.json: literal expectations of the reference's own unit tests for the emit gate
(filters, raypath colour, the gate's composition), the shape-scalar draw contract, the distribution slot semantics and the 64-bit ray
index — each group run on the ORACLE (oracle/halo_oracle.c) and on the PRODUCT (host builders of libhalo_hip.so through the C ABI; the HIP
kernels in the `-m gpu` half).  The expectations are data transcribed from

  test/unit-correctness/core/test_filter_spec.cpp            EntryExitSpec_Match.*, DirectionSpec.*, FilterSpec_ManyOrClauses,
                                                             FilterSpec_MultiCrystal, FilterSpec_MatchOrbitInvariant.*
  test/unit-correctness/core/test_component_gate.cpp         ComponentGateMatchSummand.*, ComponentGateCollectData.*
  test/unit-correctness/core/test_crosslayer_component.cpp   CrossLayerAccumulation.*
  test/unit-correctness/core/test_collect_data_symmetry_groups.cpp   CollectDataColorGroups.*
  test/unit-correctness/core/test_color_symmetry_oracle.cpp  ColorSymmetryOracle.*
  test/unit-correctness/core/test_crystal_sync_group_sampling.cpp    ShapeScalarSyncGroupSampling.*
  test/unit-correctness/core/test_distribution_slots.cpp     DistributionSlots.*
  test/unit-correctness/core/test_pcg_ray_base_split.cpp     SplitPcgRayBase.*, PcgAdvanceHi.*, PcgSeedWithHigh.*, InRangeParity, CrossU32

(each group's `source` has the lines).  Nothing here reads /root/reference.
"""
import ctypes as C
import itertools
import json
import os

import numpy as np
import pytest

from ice_halo_sim_amd import abi, backend, scenes

from _libs import HoStream, fptr, have_ref, oracle, ref

HERE = os.path.dirname(os.path.abspath(__file__))
V = json.load(open(os.path.join(HERE, "golden", "ref_test_vectors.json")))
T = scenes.filter_term
PRISM_FACES = [1, 2, 3, 4, 5, 6, 7, 8]
PYRAMID_FACES = [1, 2] + list(range(3, 9)) + list(range(13, 19)) + list(range(23, 29))
NONCE_SHAPE = 0x6A09E667


def make_axis(spec):
    """AxisDistribution with the INTERNAL latitude given directly (the reference tests fill the struct, not the JSON form)."""
    a = abi.HaloAxis()
    a.azimuth, a.latitude, a.roll = abi.dist(spec["azimuth"]), abi.dist(spec["latitude"]), abi.dist(spec["roll"])
    return a


def test_axis(roll=0.0):
    """MakeAxis(roll_mean_deg) of the reference's filter tests: azimuth uniform 360, latitude fixed 90, roll fixed."""
    return make_axis({"azimuth": {"type": "uniform", "mean": 0, "std": 360}, "latitude": {"type": "none", "mean": 90, "std": 0},
                      "roll": {"type": "none", "mean": roll, "std": 0}})


test_axis.__test__ = False


def filter_both(f, ax, path, d=(0.0, 0.0, 1.0), cid=0):
    """(product fast tables | None when the path is longer than the kernels' 128-bit register, oracle)"""
    L, O = backend.load_library(), oracle()
    n = len(path)
    p = (C.c_uint8 * max(n, 1))(*path)
    dv = (C.c_float * 3)(*d)
    want = int(O.ho_filter_check(C.byref(f), C.byref(ax), C.cast(p, C.POINTER(C.c_uint8)), n, C.cast(dv, C.POINTER(C.c_float)), cid) != 0)
    got = None
    if n <= 16:
        g = C.c_int32(-1)
        assert L.halo_host_filter_fast_check(C.byref(f), C.byref(ax), p, n, dv, cid, C.byref(g)) == 0
        got = g.value
    return got, want


def color_both(cs, ax, path, d=(0.0, 0.0, 1.0), cid=0, carried=0):
    L, O = backend.load_library(), oracle()
    n = len(path)
    p = (C.c_uint8 * max(n, 1))(*path)
    dv = (C.c_float * 3)(*d)
    want = int(O.ho_color_mask(C.byref(cs), C.byref(ax), C.cast(p, C.POINTER(C.c_uint8)), n, C.cast(dv, C.POINTER(C.c_float)), cid, carried))
    g = C.c_uint64(0)
    assert L.halo_host_color_fast_mask(C.byref(cs), C.byref(ax), p, n, dv, cid, carried, C.byref(g)) == 0
    return int(g.value), want


def summand_term(s):
    if "none" in s:
        return T("none")
    if "raypath" in s:
        return T("raypath", raypath=s["raypath"])
    return T("entry_exit", entry=s.get("entry"), exit=s.get("exit"), min_len=s.get("min_len", 1), max_len=s.get("max_len"))


def raypath_or_filter(clauses, symmetry="", action="filter_in"):
    """MakeRaypathSpec / MakeComplexSpec: one clause = a simple raypath filter, several = an OR of one-term AND clauses"""
    if len(clauses) == 1:
        return scenes.simple_filter(T("raypath", raypath=clauses[0]), symmetry=symmetry, action=action)
    return scenes.complex_filter([[T("raypath", raypath=c)] for c in clauses], symmetry=symmetry, action=action)


# ---------------------------------------------------------------------------------------------------------------------
# test_filter_spec.cpp
# ---------------------------------------------------------------------------------------------------------------------
def test_entry_exit_spec_match():
    n_cases = 0
    for spec in V["entry_exit_spec_match"]["specs"]:
        f = scenes.simple_filter(T("entry_exit", entry=spec["entry"], exit=spec["exit"], min_len=spec["min_len"], max_len=spec["max_len"]),
                                 symmetry=spec["symmetry"])
        ax = test_axis(spec["roll"])
        for c in spec["cases"]:
            got, want = filter_both(f, ax, c["path"])
            assert want == c["expected"], ("oracle", spec["name"], c)
            if got is not None:
                assert got == c["expected"], ("product", spec["name"], c)
            n_cases += 1
    assert n_cases == 39 + 72


def test_direction_spec():
    g = V["direction_spec"]
    for c in g["cases"]:
        f = scenes.simple_filter(T("direction", az=g["lon"], el=g["lat"], radii=g["radii"]), action=c["action"])
        for path in ([], [3, 5]):   # DirectionSpec reads only the direction (StateAgnostic): the recorder may be empty
            got, want = filter_both(f, test_axis(), path, d=tuple(float(x) for x in c["dir"]))
            assert want == c["expected"] and got == c["expected"], (c, path)


def test_many_or_clauses():
    g = V["many_or_clauses"]
    assert len(g["clauses"]) == 30 > 16
    f = raypath_or_filter(g["clauses"], g["symmetry"])
    for c in g["cases"]:
        got, want = filter_both(f, test_axis(), c["path"])
        assert want == c["expected"] and got == c["expected"], c


def test_multi_crystal_canonical():
    g = V["multi_crystal_canonical"]
    prism_spec = raypath_or_filter([g["prism_seed"]], g["symmetry"])
    pyr_spec = raypath_or_filter([g["pyramid_seed"]], g["symmetry"])
    assert len(g["prism_orbit"]) == 6 and len(g["pyramid_orbit"]) == 6
    for rp in g["prism_orbit"]:
        assert filter_both(prism_spec, test_axis(), rp) == (1, 1), rp
        assert filter_both(pyr_spec, test_axis(), rp) == (0, 0), rp      # no canonical leak between crystals
    for rp in g["pyramid_orbit"]:
        assert filter_both(pyr_spec, test_axis(), rp) == (1, 1), rp
        assert filter_both(prism_spec, test_axis(), rp) == (0, 0), rp


def test_match_orbit_invariant():
    O = oracle()
    members = 0
    for c in V["match_orbit_invariant"]["cases"]:
        ax = test_axis(c["roll"])
        assert O.ho_compute_sigma_a(c["roll"]) == c["sigma_a"]          # kSigmaARollDeg is the inverse of ComputeSigmaA
        f = raypath_or_filter(c["clauses"], c["symmetry"])
        for seed in c["clauses"]:
            assert seed in c["members"]
        for m in c["members"]:
            assert filter_both(f, ax, m) == (1, 1), (c["name"], m)
            members += 1
    assert members > 300


# ---------------------------------------------------------------------------------------------------------------------
# test_component_gate.cpp / test_crosslayer_component.cpp / test_collect_data_symmetry_groups.cpp: host level
# ---------------------------------------------------------------------------------------------------------------------
def color_set_from(summands, bits=None):
    return scenes.color_set([(summand_term(s), s.get("symmetry", ""), (bits[k] if bits else s.get("bit", k))) for k, s in enumerate(summands)])


def test_summand_mask():
    for g in V["summand_mask"]["groups"]:
        cs = color_set_from(g["summands"], bits=list(range(len(g["summands"]))))
        for c in g["cases"]:
            got, want = color_both(cs, test_axis(), c["path"])
            assert want == c["mask"], ("oracle", g["name"], c)
            assert got == c["mask"], ("product", g["name"], c)
            if "pass" in c:   # CheckSummandMask: the gate applies the action, the mask is pre-action
                f = scenes.complex_filter([[summand_term(s)] for s in g["summands"]], action=g["action"])
                assert filter_both(f, test_axis(), c["path"]) == (c["pass"], c["pass"]), (g["name"], c)


def gate_on_host(case, which):
    """The gate's composition from the two host predicates: physical filter first (fail = terminated, the colour pass does not
    run), then the colour pass on the carried mask, then the prob roll (0 emits, 1 continues) — simulator.cpp:665-742."""
    ax = test_axis()
    if case["physical"] is not None:
        f = scenes.complex_filter([[T("raypath", raypath=rp)] for rp in case["physical"]])
        if not filter_both(f, ax, case["path"])[which]:
            return "terminated", case["carried"]
    mask = case["carried"]
    if case["color"]:
        mask = color_both(color_set_from(case["color"]), ax, case["path"], carried=case["carried"])[which]
    return ("continue" if case["prob"] >= 1.0 else "emit"), mask


def test_collect_data_gate_on_the_host_predicates():
    for case in V["collect_data_gate"]["cases"]:
        for which, name in ((0, "product"), (1, "oracle")):
            assert gate_on_host(case, which) == (case["outcome"], case["mask"]), (name, case["name"])


def test_color_symmetry_oracle():
    g = V["color_symmetry_oracle"]
    n = 0
    for c in g["cases"]:
        ax = make_axis(c["axis"])
        faces = PRISM_FACES if c["crystal"] == "prism" else PYRAMID_FACES
        phys = raypath_or_filter([c["seed"]], c["symmetry"])
        cs = scenes.color_set([(T("raypath", raypath=c["seed"]), c["symmetry"], 0)])
        assert filter_both(phys, ax, c["seed"]) == (1, 1) and color_both(cs, ax, c["seed"]) == (1, 1)   # the seed matches itself, both ways
        hits = 0
        for path in [list(p) for p in itertools.product(faces, repeat=2)] + [g["outsider"]]:
            pg, pw = filter_both(phys, ax, path)
            cg, cw = color_both(cs, ax, path)
            assert pg == pw == cg == cw, (c, path, pg, pw, cg, cw)    # identical membership, line by line
            hits += pw
            n += 1
        assert 1 <= hits <= 24
        assert filter_both(phys, ax, g["outsider"]) == (0, 0)
    assert n > 5000


# the same composition where it is made: the emit gate of a TRACE (oracle here, the HIP kernels in the gpu half) ------
def run_capture(b, scene, n, sets=None, classes=None, filters=None, rd=None):
    from tests._oracle_backend import run_session
    rd = rd or scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, 256, 128, visible=abi.VISIBLE_FULL)
    if filters is not None:
        b.set_filters(filters)
    if sets is not None:
        b.set_color(sets, classes)
    stats = run_session(b, scene, rd, scenes.wl_discrete(550.0), n)
    return stats, b.DrainExits()


def random_prism_entry(filter_id=0, color_id=0, cid=1):
    full = {"type": "uniform", "mean": 0.0, "std": 360.0}
    return scenes.entry(scenes.prism_crystal(1.0), scenes.axis(zenith={"type": "uniform", "mean": 90, "std": 360}, azimuth=full, roll=full), 1.0, cid,
                        filter_id=filter_id, color_id=color_id)


def paths_of(ex):
    return [tuple(int(x) for x in e["path"][: e["path_len"]]) for e in ex]


def expected_mask(path, color):
    m = 0
    for t in color:
        if "none" in t or tuple(t["raypath"]) == tuple(path):
            m |= 1 << t["bit"]
    return m


def check_gate_composition(make_backend):
    G = V["collect_data_gate"]
    by = {c["name"]: c for c in G["cases"]}
    n = 40_000
    any_class = [scenes.color_class(range(0, 12), "any")]
    # 1. colour bits are exactly the matching predicates' (EmitRay..., ColorMiss..., ColorGroups.*, MatchAll...)
    for name in ("EmitRayGetsMappedBitForMatchedColorSummand", "ColorMissDoesNotAffectPhysicalSurvival", "ColorGroups.MultipleGroupsEachContributeTheirOwnBits",
                 "ColorGroups.NonMatchingGroupContributesNothing", "MatchAllColorSpecTagsEveryRayRegardlessOfPath[3,5]"):
        color = by[name]["color"]
        b = make_backend()
        _, ex = run_capture(b, scenes.scene([(0.0, [random_prism_entry(color_id=1)])], max_hits=5), n, [color_set_from(color)], any_class)
        b.close()
        want = np.array([expected_mask(p, color) for p in paths_of(ex)], np.uint64)
        assert len(ex) > 3 * n and (ex["color_mask"] == want).all(), name
        seen = set(int(m) for m in np.unique(ex["color_mask"]))
        assert by[name]["mask"] in seen, (name, seen)                      # the reference's literal case occurs in the render
        if "none" in color[0]:
            assert seen == {1 << 11} and (ex["path_len"] == 1).any()      # whole-crystal bit on every exit, the entry-face reflection included
    # 2. a physical filter that rejects an exit ends it before the colour pass: PhysicalFailTerminatesRayAndProducesNoBits
    c = by["PhysicalFailTerminatesRayAndProducesNoBits"]
    phys = scenes.complex_filter([[T("raypath", raypath=rp)] for rp in c["physical"]])
    b = make_backend()
    _, ex = run_capture(b, scenes.scene([(0.0, [random_prism_entry(filter_id=1, color_id=1)])], max_hits=5), n, [color_set_from(c["color"])], any_class, [phys])
    b.close()
    assert len(ex) > 100 and set(paths_of(ex)) == {(3, 5)} and not ex["color_mask"].any()
    # 3. two layers: CrossLayer.OrsBothLayerBits / NonMatchingSecondLayerAddsNoBit — the carried layer-0 bit survives the pool and
    #    the shuffle, the final mask is exactly one layer-0 bit | the layer-1 bit of the exit's own path
    l0, l1 = by["CrossLayer.OrsBothLayerBits[layer 0]"], by["CrossLayer.OrsBothLayerBits[layer 1]"]
    f0 = scenes.complex_filter([[T("raypath", raypath=rp)] for rp in l0["physical"]])
    f1 = scenes.complex_filter([[T("raypath", raypath=rp)] for rp in l1["physical"]])
    sc = scenes.scene([(1.0, [random_prism_entry(filter_id=1, color_id=1)]), (0.0, [random_prism_entry(filter_id=2, color_id=2, cid=2)])], max_hits=5)
    b = make_backend()
    stats, ex = run_capture(b, sc, 4 * n, [color_set_from(l0["color"]), color_set_from(l1["color"])], any_class, [f0, f1])
    b.close()
    assert stats[0].exit_count == 0 and stats[0].continuation_count > 1000          # prob 1: everything that passes the filter continues
    assert len(ex) > 50 and (ex["layer"] == 1).all()
    for p, m in zip(paths_of(ex), ex["color_mask"]):
        assert p in ((2, 4), (4, 6))
        assert int(m) & 0b1100 == (4 if p == (2, 4) else 8)
        assert int(m) & 0b0011 in (1, 2)
    assert l1["mask"] in set(int(m) for m in ex["color_mask"])                       # {3,5}@L0 then {2,4}@L1 = 0b0101 occurs
    # 4. the prob roll and the physical filter run once per candidate whatever the colour configuration
    gi = G["gate_independent_of_color_groups"]
    results = []
    for cnt in gi["group_counts"]:
        b = make_backend()
        sets = [color_set_from([gi["color_term"]] * cnt)] if cnt else None
        sc = scenes.scene([(gi["prob"], [random_prism_entry(color_id=1 if cnt else 0)]), (0.0, [random_prism_entry(cid=2)])], max_hits=5)
        stats, ex = run_capture(b, sc, n, sets, any_class if cnt else None)
        b.close()
        e0 = ex[ex["layer"] == 0]
        results.append((int(stats[0].exit_count), int(stats[0].continuation_count), np.sort((e0["root"].astype(np.uint64) << np.uint64(16)) | e0["seq"].astype(np.uint64))))
    assert 0.3 < results[0][1] / (results[0][0] + results[0][1]) < 0.7
    for r in results[1:]:
        assert r[0] == results[0][0] and r[1] == results[0][1] and np.array_equal(r[2], results[0][2])


def test_collect_data_gate_in_the_oracle_trace():
    from tests._oracle_backend import OracleBackend
    check_gate_composition(lambda: OracleBackend(seed=42, capture_exits=1, threads=8))


@pytest.mark.gpu
def test_collect_data_gate_in_the_hip_kernels():
    from ice_halo_sim_amd.backend import HipTraceBackend
    check_gate_composition(lambda: HipTraceBackend(device=0, seed=42, capture_exits=1))


# ---------------------------------------------------------------------------------------------------------------------
# test_crystal_sync_group_sampling.cpp: the draw contract, replayed by hand on the primitive stream
# ---------------------------------------------------------------------------------------------------------------------
def crystal_from(c):
    if c["kind"] == "prism":
        return scenes.prism_crystal(c["height"][0], c["face_dist"], c["sync_group"])
    cr = scenes.pyramid_crystal(c["height"][0], c["height"][1], c["height"][2], face_distance=c["face_dist"], sync_group=c["sync_group"])
    return cr


SLOT = {"h0": 0, "h1": 1, "h2": 2, "d0": 3, "d1": 4, "d2": 5, "d3": 6, "d4": 7, "d5": 8}


def product_scalars(cr, seed, idx, via_plan):
    out = np.zeros(9, np.float32)
    assert backend.load_library().halo_host_shape_scalars(C.byref(cr), seed, idx, via_plan, fptr(out)) == 0
    out[:3] = np.abs(out[:3])      # heights fold at their use site (BuildPrismShape / BuildPyramidDispatch take fabsf)
    return out


def oracle_scalars(cr, seed, idx):
    out = np.zeros(9, np.float32)
    oracle().ho_shape_scalars(C.byref(cr), seed, idx, fptr(out))
    return out


def contract_replay(cr, seed, idx, order=None):
    """the contract stream by hand: heights (upper, prism, lower) then d[0..5], on ONE primitive stream; `order` names the slots that
    draw when sync groups are declared (the others reuse)"""
    O = oracle()
    s = HoStream((seed ^ NONCE_SHAPE) & 0xFFFFFFFF, idx & 0xFFFFFFFF, 0)
    nh = 1 if cr.kind == abi.CRYSTAL_PRISM else 3
    slots = order if order is not None else [["h0", "h1", "h2"][i] for i in range(nh)] + ["d%d" % i for i in range(6)]
    out = {}
    for name in slots:
        q = SLOT[name]
        d = cr.height[q] if q < 3 else cr.face_dist[q - 3]
        v = O.ho_pcg_get_dist(C.byref(s), d.type, d.center, d.spread)
        out[name] = abs(v) if q < 3 else v
    return out


def test_sync_group_sampling_contract():
    g = V["sync_group_sampling"]
    for c in g["cases"]:
        cr = crystal_from(c)
        seed = c["seed"]
        for idx in range(g["replay_draws"]):
            po, pp, oo = product_scalars(cr, seed, idx, 0), product_scalars(cr, seed, idx, 1), oracle_scalars(cr, seed, idx)
            assert np.array_equal(po.view(np.uint32), pp.view(np.uint32)), (c["name"], idx)       # serial walk == draw plan
            assert np.allclose(po, oo, rtol=3e-7, atol=0), (c["name"], idx, po, oo)                # libm vs hipcc host libm: last bit
            if not any(c["sync_group"]):
                want = contract_replay(cr, seed, idx)
                for name, v in want.items():
                    assert oo[SLOT[name]] == np.float32(v), (c["name"], idx, name)
                    assert po[SLOT[name]] == pytest.approx(v, rel=3e-7), (c["name"], idx, name)
            if "draw_order" in c:
                want = contract_replay(cr, seed, idx, c["draw_order"])
                for name, v in want.items():
                    assert oo[SLOT[name]] == np.float32(v) and po[SLOT[name]] == pytest.approx(v, rel=3e-7), (c["name"], idx, name)
            for arr in (po, oo):
                for a, b in c.get("equal", []):
                    assert arr[SLOT[a]] == arr[SLOT[b]], (c["name"], idx, a, b)                    # bit for bit: one draw, one value
                for a, b in c.get("not_equal", []):
                    assert arr[SLOT[a]] != arr[SLOT[b]], (c["name"], idx, a, b)
                for a in c.get("negative", []):
                    assert arr[SLOT[a]] < 0.0, (c["name"], idx, a)
                for a, b in c.get("abs_equal", []):
                    assert arr[SLOT[a]] == abs(arr[SLOT[b]]), (c["name"], idx, a, b)


# ---------------------------------------------------------------------------------------------------------------------
# test_distribution_slots.cpp
# ---------------------------------------------------------------------------------------------------------------------
DIST_TYPE = {"none": abi.DIST_NONE, "uniform": abi.DIST_UNIFORM, "gauss": abi.DIST_GAUSS, "zigzag": abi.DIST_ZIGZAG, "laplacian": abi.DIST_LAPLACIAN,
             "gauss_legacy": abi.DIST_GAUSS_LEGACY}


def expected_draw(O, s, c):
    """the documented formula per type over the primitive draws (ExpectedDraw, test_distribution_slots.cpp:84-107), in float"""
    f = np.float32
    a, sp = f(c["anchor"]), f(c["spread"])
    if c["type"] == "none":
        return a
    if c["type"] == "uniform":
        return f((f(O.ho_pcg_uniform(C.byref(s))) - f(0.5)) * sp + a)
    if c["type"] in ("gauss", "gauss_legacy"):
        return f(f(O.ho_pcg_gaussian(C.byref(s))) * sp + a)
    if c["type"] == "zigzag":
        u = f(O.ho_pcg_uniform(C.byref(s)))
        return f(abs(sp * f(np.sin(f(u * f(2.0) * f(np.pi)))) + a))
    u = f(O.ho_pcg_uniform(C.byref(s)))
    sign = f(-1.0) if u < f(0.5) else f(1.0)
    arg = max(f(1.0) - f(2.0) * abs(u - f(0.5)), np.finfo(np.float32).tiny)
    return f(a - sp * sign * f(np.log(arg)))


def ulps(a, b):
    ia, ib = np.float32(a).view(np.int32), np.float32(b).view(np.int32)
    return abs(int(ia) - int(ib))


def test_distribution_slots():
    g = V["distribution_slots"]
    O = oracle()
    seed = g["seed"]
    # the formula, on the oracle's sampler (and on the reference's own pcg_get_dist where its headers are compiled) ...
    for c in g["formula_cases"]:
        s, m = HoStream(seed, 0, 0), HoStream(seed, 0, 0)
        st = np.array([seed, 0, 0], np.uint32)
        for i in range(g["draws"]):
            got = O.ho_pcg_get_dist(C.byref(s), DIST_TYPE[c["type"]], c["anchor"], c["spread"])
            want = expected_draw(O, m, c)
            assert ulps(got, want) <= 4, (c, i, got, want)
            assert (s.slot, s.global_idx) == (m.slot, m.global_idx)           # same number of primitive draws consumed
            if have_ref():
                r = ref().ref_pcg_get_dist(st.ctypes.data_as(C.POINTER(C.c_uint32)), DIST_TYPE[c["type"]], c["anchor"], c["spread"])
                assert np.float32(r).view(np.uint32) == np.float32(got).view(np.uint32)
    # ... and on the product's host sampler: the shape-scalar draw of a prism's height IS Draw(dist) on slot 0 of the shape stream
    for c in g["formula_cases"]:
        cr = scenes.prism_crystal({"type": c["type"], "mean": c["anchor"], "std": c["spread"]} if c["type"] != "none" else c["anchor"])
        for idx in range(g["draws"]):
            m = HoStream((seed ^ NONCE_SHAPE) & 0xFFFFFFFF, idx, 0)
            raw = np.zeros(9, np.float32)
            assert backend.load_library().halo_host_shape_scalars(C.byref(cr), seed, idx, 0, fptr(raw)) == 0
            assert ulps(raw[0], expected_draw(O, m, c)) <= 4, (c, idx)
    # Gaussian and legacy Gaussian share one branch
    a, b = HoStream(seed, 0, 0), HoStream(seed, 0, 0)
    for i in range(g["draws"]):
        x = O.ho_pcg_get_dist(C.byref(a), abi.DIST_GAUSS, 30.0, 5.0)
        y = O.ho_pcg_get_dist(C.byref(b), abi.DIST_GAUSS_LEGACY, 30.0, 5.0)
        assert np.float32(x).view(np.uint32) == np.float32(y).view(np.uint32)
    # anchor and spread are not swapped: ranges that do not reference the arithmetic
    for c in g["range_cases"]:
        s = HoStream(seed, 0, 0)
        cr = scenes.prism_crystal({"type": c["type"], "mean": c["anchor"], "std": c["spread"]} if c["type"] != "none" else c["anchor"])
        for i in range(g["range_draws"]):
            v = O.ho_pcg_get_dist(C.byref(s), DIST_TYPE[c["type"]], c["anchor"], c["spread"])
            raw = np.zeros(9, np.float32)
            assert backend.load_library().halo_host_shape_scalars(C.byref(cr), seed, i, 0, fptr(raw)) == 0
            for x in (v, float(raw[0])):
                assert c["lo"] - 1e-4 * abs(c["lo"]) <= x <= c["hi"] + 1e-4 * abs(c["hi"]), (c, i, x)


# ---------------------------------------------------------------------------------------------------------------------
# test_pcg_ray_base_split.cpp
# ---------------------------------------------------------------------------------------------------------------------
def pcg_libs():
    out = [("oracle", oracle().ho_pcg_hash, oracle().ho_pcg_advance_hi, oracle().ho_pcg_seed_with_high)]
    if have_ref():
        out.append(("ref", ref().ref_pcg_hash, ref().ref_pcg_advance_hi, ref().ref_pcg_seed_with_high))
    return out


def test_pcg_ray_base_split_primitives():
    g = V["pcg_ray_base_split"]
    M = 0xFFFFFFFF
    for c in g["split"]:   # SplitPcgRayBase is the lo / hi cut of the 64-bit index; the backends' counters are uint64 and cut the same way
        assert (c["ray_base"] & M, c["ray_base"] >> 32) == (c["lo"], c["hi"]) and ((c["hi"] << 32) | c["lo"]) == c["ray_base"]
    for name, h, adv, swh in pcg_libs():
        for c in g["advance_hi"]:
            assert adv(c["base_lo"], c["base_hi"], c["tid"]) == c["expected"], (name, c)
        for seed in g["seed_with_high_identity_seeds"]:
            assert swh(seed, 0) == seed, (name, seed)
        dv = g["seed_with_high_diverges"]
        mixed = {swh(dv["seed"], hi) for hi in dv["hi"]}
        assert len(mixed) == len(dv["hi"]) and dv["seed"] not in mixed, name

        def pre(seed, gidx, slot):
            return h(seed ^ h((gidx * 1000003 + slot) & M))

        def post(seed, lo, hi, tid, slot):
            return h(swh(seed, adv(lo, hi, tid)) ^ h((((lo + tid) & M) * 1000003 + slot) & M))

        gr = g["in_range_grid"]
        for seed, base, tid, slot in itertools.product(gr["seeds"], gr["bases"], gr["tids"], gr["slots"]):
            assert pre(seed, base + tid, slot) == post(seed, base, 0, tid, slot), (name, seed, base, tid, slot)
        x = g["cross_u32"]
        draws = {post(x["seed"], x["base_lo"], x["base_hi"], tid, x["slot"]) for tid in range(x["tids_span"])}
        assert len(draws) == x["tids_span"], name
        a = post(x["seed"], 0, x["base_hi"], x["target_lo"], x["slot"])
        lo_b, tid_b = (M - x["target_lo"] + 1) & M, 2 * x["target_lo"]
        assert (lo_b + tid_b) & M == x["target_lo"] and adv(lo_b, x["base_hi"], tid_b) == x["base_hi"] + 1
        assert a != post(x["seed"], lo_b, x["base_hi"], tid_b, x["slot"]), name


def rays_at(make_backend, base, n):
    from tests._oracle_backend import run_session
    b = make_backend()
    b.set_option("ray_base", base)
    run_session(b, scenes.config2_scene(), scenes.config2_render(256, 144), scenes.wl_discrete(550.0), n)
    ex = b.DrainExits()
    b.close()
    order = np.lexsort((ex["seq"], ex["root"]))
    return ex[order]


def check_wrap_in_a_session(make_backend):
    """a session whose ray index crosses 2^32 (option ray_base = SplitPcgRayBase's input): ray i of the session is the ray a session
    started AT base + i traces first (the carry into the high word happens per ray), and a ray of epoch hi = 1 is not the ray with
    the same low word in epoch hi = 0 (CrossU32.WrapDoesNotCollapseStreams on the trace path)."""
    x = V["pcg_ray_base_split"]["cross_u32"]
    base = (1 << 32) - 6
    ex = rays_at(make_backend, base, x["tids_span"])
    assert set(ex["root"]) == set(range(x["tids_span"]))
    for i in (0, 5, 6, 11):
        one = rays_at(make_backend, base + i, 1)
        mine = ex[ex["root"] == i]
        assert len(one) == len(mine) and np.array_equal(one["seq"], mine["seq"])
        assert np.allclose(one["dir"], mine["dir"], atol=1e-6) and np.allclose(one["weight"], mine["weight"], rtol=1e-5)
    lo0 = rays_at(make_backend, 0, 6)                        # indices 0..5 of epoch 0
    wrapped = ex[ex["root"] >= 6]                            # indices 2^32 + 0..5: same low words, epoch 1
    for k in range(6):
        a, b = lo0[lo0["root"] == k], wrapped[wrapped["root"] == 6 + k]
        assert len(a) != len(b) or not np.allclose(a["dir"], b["dir"], atol=1e-4), k
    return ex


def test_ray_index_wraps_in_the_oracle_trace():
    from tests._oracle_backend import OracleBackend
    check_wrap_in_a_session(lambda: OracleBackend(seed=42, capture_exits=1, threads=1))


@pytest.mark.gpu
def test_ray_index_wraps_in_the_hip_kernels():
    from ice_halo_sim_amd.backend import HipTraceBackend
    from tests._oracle_backend import OracleBackend
    eh = check_wrap_in_a_session(lambda: HipTraceBackend(device=0, seed=42, capture_exits=1))
    eo = check_wrap_in_a_session(lambda: OracleBackend(seed=42, capture_exits=1, threads=1))
    assert len(eh) == len(eo) and np.array_equal(eh["root"], eo["root"]) and np.array_equal(eh["seq"], eo["seq"])
    assert np.abs(eh["dir"] - eo["dir"]).max() <= 2e-5 and np.allclose(eh["weight"], eo["weight"], rtol=2e-4, atol=1e-7)
