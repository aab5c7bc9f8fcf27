"""GPU parity of the PRODUCTION filter / raypath-colour kernels (capture OFF) against the oracle.

test_gpu_parity.py compares filtered scenes per ray, which needs exit capture — the MODE=2 kernels, whose filter is the generic
device-side symmetry reduction.  A filtered render a user runs takes other code: `halo_trace_kernel<kModeFilter, ...>` (mode bit
1) or `<kModeColor, ...>` (bit 4) — path in a 128-bit register, the predicates as host-built member tables (FastTables), exit
queue, hit log, regular-prism search — or, for max_hits > 16, the generic kernels `<kModeGeneric, ...>` (bit 3).  Here nothing
is captured: each test asserts through halo_last_route WHICH instantiation ran and compares what a render delivers — image,
landed weight, per-channel sums, exit count, class lanes — with the oracle's (reference semantics: DeviceFilterCheck,
src/core/shared/filter_shared.h:308; emit gate src/core/simulator.cpp:665-762; colour bits / lanes cuda_trace_backend.cu:498-556).

Tolerances (stated).  One scattering layer: both sides trace the SAME rays (shared counter-based streams): landed weight rel
2e-4, exit count rel 3e-4, 8x8 block-mean image rel L2 <= 3e-3, per-channel sums rel 3e-4 (abs floor for near-empty channels).
The oracle runs with its option acc64 (hits summed in double, per-thread pixel caches): the reference's one-float-image accumulator is
itself 1e-3-inexact at these session sizes (oracle/halo_oracle.c, HoBackend::acc64), which would be the largest term of the comparison.
More layers: continuation order is nondeterministic on a GPU, so the comparison is the reference battery's (4x4... here 16x16
block-mean Pearson, sum-Y, landed) against the oracle's own cross-seed floor: HIP within 4 floors (+ a small absolute term).
"""
import os

import numpy as np
import pytest

from ice_halo_sim_amd import abi, config, scenes
from tests._oracle_backend import OracleBackend, run_session
from tests.test_gpu_parity import _E2E_DOCS, _filter_table, _with, block_mean, hip_backend, rel_l2

pytestmark = pytest.mark.gpu

THREADS = max(8, min(os.cpu_count() or 8, 128))


def _render(backend, sc, rd, wl, n, filters, colors=None, geom_clock=None):
    if geom_clock:
        backend.set_option("geom_clock", geom_clock)
    backend.set_filters(filters)
    if colors:
        backend.set_color(*colors)
    st = run_session(backend, sc, rd, wl, n)
    route = backend.last_route() if hasattr(backend, "last_route") else None
    img, landed = backend.ReadbackXyzAccum()
    lanes = backend.ReadbackClassLanes() if colors else None
    backend.close()
    return dict(st=st, route=route, img=img, landed=landed, lanes=lanes)


def _single_layer_checks(h, o, tag, landed_rel=2e-4):
    assert h["st"][0].exit_count == pytest.approx(o["st"][0].exit_count, rel=3e-4, abs=20), tag
    assert abs(h["landed"] - o["landed"]) <= landed_rel * max(o["landed"], 1.0), (tag, h["landed"], o["landed"])
    err = None
    if o["img"].sum() > 0:
        err = rel_l2(block_mean(h["img"]), block_mean(o["img"]))
        assert err <= 3e-3, (tag, err)
        tot = float(o["img"].sum(dtype=np.float64))
        for ch in range(3):
            assert h["img"][..., ch].sum(dtype=np.float64) == pytest.approx(o["img"][..., ch].sum(dtype=np.float64), rel=3e-4, abs=1e-5 * tot), (tag, ch)
    return err


CASES = ["raypath_P", "entry_exit_PBD_d_applicable", "direction_out", "complex", "multi_scatter_gate", "pyramid_PB", "long_paths"]


@pytest.mark.parametrize("case", CASES)
def test_filter_production_kernels_vs_oracle(case):
    """The seven scenes of test_emit_gate_filter_parity, capture off, 6 Mi rays (so that the launches of the larger entries are
    >= 2 Mi rays and take the hit log, the smaller ones direct atomics): kModeFilter for max_hits 7, kModeGeneric for the scene
    whose paths outgrow the 16-face register (max_hits 24)."""
    col = scenes.column_crystal_entry()
    plate = scenes.entry(scenes.prism_crystal(0.3), scenes.axis(zenith={"type": "gauss", "mean": 0, "std": 0.8}), 5.0, 6)
    parry = scenes.entry(scenes.prism_crystal(1.5), scenes.axis(zenith={"type": "gauss", "mean": 90, "std": 0.5}, roll=30.0), 4.0, 2)
    pyr = scenes.entry(scenes.pyramid_crystal(0.3, 1.0, 0.3, upper_wedge=28.0, lower_wedge=28.0),
                       scenes.axis(zenith={"type": "uniform", "mean": 90, "std": 360}, azimuth={"type": "uniform", "mean": 0, "std": 360}), 3.0, 5)
    rd = scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, 512, 256, visible=abi.VISIBLE_FULL)
    layers = {
        "raypath_P": [(0.0, [_with(col, 1), _with(plate, 0)])],
        "entry_exit_PBD_d_applicable": [(0.0, [_with(parry, 2)])],
        "direction_out": [(0.0, [_with(col, 3)])],
        "complex": [(0.0, [_with(col, 4), _with(plate, 4)])],
        "multi_scatter_gate": [(0.6, [_with(plate, 2)]), (0.0, [_with(col, 1)])],
        "pyramid_PB": [(0.0, [_with(pyr, 6), _with(col, 5)])],
        "long_paths": [(0.0, [_with(col, 7), _with(plate, 7)])],
    }[case]
    long_paths = case == "long_paths"
    sc = scenes.scene(layers, max_hits=24 if long_paths else 7)
    n = (4 << 20) if long_paths else (6 << 20)
    wl = scenes.wl_discrete(550.0)
    h = _render(hip_backend(seed=21), sc, rd, wl, n, _filter_table())
    o = _render(OracleBackend(seed=21, threads=THREADS, acc64=1), sc, rd, wl, n, _filter_table())
    r = h["route"]
    # the filtered entries on the fast kernels (the generic ones when paths outgrow the register); entries without a filter run the plain kernels
    assert r.mode_mask & ~abi.MODE_PLAIN == (abi.MODE_GENERIC if long_paths else abi.MODE_FILTER), (case, r.mode_mask)
    assert not (r.mode_mask & abi.MODE_CAPTURE)
    if not long_paths:
        assert r.accum_mask & abi.ACCUM_LOG, r.accum_mask          # the production-shaped filter kernels take the hit log
    if case in ("raypath_P", "direction_out", "complex"):
        assert r.geom_mask & (1 << 3), r.geom_mask                 # ... and the regular-prism search
    if case == "multi_scatter_gate":
        o2 = _render(OracleBackend(seed=7, threads=THREADS, acc64=1), sc, rd, wl, n, _filter_table())

        def within(x, a, b, abs_floor):
            return abs(x - a) <= 4.0 * abs(a - b) + abs_floor
        a, b = o["st"][0].continuation_count, o2["st"][0].continuation_count
        assert within(h["st"][0].continuation_count, a, b, 2e-3 * a + 50)
        assert within(h["landed"], o["landed"], o2["landed"], 5e-3 * o["landed"] + 1.0), (h["landed"], o["landed"], o2["landed"])
        assert within(h["st"][1].exit_count, o["st"][1].exit_count, o2["st"][1].exit_count, 5e-3 * o["st"][1].exit_count + 50)
        pear = lambda x, y: float(np.corrcoef(block_mean(x, 16)[..., 1].ravel(), block_mean(y, 16)[..., 1].ravel())[0, 1])
        floor = pear(o["img"], o2["img"])
        assert pear(h["img"], o["img"]) >= floor - 0.02, (pear(h["img"], o["img"]), floor)
        print("%s: mode %d geom %d accum %d; Pearson %.5f (oracle cross-seed %.5f)" % (case, r.mode_mask, r.geom_mask, r.accum_mask, pear(h["img"], o["img"]), floor))
        return
    assert 0 < o["st"][0].exit_count < (8 if case == "direction_out" else 4.5) * n   # the filter really removed exits
    # paths of 12..24 faces: rounding differences grow with depth until a few rays in a thousand take another face (the per-ray tests
    # ask for 99.8 % identical exits), and this scene's landed weight is the sum of nothing but such exits
    err = _single_layer_checks(h, o, case, landed_rel=1e-3 if long_paths else 2e-4)
    print("%s: mode %d geom %d accum %d; exits %d; block-mean rel L2 %s" % (case, r.mode_mask, r.geom_mask, r.accum_mask, h["st"][0].exit_count, err))


@pytest.mark.parametrize("wl_kind", ["d65_xyz", "d65_planes_log64", "d65_planes_log31", "d65_planes_binned", "d65_planes_direct"])
def test_filter_production_kernels_illuminant_sessions(wl_kind):
    """The complex filter on an illuminant session, the plane layouts a filtered dispatch can meet on a small image: X/Y/Z planes
    (sessions under 8 Mi rays: `<kModeFilter, hex, MONO=false, kAccDirect>`); one scalar plane per pool entry (>= 8 Mi rays) with the hit
    log over the planes (the backend's choice on the reference's 512x256 scenes: `<kModeFilter, hex, true, kAccLogFinal>`, 64 x 8 or 31 x 8
    interleaved tiles), with binned accumulation (option bin = 1, 64 planes = 512 tiles: `<kModeFilter, one, true, kAccBin>`), and with
    direct atomics (option hit_log = 0; a pool of 31 is 248 tiles, not a power of two, so the binned route does not apply)."""
    col = scenes.column_crystal_entry()
    sc = scenes.scene([(0.0, [_with(col, 4)])], max_hits=7)
    rd = scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, 512, 256, visible=abi.VISIBLE_FULL)
    pool = 64 if wl_kind in ("d65_planes_binned", "d65_planes_log64") else 31
    wl = scenes.wl_illuminant("D65", pool)
    n = (5 << 20) if wl_kind == "d65_xyz" else (9 << 20)
    opts = {"d65_planes_binned": {"bin": 1}, "d65_planes_direct": {"hit_log": 0}}.get(wl_kind, {})
    h = _render(hip_backend(seed=33, **opts), sc, rd, wl, n, _filter_table())
    o = _render(OracleBackend(seed=33, threads=THREADS, acc64=1), sc, rd, wl, n, _filter_table())
    r = h["route"]
    assert r.mode_mask == abi.MODE_FILTER, r.mode_mask
    assert r.plane_cnt == (3 if wl_kind == "d65_xyz" else pool)
    assert r.accum_mask == {"d65_xyz": abi.ACCUM_XYZ, "d65_planes_binned": abi.ACCUM_BIN1, "d65_planes_direct": abi.ACCUM_SCALAR}.get(wl_kind, abi.ACCUM_LOG), r.accum_mask
    if wl_kind.startswith("d65_planes_log"):
        assert r.geom_mask == 1 << 3 and r.spec_mask & abi.SPEC_LAST
    err = _single_layer_checks(h, o, wl_kind)
    print("%s: accum %d geom %d, block-mean rel L2 %s" % (wl_kind, r.accum_mask, r.geom_mask, err))


def test_fast_and_generic_filter_kernels_render_the_same_image():
    """Route equivalence (HIP vs HIP, not parity): option filter_fast = 0 sends the same filtered session through the generic
    kernels (device-side reduction, direct atomics).  Same rays, same survivors: equal exit counts, landed weight to 1e-6, image
    to float summation order."""
    col = scenes.column_crystal_entry()
    plate = scenes.entry(scenes.prism_crystal(0.3), scenes.axis(zenith={"type": "gauss", "mean": 0, "std": 0.8}), 5.0, 6)
    sc = scenes.scene([(0.0, [_with(col, 4), _with(plate, 2)])], max_hits=7)
    rd = scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, 512, 256, visible=abi.VISIBLE_FULL)
    out = []
    for fast in (1, 0):
        out.append(_render(hip_backend(seed=3, filter_fast=fast), sc, rd, scenes.wl_discrete(550.0), 5 << 20, _filter_table()))
    a, b = out
    assert a["route"].mode_mask == abi.MODE_FILTER and b["route"].mode_mask == abi.MODE_GENERIC
    assert a["st"][0].exit_count == b["st"][0].exit_count > 0
    assert a["landed"] == pytest.approx(b["landed"], rel=1e-6)
    assert rel_l2(a["img"], b["img"]) <= 1e-5


DOCS = ["ms_multi_crystal_filtered", "ms_multi_crystal_complex_filter", "ms_multi_crystal_filtered_bd", "parity_big_or_with_color",
        "parity_single_ms_filter", "parity_single_ms_complex_filter", "parity_single_ms_bd_filter", "raypath_color_three_arcs", "raypath_color_multi_layer"]


def oracle_doc_multilayer(name, seed):
    """the oracle's render of a MULTI-layer filter document at 4 Mi rays, as the summary the test below reads — a committed fixture
    (tests/_oracle_cache.py, rendered by tests/golden/make_oracle_render_fixtures.py)"""
    def compute():
        job = config.load_config(_E2E_DOCS[name])
        rd = job.renders[sorted(job.renders)[0]]
        colors = (job.color_sets, job.color_classes) if job.color_classes else None
        o = _render(OracleBackend(seed=seed, threads=THREADS, acc64=1), job.scene, rd, job.wavelengths[0], 4 << 20, job.filters, colors, job.geom_clock)
        return {"cont": np.asarray([s.continuation_count for s in o["st"]], np.int64), "landed": np.float64(o["landed"]),
                "sum_y": np.float64(o["img"][..., 1].sum(dtype=np.float64)), "img_sum": np.float64(o["img"].sum(dtype=np.float64)),
                "y16": block_mean(o["img"], 16)[..., 1].astype(np.float32),
                "lane_sums": o["lanes"].sum(axis=(1, 2), dtype=np.float64) if colors else np.zeros(0)}
    from tests._oracle_cache import cached
    return cached("filterdoc_%s_seed%d" % (name, seed), compute, inputs=("ice_halo_sim_amd/config.py", ("doc", __import__("json").dumps(_E2E_DOCS[name], sort_keys=True))))


def multilayer_docs():
    return [n for n in DOCS if len(_E2E_DOCS[n]["scene"]["scattering"]) > 1]


@pytest.mark.parametrize("name", DOCS)
def test_reference_filter_documents_on_the_production_kernels(name):
    """The reference's filtered / colour-tagged end-to-end documents — its published GPU benchmark scenes ms_multi_crystal_filtered,
    ms_multi_crystal_complex_filter, ms_multi_crystal_filtered_bd (doc/performance-testing.md:465-468) among them — through the JSON
    reader at 4 Mi rays (5 Mi for single-layer documents), capture off, against the oracle: image, landed weight, channel sums,
    class lanes.  The route is asserted: filter dispatches on kModeFilter, colour-tagged ones on kModeColor, nothing on the capture or
    generic kernels.  (Multi-layer documents: the oracle's two renders are committed fixtures, oracle_doc_multilayer.)"""
    job = config.load_config(_E2E_DOCS[name])
    rd = job.renders[sorted(job.renders)[0]]
    wl = job.wavelengths[0]
    layers = job.scene.layer_count
    n = (5 << 20) if layers == 1 else (4 << 20)
    colors = (job.color_sets, job.color_classes) if job.color_classes else None
    h = _render(hip_backend(seed=42), job.scene, rd, wl, n, job.filters, colors, job.geom_clock)
    r = h["route"]
    assert not (r.mode_mask & (abi.MODE_CAPTURE | abi.MODE_GENERIC)), r.mode_mask
    assert r.mode_mask & (abi.MODE_COLOR if colors else abi.MODE_FILTER), r.mode_mask
    if colors:
        assert not (r.mode_mask & abi.MODE_FILTER)     # with raypath colour on, every dispatch carries masks
    if layers == 1:
        o = _render(OracleBackend(seed=42, threads=THREADS, acc64=1), job.scene, rd, wl, n, job.filters, colors, job.geom_clock)
        err = _single_layer_checks(h, o, name)
        if colors:
            th, to = h["lanes"].sum(axis=(1, 2), dtype=np.float64), o["lanes"].sum(axis=(1, 2), dtype=np.float64)
            # (class lanes are float images fed by one float atomic per hit on the GPU, like the reference's — cu:535-556 — and 5e7 of Y on a
            # 512x256 lane is where that accumulator itself is good to ~5e-4; the oracle sums in double here)
            assert th == pytest.approx(to, rel=1.5e-3, abs=1e-4 * max(float(to.max()), 1.0))
            for k in range(len(to)):      # and as images, class by class
                if to[k] > 1.0:
                    assert rel_l2(block_mean(h["lanes"][k][..., None]), block_mean(o["lanes"][k][..., None])) <= 4e-3, (name, k)
        print("%s: mode %d geom %d accum %d, block-mean rel L2 %s" % (name, r.mode_mask, r.geom_mask, r.accum_mask, err))
        return
    a, b = oracle_doc_multilayer(name, 42), oracle_doc_multilayer(name, 7)

    def within(x, p, q, abs_floor):
        return abs(x - p) <= 4.0 * abs(p - q) + abs_floor
    for l in range(layers - 1):
        p, q = int(a["cont"][l]), int(b["cont"][l])
        assert within(h["st"][l].continuation_count, p, q, 3e-3 * p + 50), (l, h["st"][l].continuation_count, p, q)
    la, lb = float(a["landed"]), float(b["landed"])
    assert within(h["landed"], la, lb, 6e-3 * la + 1.0), (h["landed"], la, lb)
    ya, yb, yh = float(a["sum_y"]), float(b["sum_y"]), float(h["img"][..., 1].sum(dtype=np.float64))
    assert within(yh, ya, yb, 6e-3 * ya + 1e-3), (yh, ya, yb)
    if float(a["img_sum"]) > 0 and la > 1000.0:
        pear = lambda x, y: float(np.corrcoef(x.ravel().astype(np.float64), y.ravel().astype(np.float64))[0, 1])
        floor = pear(a["y16"], b["y16"])
        got = pear(block_mean(h["img"], 16)[..., 1], a["y16"])
        assert got >= floor - 0.02, (got, floor)
        print("%s: mode %d geom %d accum %d, Pearson %.5f (oracle cross-seed %.5f), landed %.1f vs %.1f / %.1f" % (
            name, r.mode_mask, r.geom_mask, r.accum_mask, got, floor, h["landed"], la, lb))
    if colors:
        th, to, to2 = h["lanes"].sum(axis=(1, 2), dtype=np.float64), a["lane_sums"], b["lane_sums"]
        for k in range(len(to)):
            assert within(th[k], to[k], to2[k], 2e-2 * max(float(to.max()), 1.0) + 0.5), (name, k, th[k], to[k], to2[k])


def test_shape_pool_clock_does_not_bias_entry_exit_filter_admission():
    """test/parity-cross-backend/backend/test_cuda_kshape_filter_parity.cpp, both arms, on this engine.  Its scene: one layer, a stochastic
    hexagonal prism (height and six face distances gauss(1, 0.15), cuda_test_helpers.hpp:150-161) on the default fixed axis, sun (30, 0, 0.5),
    max_hits 6, an EntryExit(min_len = 3) filter with both faces wild — a filter that CONSUMES the recorded path — 1024 rays per batch, eight
    seeds, 64 x 64 fisheye.  Arm 1 (AC1, :196-240): the shape-pool clock only redistributes shape draws, so the mean admitted weight over the
    seeds is the same with a new shape per ray (geom_clock 1) as with one per 8 rays, inside 3 x the pooled standard error; both arms admit
    something.  Arm 2 (:241-293), which the reference can only hold to a loose ballpark (its CPU and CUDA paths trace different ray
    populations), is exact here: at clock 8 the engine admits the oracle's exits, count for count and weight for weight, every seed."""
    import math
    ee = scenes.simple_filter(scenes.filter_term("entry_exit", min_len=3), "")
    g = {"type": "gauss", "mean": 1.0, "std": 0.15}
    ent = scenes.entry(scenes.prism_crystal(g, [g] * 6), scenes.axis(), 1.0, 0, 1)
    sc = scenes.scene([(0.0, [ent])], max_hits=6, sun_altitude=30.0, sun_azimuth=0.0, sun_diameter=0.5)
    rd = scenes.render(abi.LENS_FISHEYE_EQUAL_AREA, 64, 64, fov=180.0, el=90.0, visible=abi.VISIBLE_UPPER)
    wl, n = scenes.wl_discrete(550.0), 1024
    w = {1: [], 8: []}
    for seed in range(1, 9):
        for clock in (1, 8):
            hb = hip_backend(seed=seed, geom_clock=clock)
            hb.set_filters([ee])
            st = run_session(hb, sc, rd, wl, n)[0]
            hb.close()
            w[clock].append(float(st.exit_w_sum))
            if clock == 8:
                ob = OracleBackend(seed=seed, threads=2, geom_clock=8)
                ob.set_filters([ee])
                so = run_session(ob, sc, rd, wl, n)[0]
                ob.close()
                assert int(st.exit_count) == int(so.exit_count) and 0 < so.exit_count < n * 6, (seed, st.exit_count, so.exit_count)
                assert float(st.exit_w_sum) == pytest.approx(float(so.exit_w_sum), rel=2e-4), seed

    def stats(xs):
        m = sum(xs) / len(xs)
        return m, math.sqrt(sum((x - m) ** 2 for x in xs) / len(xs))
    (m1, s1), (m8, s8) = stats(w[1]), stats(w[8])
    assert m1 > 0.0 and m8 > 0.0
    assert abs(m1 - m8) < 3.0 * math.sqrt(s1 * s1 + s8 * s8) / math.sqrt(8.0), (m1, s1, m8, s8)
