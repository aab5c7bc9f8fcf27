"""Randomised scenes, HIP vs the oracle ray by ray (capture on, one scattering layer so that the rays pair up deterministically).

The hand-written parity tests and the reference's 61 e2e documents walk the features one or two at a time; this draws them together at
random — crystal family and its stochastic parameters (sync groups included), the three axis distributions of every type, one to three
crystal entries per layer, max_hits, sun, all eleven lenses with random view / field of view / visible range / lens shift, discrete
wavelength or an illuminant pool, and (half of the cases) an emit-gate filter of any kind, symmetry and action, simple or complex.  Same
per-ray bar as `test_reference_e2e_configs_parity`: exit counts, >= 99.7 % of the exits matched in direction and weight, their pixel and
face path, the landed weight and the block-mean image.

`FUZZ_SEEDS` (environment, e.g. `FUZZ_SEEDS=1000:1400`) runs another range — that is how the seeds below were first swept."""
import os

import numpy as np
import pytest

from ice_halo_sim_amd import abi, scenes
from tests._oracle_backend import OracleBackend, run_session
from tests.test_gpu_parity import block_mean, hip_backend, match_exits, match_exits_conditioned, rel_l2

pytestmark = pytest.mark.gpu

PYR_FACES = [1, 2, 3, 4, 5, 6, 7, 8, 13, 14, 15, 16, 17, 18, 23, 24, 25, 26, 27, 28]


def _dist(rng, centre, spreads, kinds=("uniform", "gauss", "zigzag", "laplacian")):
    return {"type": str(rng.choice(kinds)), "mean": float(centre), "std": float(rng.choice(spreads))}


def _face_distances(rng):
    r = rng.random()
    if r < 0.45:
        return None
    if r < 0.7:
        return [float(rng.uniform(0.6, 1.4)) for _ in range(6)]
    return [_dist(rng, rng.uniform(0.8, 1.2), [0.05, 0.15, 0.4], ("uniform", "gauss")) for _ in range(6)]


def _crystal(rng):
    # sync groups tie heights to heights and face distances to face distances; FUZZ_ANY_GROUPS=1 sweeps heights and distances in one group
    # as well (a legal document with tiny or negative face distances: before the builder took exact incidences those crystals were no
    # polytopes — seed 732 — now 599 of 600 such seeds sit inside the bars, the other at 0.993 matched)
    groups = ([int(g) for g in rng.choice([0, 0, 1], 3)] + [int(g) for g in rng.choice([0, 0, 0, 2, 3], 6)]) if rng.random() < 0.3 else None
    if groups is not None and os.environ.get("FUZZ_ANY_GROUPS"):   # a sweep with heights and face distances in one group (tiny or negative distances)
        groups = [int(g) for g in rng.choice([0, 0, 0, 1, 2], 9)]
    if rng.random() < 0.55:
        h = float(rng.choice([0.1, 0.3, 1.0, 1.3, 3.0])) if rng.random() < 0.6 else _dist(rng, rng.uniform(0.3, 2.0), [0.1, 0.5], ("uniform", "gauss"))
        return scenes.prism_crystal(h, _face_distances(rng), sync_group=groups), list(range(1, 9))

    def frac():
        return float(rng.choice([0.0, 0.1, 0.5, 1.0])) if rng.random() < 0.6 else _dist(rng, rng.uniform(0.1, 0.8), [0.05, 0.3], ("uniform", "gauss"))
    kw = {}
    if rng.random() < 0.5:
        kw["upper_miller"], kw["lower_miller"] = [(1, 1), (2, 3), (1, 2), (3, 2)][rng.integers(4)], [(1, 1), (2, 3), (1, 2)][rng.integers(3)]
    else:
        kw["upper_wedge"], kw["lower_wedge"] = float(rng.uniform(8, 80)), float(rng.uniform(8, 80))
    ph = float(rng.choice([0.0, 0.4, 1.0, 2.0])) if rng.random() < 0.6 else _dist(rng, rng.uniform(0.3, 1.5), [0.1, 0.4], ("uniform", "gauss"))
    return scenes.pyramid_crystal(frac(), ph, frac(), face_distance=_face_distances(rng), sync_group=groups, **kw), PYR_FACES


def _axis(rng):
    r = rng.random()
    if r < 0.15:
        return scenes.axis()
    kinds = ("uniform", "gauss", "zigzag", "laplacian", "gauss_legacy")
    zen = float(rng.choice([0.0, 90.0, 35.0])) if rng.random() < 0.15 else _dist(rng, rng.choice([0.0, 90.0, rng.uniform(0, 180)]), [0.3, 5.0, 40.0, 360.0], kinds)
    az = None if rng.random() < 0.5 else (float(rng.uniform(0, 360)) if rng.random() < 0.3 else _dist(rng, rng.uniform(0, 360), [2.0, 60.0, 360.0]))
    ro = None if rng.random() < 0.5 else (float(rng.uniform(0, 360)) if rng.random() < 0.3 else _dist(rng, rng.uniform(0, 360), [2.0, 60.0, 360.0]))
    return scenes.axis(zenith=zen, azimuth=az, roll=ro)


def _term(rng, faces):
    kind = str(rng.choice(["raypath", "entry_exit", "direction", "crystal"]))
    if kind == "raypath":
        return scenes.filter_term("raypath", raypath=[int(rng.choice(faces)) for _ in range(int(rng.integers(1, 6)))])
    if kind == "entry_exit":
        lo = int(rng.integers(1, 4))
        return scenes.filter_term("entry_exit", entry=int(rng.choice(faces)) if rng.random() < 0.7 else None, exit=int(rng.choice(faces)) if rng.random() < 0.7 else None,
                                  min_len=lo, max_len=None if rng.random() < 0.4 else lo + int(rng.integers(0, 5)))
    if kind == "direction":
        return scenes.filter_term("direction", az=float(rng.uniform(0, 360)), el=float(rng.uniform(-60, 80)), radii=float(rng.choice([2.0, 15.0, 60.0])))
    return scenes.filter_term("crystal", crystal_id=int(rng.integers(1, 5)))


def _filter(rng, faces):
    sym = "".join(c for c in "PBD" if rng.random() < 0.5)
    action = "filter_in" if rng.random() < 0.6 else "filter_out"
    if rng.random() < 0.7:
        return scenes.simple_filter(_term(rng, faces), sym, action)
    return scenes.complex_filter([[_term(rng, faces) for _ in range(int(rng.integers(1, 3)))] for _ in range(int(rng.integers(1, 4)))], sym, action)


def make_case(seed):
    rng = np.random.default_rng(seed)
    entries, filters = [], []
    for k in range(int(rng.choice([1, 1, 2, 3]))):
        cr, faces = _crystal(rng)
        fid = 0
        if rng.random() < 0.5:
            filters.append(_filter(rng, faces))
            fid = len(filters)
        entries.append(scenes.entry(cr, _axis(rng), float(rng.uniform(0.2, 3.0)), k + 1, fid))
    max_hits = int(rng.choice([1, 2, 4, 7, 8, 12]))
    if os.environ.get("FUZZ_MAX_HITS"):   # sweeps of the long-path side: 17+ runs the generic filter kernels (paths longer than the 128-bit register)
        max_hits = int(os.environ["FUZZ_MAX_HITS"])
    sc = scenes.scene([(0.0, entries)], max_hits=max_hits, sun_altitude=float(rng.uniform(-10, 85)), sun_azimuth=float(rng.uniform(0, 360)),
                      sun_diameter=float(rng.choice([0.0, 0.5, 2.0])))
    lens = int(rng.integers(0, 11))
    w, h = [(64, 48), (200, 100), (333, 211), (512, 256), (640, 640)][rng.integers(5)]
    fov = float(rng.uniform(20, 120)) if lens == abi.LENS_LINEAR else float(rng.choice([180.0, 120.0, 90.0, 200.0]))
    rd = scenes.render(lens, w, h, fov=fov, az=float(rng.uniform(0, 360)), el=float(rng.uniform(-30, 90)), ro=float(rng.choice([0.0, rng.uniform(0, 360)])),
                       visible=int(rng.integers(0, 3)), overlap=float(rng.choice([0.0, 0.1])), lens_shift=(int(rng.integers(-20, 21)), int(rng.integers(-20, 21))) if rng.random() < 0.3 else (0, 0))
    wl = scenes.wl_discrete(float(rng.uniform(400, 700))) if rng.random() < 0.6 else scenes.wl_illuminant(str(rng.choice(["D65", "D50", "A", "E"])), int(rng.choice([1, 7, 31, 64])))
    return sc, rd, wl, filters, int(rng.choice([8, 32, 64]))


def has_degenerate_tables(sc, seed, samples=8):
    """The fraction of a pyramid entry's sampled instances (the first `samples` of the session's shape stream) whose vertex / face table is no
    polytope (fan triangles != 2 V - 4) or has a fan corner off its own face's plane by more than 1e-6 (ordinary tables: <= 1.2e-7, the
    rounding of the corners to float).  Until the builder took exact incidences (csrc/halo_geom.h) full apexes (height fraction 1)
    over face distances 0.5 % apart did the first: a corner within the tolerance of a plane it does not belong to joined that face and tilted
    its fan off the plane.  The second is what the vertex merge leaves of a pyramid segment thinner than the merge tolerance (seed 20234: upper
    segments of 5.6e-5 and 3.8e-5 of the crystal in 2 of its 937 instances — the ring's corners merge into the basal face's, 1.2e-5 above, and
    the six lower faces' fans end 1.8e-5 .. 2.7e-5 off their planes).  On such tables the reference's two next-face strategies part ways
    (src/core/shared/traversal_shared.h:23-29: the CPU path's relaxed-threshold test lets a child leaving a tilted face 're-hit' it — 58 % of
    the outside reflections on seed 20234's two instances — the CUDA path's explicit skip, and this library's rule that the outgoing child
    leaves, do not), so the oracle (the CPU strategy) and HIP agree only loosely on those instances' rays."""
    import ctypes as C
    from ice_halo_sim_amd import backend
    from tests import _libs
    from tests._libs import fptr
    L, O = backend.load_library(), _libs.oracle()
    O.ho_pyramid_face_mask.restype = C.c_int
    L0 = sc.layers[0]
    worst = 0.0
    for i in range(L0.entry_count):
        cr = L0.entries[i].crystal
        if cr.kind != abi.CRYSTAL_PYRAMID:
            continue
        bad = 0
        for idx in range(samples):
            scal = np.zeros(9, np.float32)
            L.halo_host_shape_scalars(C.byref(cr), seed, idx, 0, fptr(scal))
            d = scal[3:9].copy()
            nv, g = C.c_int(0), abi.HaloGeomTables()
            args = (cr.wedge_upper_deg, cr.wedge_lower_deg, abs(float(scal[0])), abs(float(scal[1])), abs(float(scal[2])), fptr(d))
            O.ho_pyramid_face_mask(*args, C.byref(nv))
            O.ho_pyramid_geometry(*args, C.byref(g))
            fc, tc = g.face_cnt, g.tri_cnt
            if fc == 0:
                continue
            fn = np.frombuffer(g.face_n, np.float32)[:fc * 3].reshape(fc, 3).astype(np.float64)
            fd = np.frombuffer(g.face_d, np.float32)[:fc].astype(np.float64)
            tv = np.frombuffer(g.tri_v, np.float32)[:tc * 9].reshape(tc, 3, 3).astype(np.float64)
            tf = np.frombuffer(g.tri_face, np.int32)[:tc]
            off = float(np.abs(np.einsum("tkc,tc->tk", tv, fn[tf]) + fd[tf][:, None]).max())
            bad += int(tc != 2 * nv.value - 4 or off > 1e-6)
        worst = max(worst, bad / float(samples))
    return worst


def run_case(seed, n=60_000, hip_opts=None, **oracle_opts):
    sc, rd, wl, filters, clock = make_case(seed)
    hb = hip_backend(seed=seed, capture_exits=1, geom_clock=clock, **(hip_opts or {}))
    ob = OracleBackend(seed=seed, capture_exits=1, threads=8, geom_clock=clock, **oracle_opts)
    ob2 = OracleBackend(seed=seed, fma=True, capture_exits=1, threads=8, geom_clock=clock, **oracle_opts)   # the oracle's other rounding: which exits are ill-conditioned
    for b in (hb, ob, ob2):
        b.set_filters(filters)
    sh = run_session(hb, sc, rd, wl, n)
    so = run_session(ob, sc, rd, wl, n)
    run_session(ob2, sc, rd, wl, n)
    eh, eo, eo2 = hb.DrainExits(), ob.DrainExits(), ob2.DrainExits()
    ih, lh = hb.ReadbackXyzAccum()
    io, lo = ob.ReadbackXyzAccum()
    hb.close()
    ob.close()
    ob2.close()
    L0 = sc.layers[0]
    fixed = any(L0.entries[i].axis.latitude.type == abi.DIST_NONE and L0.entries[i].axis.azimuth.type == abi.DIST_NONE and L0.entries[i].axis.roll.type == abi.DIST_NONE
                for i in range(L0.entry_count))   # an entry whose every ray meets the crystal the same way
    # (every instance the session can draw: a layer's entries take consecutive runs of the shape stream, one instance per `clock` rays)
    out = dict(exits=(sh[0].exit_count, so[0].exit_count), landed=(lh, lo), n_exits=(len(eh), len(eo)), fixed_axes=fixed,
               degenerate=has_degenerate_tables(sc, seed, samples=n // max(int(clock), 1) + L0.entry_count + 1))
    out["match"] = match_exits(eh, eo) if len(eo) and len(eh) else (1.0 if len(eh) == len(eo) else 0.0, 1.0, 1.0)
    out["cond"] = match_exits_conditioned(eh, eo, eo2) if len(eo) and len(eh) else (1.0 if len(eh) == len(eo) else 0.0, 1.0, 1.0, 0)
    out["oracle_pair"] = match_exits(eo2, eo)[0] if len(eo) and len(eo2) else 1.0
    # what the exits that do not pair up (or pair up with another weight) can move the landed weight by, at most
    n_un = int(round((1.0 - out["match"][0]) * (len(eh) + len(eo)))) + 1
    wmax = max(float(eh["weight"].max()) if len(eh) else 0.0, float(eo["weight"].max()) if len(eo) else 0.0)
    out["unmatched_weight"] = n_un * wmax if out["match"][0] < 1.0 else 0.0
    # exits that pair up within the direction bar but land in the frame on ONE side only (a point within 1e-6 of the frame's or the visible
    # hemisphere's border): their whole weight is in one landed sum and not in the other (seed 40119: one exit of weight 0.619 in 945)
    out["border"] = (0, 0.0)
    out["moved_roots"] = 0
    if len(eo) and len(eh):
        kh = (eh["layer"].astype(np.int64) << 48) | (eh["root"].astype(np.int64) << 8) | eh["seq"].astype(np.int64)
        ko = (eo["layer"].astype(np.int64) << 48) | (eo["root"].astype(np.int64) << 8) | eo["seq"].astype(np.int64)
        oh, oo = np.argsort(kh), np.argsort(ko)
        _, ch, co = np.intersect1d(kh[oh], ko[oo], return_indices=True)
        xa, xb = eh[oh][ch], eo[oo][co]
        one_side = ((xa["pixel"] < 0) != (xb["pixel"] < 0)) & (np.abs(xa["dir"] - xb["dir"]).max(axis=1) <= 2e-5)
        out["border"] = (int(one_side.sum()), float(np.maximum(xa["weight"][one_side], xb["weight"][one_side]).sum()))
        moved = (xa["pixel"] != xb["pixel"]) & (np.abs(xa["dir"] - xb["dir"]).max(axis=1) <= 2e-5)
        out["moved_roots"] = int(len(np.unique(xa["root"][moved])))    # RAYS with a paired exit in another pixel (a plate sends one direction out many times)
    out["l2"] = rel_l2(block_mean(ih, 4), block_mean(io, 4)) if io.sum() > 0 else 0.0
    by = block_mean(io, 4)[..., 1].astype(np.float64)
    out["n_eff"] = float(by.sum() ** 2 / max((by * by).sum(), 1e-300))   # how many blocks carry the image (one heavy exit in a few dozen: seed 11011)
    return out


def check(seed, r):
    """The per-ray bars are those of the e2e documents, applied to the exits the ORACLE itself is sure of.  What a random scene does that
    the documents do not, found by sweeping 3300 seeds (tools/diag_fuzz.py shows any seed in detail): (1) an entry with every axis FIXED —
    all its rays then meet the crystal the same way, and if that way grazes a face (sun 0.35 degrees below a plate's basal plane: cos of the
    incidence angle passes through zero across the sun's disc, the transmitted weight 1 - R is ill-conditioned there) or a critical angle,
    the weights of 1-9 % of the exits move by 3e-4 .. 3e-3 on every ray that goes that way.  Round 3 blamed the Fresnel split's hardware
    reciprocal / square root and lowered the bar to 0.9 for such scenes; round 4 measured it (tools/fresnel_ab.py): IEEE division
    and square root change nothing (seed 11584: 0.905 with, 0.910 without), because the ORACLE compiled with FMA contraction disagrees with
    the oracle compiled without on the same exits (0.918 there; 0.945 / 0.951 / 0.976 on seeds 247 / 411 / 702) — the reference's own
    result is not defined more finely than that, and a HIP build WITHOUT contraction pairs up on 99.9 % (at 14 % of the throughput).  So the
    yardstick is that pair (match_exits_conditioned): 0.995 of the exits within the bars widened, per exit, by four times what the two
    oracles differ by on it.
    (2) small sparse images at 60 k rays, where ONE exit crossing a pixel border is 1 % of a block-mean distance: that bar is 5e-2 +
    2 / sqrt(blocks that carry the image) here, the per-ray bars carry the comparison; (3) illuminant weights of ~100 per exit: the
    landed weights may differ by what the unmatched exits weigh.
    Round 6's sweep of 400 new seeds (tests/golden/FUZZ_SWEEPS.md) added two: (4) an exit that pairs up (directions 9e-7 apart) and lies
    ON the frame's border lands on one side only — its whole weight is in one landed sum (seed 40119); such exits are counted (at most
    2 + 1e-4 of the exits) and their weight is allowed for; (5) with every axis fixed the oracle pair is ONE sample of the rounding, not a
    distribution: on seed 40254 (sun 0.28 degrees below a fixed pyramid's basal plane, sun diameter 0: 20 k identical rays) the two oracles
    agree with each other on 0.984 of the exits, the product with the oracle on 0.981 (0.983 conditioned) — on OTHER exits than the oracle
    pair's, so the per-exit widening does not cover them — and the strict build on 0.99991 unconditioned.  For fixed-axes scenes the
    conditioned bar is therefore the oracle pair's own agreement less 0.01, never above 0.995.  And one from the 2000 seeds after that: (6) the
    same-pixel fraction is taken over the exits on which the two oracles agree to the bit — 337 of 33 449 on seed 41973 — and ONE ray whose
    direction sits 1 ulp from a pixel column's border leaves a plate five times in that direction: 5 / 337 = 1.5 %.  The fraction stays the
    bar; where it fails, the RAYS with a paired exit (directions within 2e-5) in another pixel must be no more than 2 + 1e-4 of the exits."""
    deg = float(r["degenerate"])   # the fraction of an entry's crystal instances on which the two sides follow the reference's two next-face strategies (has_degenerate_tables)
    if deg > 0.02:
        assert r["exits"][0] == pytest.approx(r["exits"][1], rel=6e-2, abs=20), (seed, r)
        assert r["match"][0] >= 0.85, (seed, r)
        return
    assert r["exits"][0] == pytest.approx(r["exits"][1], rel=1e-3 + 2.0 * deg, abs=20), (seed, r)
    frac, pix, path, left_out = r["cond"]
    frac_bar = min(0.995, r["oracle_pair"] - 0.01) if r["fixed_axes"] else 0.995
    assert frac >= frac_bar - 2.0 * deg and path >= 0.998 - 2.0 * deg, (seed, r)
    assert pix >= 0.995 - 2.0 * deg or r["moved_roots"] <= 2 + 1e-4 * r["n_exits"][1], (seed, r)
    assert left_out <= 2e-3 * r["n_exits"][1] + 5, (seed, r)
    assert r["match"][0] >= min(0.995, r["oracle_pair"] - 0.03), (seed, r)          # and never far below what the oracles reach between themselves
    assert r["border"][0] <= 2 + 1e-4 * r["n_exits"][1], (seed, r)
    assert abs(r["landed"][0] - r["landed"][1]) <= 3e-4 * max(r["landed"][1], 1.0) + 1e-3 + r["unmatched_weight"] + r["border"][1], (seed, r)
    assert r["l2"] <= 5e-2 + 2.0 / np.sqrt(max(r["n_eff"], 1.0)), (seed, r)


def _seeds():
    spec = os.environ.get("FUZZ_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    # 20234: two instances with a segment thinner than the vertex merge (has_degenerate_tables); 40119 / 40254 / 41973: check()'s cases (4), (5) and (6)
    return list(range(100, 148)) + [20234, 40119, 40254, 41973]


@pytest.mark.parametrize("seed", _seeds())
def test_random_scene_traces_the_same_rays_as_the_oracle(seed):
    check(seed, run_case(seed))


# ---- the same random scenes on the PRODUCTION kernels (capture off, launches big enough for the hit log) --------------------------------
THREADS = max(8, min(os.cpu_count() or 8, 128))


def run_production_case(seed, n=3 << 20, big=False):
    """Capture off: plain / kModeFilter kernels with the exit queue, hit log or binned routes, pool or one-shape geometry as the scene
    asks.  Same streams on both sides, so image, landed weight and exit count are compared like the single-layer production tests
    (tests/test_gpu_filter_production.py), the oracle with its double accumulator."""
    sc, rd, wl, filters, clock = make_case(seed)
    if big:   # a production-size image, and sessions long enough for the per-entry-plane / binned / two-level routes of illuminant renders
        rd.width, rd.height = [(1920, 1080), (2048, 1024), (4096, 2048)][seed % 3]
        clock = 32 if clock == 8 else clock
    hb = hip_backend(seed=seed, geom_clock=clock)
    ob = OracleBackend(seed=seed, threads=THREADS, acc64=1, geom_clock=clock)
    for b in (hb, ob):
        b.set_filters(filters)
    sh = run_session(hb, sc, rd, wl, n)
    route = hb.last_route()
    so = run_session(ob, sc, rd, wl, n)
    ih, lh = hb.ReadbackXyzAccum()
    io, lo = ob.ReadbackXyzAccum()
    hb.close()
    ob.close()
    L0 = sc.layers[0]
    fixed = any(L0.entries[i].axis.latitude.type == abi.DIST_NONE and L0.entries[i].axis.azimuth.type == abi.DIST_NONE and L0.entries[i].axis.roll.type == abi.DIST_NONE
                for i in range(L0.entry_count))   # an entry whose every ray meets the crystal the same way
    io = np.asarray(io, np.float32)
    return dict(exits=(sh[0].exit_count, so[0].exit_count), landed=(lh, lo), mode_mask=route.mode_mask, accum_mask=route.accum_mask, fixed_axes=fixed,
                l2=rel_l2(block_mean(ih, 8), block_mean(io, 8)) if io.sum() > 0 else 0.0,
                sums=(ih.sum(axis=(0, 1), dtype=np.float64), io.sum(axis=(0, 1), dtype=np.float64)),
                floor=oracle_pair_floor(seed, sc, rd, wl, filters, clock, n, so, io, lo) if fixed else None)


def oracle_pair_floor(seed, sc, rd, wl, filters, clock, n, so, io, lo, sets=None, classes=None, lanes_o=None):
    """How far the oracle is from ITSELF on this scene when its products are contracted into FMAs (liboracle_fma.so): the yardstick for
    scenes whose every ray meets the crystal the same way (see check()).  Relative distances in exit count, landed weight, 8x8 block-mean
    image, channel sums (and class lanes)."""
    ob2 = OracleBackend(seed=seed, fma=True, threads=THREADS, acc64=1, geom_clock=clock)
    ob2.set_filters(filters)
    if sets is not None:
        ob2.set_color(sets, classes)
    s2 = run_session(ob2, sc, rd, wl, n)
    i2, l2 = ob2.ReadbackXyzAccum()
    lanes2 = ob2.ReadbackClassLanes() if sets is not None else None
    ob2.close()
    i2 = np.asarray(i2, np.float32)
    f = dict(exits=abs(int(s2[0].exit_count) - int(so[0].exit_count)) / max(int(so[0].exit_count), 1), landed=abs(l2 - lo) / max(lo, 1.0),
             l2=rel_l2(block_mean(i2, 8), block_mean(io, 8)) if io.sum() > 0 else 0.0,
             sums=float(np.max(np.abs(i2.sum(axis=(0, 1), dtype=np.float64) - io.sum(axis=(0, 1), dtype=np.float64)) / max(float(io.sum(dtype=np.float64)), 1e-30))))
    if lanes2 is not None:
        a, b = lanes2.sum(axis=(1, 2), dtype=np.float64), lanes_o.sum(axis=(1, 2), dtype=np.float64)
        f["lanes"] = float(np.max(np.abs(a - b) / np.maximum(b, 1.0)))
        f["lane_l2"] = max([rel_l2(block_mean(lanes2[k][..., None], 8), block_mean(lanes_o[k][..., None], 8)) if lanes_o[k].sum() > 0 else 0.0 for k in range(len(lanes_o))] + [0.0])
    return f


def check_production(seed, r):
    if int(os.environ.get("FUZZ_MAX_HITS", "0")) <= 16:
        assert not (r["mode_mask"] & (abi.MODE_CAPTURE | abi.MODE_GENERIC)), (seed, r)      # max_hits <= 16: the production-shaped kernels
    else:
        assert not (r["mode_mask"] & abi.MODE_CAPTURE), (seed, r)                           # (a sweep with longer paths: filtered entries run the generic kernels)
    # every ray the same way (see check()): the plain bars plus four times what the oracle's two roundings differ by on this scene
    fl = r["floor"] or dict(exits=0.0, landed=0.0, l2=0.0, sums=0.0)
    assert r["exits"][0] == pytest.approx(r["exits"][1], rel=3e-4 + 4.0 * fl["exits"], abs=20), (seed, r)
    # (a heavily filtered scene lands a few thousand exits of 9 Mi rays: the two or three that differ between the sides weigh what an exit weighs)
    per_exit = max(r["landed"][1], 1.0) / max(r["exits"][1], 1)
    slack = 3.0 * per_exit * (abs(r["exits"][0] - r["exits"][1]) + 2)
    assert abs(r["landed"][0] - r["landed"][1]) <= (3e-4 + 4.0 * fl["landed"]) * max(r["landed"][1], 1.0) + 1e-3 + slack, (seed, r)
    assert r["l2"] <= 3e-3 + 4.0 * fl["l2"] + 4.0 / np.sqrt(max(r["exits"][1], 1)), (seed, r)   # (seed 6125: 3e5 exits of 9 Mi rays on 2 M pixels, 7.2e-3)
    tot = float(r["sums"][1].sum())
    for ch in range(3):
        assert r["sums"][0][ch] == pytest.approx(r["sums"][1][ch], rel=5e-4, abs=(1e-5 + 4.0 * fl["sums"]) * tot + 1e-6 + slack * tot / max(r["landed"][1], 1.0)), (seed, ch, r)


def _big_seeds():
    spec = os.environ.get("FUZZ_BIG_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    return list(range(6000, 6005)) + [6011]   # 6011: a one-entry illuminant pool on 4096 x 2048 — 512 hit-log tiles on ONE plane (it faulted before the fix)


@pytest.mark.parametrize("seed", _big_seeds())
def test_random_scene_on_the_production_kernels_big_image(seed):
    """The same at 9 Mi rays on 1920x1080 / 2048x1024 / 4096x2048: scalar-plane hit log, X/Y/Z log, one plane per pool entry with the binned
    and two-level routes — whichever the session asks for."""
    check_production(seed, run_production_case(seed, n=9 << 20, big=True))


def _prod_seeds():
    spec = os.environ.get("FUZZ_PROD_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    return list(range(2000, 2016))


@pytest.mark.parametrize("seed", _prod_seeds())
def test_random_scene_on_the_production_kernels(seed):
    check_production(seed, run_production_case(seed))


# ---- multi-scattering: two or three layers --------------------------------------------------------------------------------------------
def make_ms_case(seed):
    rng = np.random.default_rng(seed)
    layers, filters = [], []
    n_layers = int(rng.choice([2, 2, 3]))
    for li in range(n_layers):
        entries = []
        for k in range(int(rng.choice([1, 1, 2]))):
            cr, faces = _crystal(rng)
            fid = 0
            if rng.random() < 0.3:
                filters.append(_filter(rng, faces))
                fid = len(filters)
            entries.append(scenes.entry(cr, _axis(rng), float(rng.uniform(0.2, 3.0)), 10 * li + k + 1, fid))
        prob = 0.0 if li == n_layers - 1 else float(rng.choice([0.2, 0.5, 0.8, 1.0]))
        layers.append((prob, entries))
    sc = scenes.scene(layers, max_hits=int(rng.choice([2, 4, 7, 8])), sun_altitude=float(rng.uniform(0, 70)), sun_azimuth=float(rng.uniform(0, 360)), sun_diameter=0.5)
    lens = int(rng.choice([abi.LENS_FISHEYE_EQUAL_AREA, abi.LENS_DUAL_FISHEYE_EQUAL_AREA, abi.LENS_RECTANGULAR, abi.LENS_LINEAR]))
    rd = scenes.render(lens, 256, 128, fov=float(rng.uniform(40, 110)) if lens == abi.LENS_LINEAR else 180.0, az=float(rng.uniform(0, 360)), el=float(rng.uniform(0, 90)),
                       visible=int(rng.choice([abi.VISIBLE_UPPER, abi.VISIBLE_FULL])))
    wl = scenes.wl_discrete(float(rng.uniform(400, 700))) if rng.random() < 0.6 else scenes.wl_illuminant("D65", int(rng.choice([7, 31])))
    return sc, rd, wl, filters, int(rng.choice([32, 64]))


def run_ms_case(seed, n=100_000):
    sc, rd, wl, filters, clock = make_ms_case(seed)
    hb = hip_backend(seed=seed, capture_exits=1, geom_clock=clock)
    ob = OracleBackend(seed=seed, capture_exits=1, threads=8, geom_clock=clock)
    for b in (hb, ob):
        b.set_filters(filters)
    sh = run_session(hb, sc, rd, wl, n)
    so = run_session(ob, sc, rd, wl, n)
    eh, eo = hb.DrainExits(), ob.DrainExits()
    ih, lh = hb.ReadbackXyzAccum()
    io, lo = ob.ReadbackXyzAccum()
    hb.close()
    ob.close()
    e0h, e0o = eh[eh["layer"] == 0], eo[eo["layer"] == 0]
    # first-layer exits of the oracle's other rounding (liboracle_fma.so): which of them are ill-conditioned (see check())
    o3 = OracleBackend(seed=seed, fma=True, capture_exits=1, threads=8, geom_clock=clock)
    o3.set_filters(filters)
    o3.BeginSession(sc, rd, wl, n)
    o3.TraceLayer(n)
    e0o2 = o3.DrainExits()
    o3.EndSession()
    o3.close()
    e0o2 = e0o2[e0o2["layer"] == 0]
    # the oracle's own seed-to-seed scatter of the landed weight on this scene (two more seeds): the yardstick for the layers >= 1
    others = []
    for s2 in (seed + 1000, seed + 2000):
        o2 = OracleBackend(seed=s2, threads=8, geom_clock=clock)
        o2.set_filters(filters)
        run_session(o2, sc, rd, wl, n)
        others.append(o2.ReadbackXyzAccum()[1])
        o2.close()
    L0 = sc.layers[0]
    fixed = any(L0.entries[i].axis.latitude.type == abi.DIST_NONE and L0.entries[i].axis.azimuth.type == abi.DIST_NONE and L0.entries[i].axis.roll.type == abi.DIST_NONE
                for i in range(L0.entry_count))   # an entry whose every ray meets the crystal the same way
    return dict(layers=sc.layer_count, landed_other_seeds=others, fixed_axes=fixed, cont=[(a.continuation_count, b.continuation_count) for a, b in zip(sh, so)], exits=(len(eh), len(eo)),
                first=match_exits(e0h, e0o) if len(e0h) and len(e0o) else (1.0 if len(e0h) == len(e0o) else 0.0, 1.0, 1.0), n_first=(len(e0h), len(e0o)), landed=(lh, lo),
                first_cond=match_exits_conditioned(e0h, e0o, e0o2) if len(e0h) and len(e0o) else (1.0 if len(e0h) == len(e0o) else 0.0, 1.0, 1.0, 0))


def check_ms(seed, r):
    """The first layer traces the same rays on both sides: its exits pair up ray by ray and its continuation count is the same number.  From
    the second layer on the continuation order differs (GPU append order vs the oracle's threads), so the rays are other draws of the same
    population: counts agree statistically — a few per cent at 100 k roots, against 1/sqrt(N) of the smallest count — and the landed weight
    within the oracle's own seed-to-seed scatter."""
    # (seed 3201 has the sun 0.66 degrees above the horizon on crystals whose c axis is held horizontal — grazing incidence, 0.55 % of the
    # first layer's exits differ in weight by 3e-4 .. 1e-3, between the oracle's two roundings as well: the conditioned match of check())
    frac, pix, path, left_out = r["first_cond"]
    assert frac >= 0.995 and path >= 0.998, (seed, r)
    c0h, c0o = r["cont"][0]
    assert c0h == pytest.approx(c0o, rel=1e-3, abs=20), (seed, r)
    for l in range(1, r["layers"] - 1):
        a, b = r["cont"][l]
        assert a == pytest.approx(b, rel=3e-2, abs=6.0 * np.sqrt(max(b, 1.0)) + 50), (seed, l, r)
    assert r["exits"][0] == pytest.approx(r["exits"][1], rel=3e-2, abs=6.0 * np.sqrt(max(r["exits"][1], 1.0)) + 50), (seed, r)
    # landed weight = what falls inside the frame: a few heavy exits can carry it (seed 3019: 265 .. 398 over four oracle seeds), so the bar
    # is the oracle's own scatter on this scene — HIP within 4 spreads of the oracle's three renders (+ 2 % + a small absolute term)
    ref = [r["landed"][1]] + list(r["landed_other_seeds"])
    spread = max(ref) - min(ref)
    assert abs(r["landed"][0] - float(np.mean(ref))) <= 4.0 * spread + 2e-2 * float(np.mean(ref)) + 2.0, (seed, r)


def _ms_seeds():
    spec = os.environ.get("FUZZ_MS_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    return list(range(3000, 3016))


@pytest.mark.parametrize("seed", _ms_seeds())
def test_random_multi_scatter_scene(seed):
    check_ms(seed, run_ms_case(seed))


def run_ms_production_case(seed, n=1 << 21):
    """Multi-scatter with capture off at 2 Mi roots: first layers on the no-accumulation / filter kernels, later layers from the continuation
    pool (transit source) through the hit log."""
    sc, rd, wl, filters, clock = make_ms_case(seed)
    hb = hip_backend(seed=seed, geom_clock=clock)
    hb.set_filters(filters)
    sh = run_session(hb, sc, rd, wl, n)
    route = hb.last_route()
    ih, lh = hb.ReadbackXyzAccum()
    hb.close()
    ref = []
    for s2 in (seed, seed + 1000):
        ob = OracleBackend(seed=s2, threads=THREADS, acc64=1, geom_clock=clock)
        ob.set_filters(filters)
        so = run_session(ob, sc, rd, wl, n)
        io, lo = ob.ReadbackXyzAccum()
        ob.close()
        ref.append((so, np.asarray(io, np.float32), lo))
    return dict(layers=sc.layer_count, mode_mask=route.mode_mask, source_mask=route.source_mask, cont=[[x.continuation_count for x in sh]] + [[x.continuation_count for x in r[0]] for r in ref],
                exits=[sum(x.exit_count for x in sh)] + [sum(x.exit_count for x in r[0]) for r in ref], landed=[lh] + [r[2] for r in ref],
                y=[float(ih[..., 1].sum(dtype=np.float64))] + [float(r[1][..., 1].sum(dtype=np.float64)) for r in ref])


def check_ms_production(seed, r):
    assert not (r["mode_mask"] & (abi.MODE_CAPTURE | abi.MODE_GENERIC)), (seed, r)   # production kernels
    if all(c[0] == 0 for c in r["cont"][1:]):   # a first-layer filter nothing passes (six of the sixty scenes 20000..20059): the oracle continues no ray, nor may the device
        assert r["cont"][0][0] == 0 and r["exits"][0] == r["exits"][1] and r["landed"][0] == pytest.approx(r["landed"][1], rel=1e-4, abs=1e-3), (seed, r)
        return
    assert r["source_mask"] & 2, (seed, r)   # transit source used
    assert r["cont"][0][0] == pytest.approx(r["cont"][1][0], rel=1e-3, abs=20), (seed, r)     # same seed, same first-layer rays

    def within(vals, rel, abs_floor):   # HIP (vals[0]) against the oracle's two seeds
        ref = vals[1:]
        return abs(vals[0] - float(np.mean(ref))) <= 5.0 * (max(ref) - min(ref)) + rel * float(np.mean(ref)) + abs_floor
    for l in range(1, r["layers"] - 1):
        assert within([c[l] for c in r["cont"]], 5e-3, 50.0), (seed, l, r)
    assert within(r["exits"], 5e-3, 50.0), (seed, r)
    assert within(r["landed"], 1e-2, 2.0), (seed, r)
    assert within(r["y"], 1e-2, 2.0), (seed, r)


def _ms_prod_seeds():
    spec = os.environ.get("FUZZ_MS_PROD_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    return [4001, 4004, 20003]   # 20003: a first-layer filter nothing passes (round-4 sweep); (5 - 15 s each; 4003 and 4005 take 30 s, nearly all of it the oracle; a three-layer prob-1 scene like seed 4000 takes it six minutes)


@pytest.mark.parametrize("seed", _ms_prod_seeds())
def test_random_multi_scatter_scene_on_the_production_kernels(seed):
    check_ms_production(seed, run_ms_production_case(seed))


# ---- raypath colour on the production kernels (kModeColor): masks -> classes -> Y lanes ----------------------------------------------------
def make_color_tables(seed, sc):
    """Random colour sets for the scene of `seed` (own generator: the scene itself stays what make_case draws): every crystal entry gets 1-3
    predicates on distinct bits, then 1-4 classes over the bits in use ("any" / "all")."""
    rng = np.random.default_rng(seed + 7777)
    L0 = sc.layers[0]
    sets, bit = [], 0
    for i in range(L0.entry_count):
        faces = list(range(1, 9)) if L0.entries[i].crystal.kind == abi.CRYSTAL_PRISM else PYR_FACES
        terms = []
        for _ in range(int(rng.integers(1, 4))):
            terms.append((_term(rng, faces), "".join(c for c in "PBD" if rng.random() < 0.5), bit))
            bit += 1
        sets.append(scenes.color_set(terms))
        L0.entries[i].color_id = len(sets)
    classes = [scenes.color_class([int(b) for b in rng.choice(bit, size=int(rng.integers(1, min(bit, 3) + 1)), replace=False)], str(rng.choice(["any", "all"])))
               for _ in range(int(rng.integers(1, 5)))]
    return sets, classes


def run_color_case(seed, n=3 << 20):
    sc, rd, wl, filters, clock = make_case(seed)
    sets, classes = make_color_tables(seed, sc)
    hb = hip_backend(seed=seed, geom_clock=clock)
    ob = OracleBackend(seed=seed, threads=THREADS, acc64=1, geom_clock=clock)
    for b in (hb, ob):
        b.set_filters(filters)
        b.set_color(sets, classes)
    sh = run_session(hb, sc, rd, wl, n)
    route = hb.last_route()
    so = run_session(ob, sc, rd, wl, n)
    ih, lh = hb.ReadbackXyzAccum()
    io, lo = ob.ReadbackXyzAccum()
    lanes_h, lanes_o = hb.ReadbackClassLanes(), ob.ReadbackClassLanes()
    hb.close()
    ob.close()
    L0 = sc.layers[0]
    fixed = any(L0.entries[i].axis.latitude.type == abi.DIST_NONE and L0.entries[i].axis.azimuth.type == abi.DIST_NONE and L0.entries[i].axis.roll.type == abi.DIST_NONE
                for i in range(L0.entry_count))   # an entry whose every ray meets the crystal the same way
    return dict(exits=(sh[0].exit_count, so[0].exit_count), landed=(lh, lo), mode_mask=route.mode_mask, fixed_axes=fixed, y_image=float(np.asarray(io)[..., 1].sum(dtype=np.float64)),
                floor=oracle_pair_floor(seed, sc, rd, wl, filters, clock, n, so, np.asarray(io, np.float32), lo, sets, classes, lanes_o) if fixed else None,
                lanes=(lanes_h.sum(axis=(1, 2), dtype=np.float64), lanes_o.sum(axis=(1, 2), dtype=np.float64)),
                lane_l2=[rel_l2(block_mean(lanes_h[k][..., None], 8), block_mean(lanes_o[k][..., None], 8)) if lanes_o[k].sum() > 0 else 0.0 for k in range(len(classes))],
                # the same distances on the scale of the image's Y, of which a lane is a subset (check_color)
                lane_l2_of_image=[float(np.linalg.norm(block_mean(lanes_h[k][..., None], 8).astype(np.float64) - block_mean(lanes_o[k][..., None], 8)) /
                                        max(np.linalg.norm(block_mean(np.asarray(io)[..., 1:2], 8).astype(np.float64)), 1e-30)) for k in range(len(classes))])


def check_color(seed, r):
    assert r["mode_mask"] & abi.MODE_COLOR and not (r["mode_mask"] & (abi.MODE_CAPTURE | abi.MODE_GENERIC | abi.MODE_FILTER)), (seed, r)   # every dispatch carries masks
    fl = r["floor"] or dict(exits=0.0, landed=0.0, lanes=0.0, lane_l2=0.0)   # every ray the same way: + four times the oracle pair's own distance (check())
    assert r["exits"][0] == pytest.approx(r["exits"][1], rel=3e-4 + 4.0 * fl["exits"], abs=20), (seed, r)
    assert abs(r["landed"][0] - r["landed"][1]) <= (3e-4 + 4.0 * fl["landed"]) * max(r["landed"][1], 1.0) + 1e-3, (seed, r)
    # A lane is a subset of the image's Y: the few rays per 10^6 that take the other side of an fp32 threshold (the exits bar above) may all
    # belong to one rare class — seed 20030, fixed axes: 21 of 4.9 M exits differ, 1.9 units of Y, all of them in a class that holds 1125 of the
    # image's 1.4e6 — so the absolute part of a lane's bar is 1e-5 of the IMAGE's Y (thirty times tighter than the image's own 3e-4), not of the lane
    top = float(max(r["lanes"][1].max(), r.get("y_image", 0.0), 1.0))
    # The lanes are summed with global fp64 atomics since round 4 against the oracle's doubles here.  With fp32 atomics seed 5103
    # (a one-entry illuminant pool: 50 units of weight per exit, 3 Mi rays on 512x256, 87 % of the light in one class and most of that on the
    # sun's pixels, whose sums pass 1e7) read 3.1e-3 low and the bar was 5e-3; now 2.7e-7, and the bars are the image's
    for k in range(len(r["lanes"][1])):
        assert r["lanes"][0][k] == pytest.approx(r["lanes"][1][k], rel=3e-4 + 4.0 * fl["lanes"], abs=1e-5 * top + 1e-3), (seed, k, r)
        # (a rare class again: seed 20030's 21 exits land in two blocks of a lane that holds a thousandth of the image — 2.3 % of the lane's own
        # norm, 4e-6 of the image's; a lane within 3e-4 of the IMAGE's norm, a tenth of the image's own bar, is as close as the image is)
        assert r["lane_l2"][k] <= 3e-3 + 4.0 * fl["lane_l2"] or r["lane_l2_of_image"][k] <= 3e-4, (seed, k, r)


def _color_seeds():
    spec = os.environ.get("FUZZ_COLOR_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    return list(range(5000, 5012)) + [20030]   # 20030: the round-4 sweep's rare-class case (check_color)


@pytest.mark.parametrize("seed", _color_seeds())
def test_random_scene_with_raypath_colour_on_the_production_kernels(seed):
    check_color(seed, run_color_case(seed))


# ---- sequences of sessions on ONE backend: the accumulation planes, their layout and the routes change from session to session -------------
def run_sequence_case(seed):
    """Three to five sessions on one backend instance — wavelength source (discrete / illuminant pools of 1 .. 64 entries) and session length
    (50 k .. 9 Mi rays: direct atomics, hit log, X/Y/Z log, per-entry planes, binned routes) drawn per session, the image read back (and so
    zeroed) after some of them — against the oracle doing the same sequence.  What is compared is every readback."""
    rng = np.random.default_rng(seed + 424242)
    w, h = [(333, 211), (512, 256), (1920, 1080), (2048, 1024)][rng.integers(4)]
    lens = int(rng.choice([abi.LENS_FISHEYE_EQUAL_AREA, abi.LENS_DUAL_FISHEYE_EQUAL_AREA, abi.LENS_RECTANGULAR, abi.LENS_LINEAR]))
    rd = scenes.render(lens, w, h, fov=80.0 if lens == abi.LENS_LINEAR else 180.0, az=float(rng.uniform(0, 360)), el=float(rng.uniform(10, 80)), visible=int(rng.choice([abi.VISIBLE_UPPER, abi.VISIBLE_FULL])))
    u = lambda m, s: {"type": "uniform", "mean": m, "std": s}
    full = u(0.0, 360.0)
    e = scenes.column_crystal_entry() if rng.random() < 0.5 else scenes.entry(scenes.prism_crystal(u(1.0, 0.6), [u(1.0, 0.3)] * 6), scenes.axis(zenith=full, azimuth=full, roll=full), 1.0, 1)
    sc = scenes.scene([(0.0, [e])], max_hits=7)
    hb = hip_backend(seed=seed)
    ob = OracleBackend(seed=seed, threads=THREADS, acc64=1)
    out = []
    n_sessions = int(rng.integers(3, 6))
    n_min = 1 << 62   # the shortest session the next readback holds
    for k in range(n_sessions):
        wl = scenes.wl_discrete(float(rng.uniform(400, 700))) if rng.random() < 0.5 else scenes.wl_illuminant("D65", int(rng.choice([1, 3, 31, 64])))
        n = int(rng.choice([50_000, 1_000_000, 2 << 20, 3 << 20, 9 << 20]))
        n_min = min(n_min, n)
        sh = run_session(hb, sc, rd, wl, n)
        so = run_session(ob, sc, rd, wl, n)
        if rng.random() < 0.5 or k == n_sessions - 1:
            ih, lh = hb.ReadbackXyzAccum()
            io, lo = ob.ReadbackXyzAccum()
            io = np.asarray(io, np.float32)
            bh, bo = block_mean(ih, 8).astype(np.float64), block_mean(io, 8).astype(np.float64)
            out.append(dict(k=k, n=n, n_min=n_min, exits=(sh[0].exit_count, so[0].exit_count), landed=(lh, lo), l2=rel_l2(block_mean(ih, 8), block_mean(io, 8)) if io.sum() > 0 else 0.0,
                            moved=float(np.abs(bh - bo).sum() / max(bo.sum(), 1e-300)), sums=(ih.sum(dtype=np.float64), io.sum(dtype=np.float64))))
            n_min = 1 << 62
    hb.close()
    ob.close()
    return out


def check_sequence(seed, out):
    for r in out:
        assert r["exits"][0] == pytest.approx(r["exits"][1], rel=3e-4, abs=20), (seed, r)
        assert abs(r["landed"][0] - r["landed"][1]) <= 3e-4 * max(r["landed"][1], 1.0) + 1e-3, (seed, r)
        # An illuminant render of a few Mi rays on 2 M pixels is a few heavy exits (80 - 100 units each) per block, and one exit in ~4000
        # lands in the neighbouring pixel or row between the two sides (a one-ulp difference of the projection is 1e-4 of a pixel on a
        # 2048-wide linear lens; the same with direct atomics: tools/route_fuzz.py): percents of a per-pixel L2 distance there, whatever the
        # route.  So: the share of the energy that sits in another 8x8 block is the bar, the block-mean L2 a loose second.
        assert r["l2"] <= 3e-2, (seed, r)
        assert r["moved"] <= 4e-3, (seed, r)
        assert r["sums"][0] == pytest.approx(r["sums"][1], rel=5e-4), (seed, r)


def _seq_seeds():
    spec = os.environ.get("FUZZ_SEQ_SEEDS")
    if spec:
        a, b = spec.split(":")
        return list(range(int(a), int(b)))
    return list(range(7000, 7006))


@pytest.mark.parametrize("seed", _seq_seeds())
def test_random_sequence_of_sessions_on_one_backend(seed):
    check_sequence(seed, run_sequence_case(seed))


def test_the_two_next_face_strategies_of_the_reference_pinned_on_seed_20234():
    """The reference excludes the face a ray stands on in two ways (src/core/shared/traversal_shared.h:23-29): its CPU path by a relaxed accept
    threshold, its CUDA backend by skipping the face — equivalent on a convex body, not where an entry point sits more than 1e-5 off its face's
    plane (a fan whose corners the vertex merge moved: one of this scene's 937 sampled pyramids).  This engine follows the CUDA strategy (its
    emitted child always leaves).  Pinned on both settings of the oracle's `rehit_strategy`: against the CUDA strategy the engine matches ray
    for ray under the plain bars with NO allowance for degenerate tables; against the CPU strategy the difference is there, and it is small
    (the phantom segments of that one instance's outside reflections)."""
    cuda = run_case(20234, rehit_strategy=1)
    cpu = run_case(20234)
    assert cuda["degenerate"] > 0.0 and cuda["degenerate"] == cpu["degenerate"]          # the scene does hold such an instance
    # CUDA strategy: exact agreement in the exit count, per-ray bars without the `2 x deg` allowance of check()
    assert abs(int(cuda["exits"][0]) - int(cuda["exits"][1])) <= 2, cuda["exits"]
    frac, pix, path, left_out = cuda["cond"]
    assert frac >= 0.995 and pix >= 0.995 and path >= 0.998, cuda["cond"]
    assert abs(cuda["landed"][0] - cuda["landed"][1]) <= 3e-4 * max(cuda["landed"][1], 1.0) + 1e-3 + cuda["unmatched_weight"]
    # CPU strategy: the exit counts differ (a phantom re-hit continues as a segment instead of leaving: it emits later, or — as here, where
    # max_hits cuts it short — never), by no more than the instance's share of the rays
    extra = abs(int(cpu["exits"][1]) - int(cpu["exits"][0]))
    assert 2 < extra <= 4.0 * cpu["degenerate"] * cpu["exits"][1] + 20, (cpu["exits"], cpu["degenerate"])
    check(20234, cpu)                                                                      # ... and inside the documented allowance
    # Round 6: the ENGINE can follow the CPU strategy too (option rehit_strategy = 0: the generic kernels propagate the child that leaves
    # through the face it stands on, and a re-hit goes on as a parked segment) — SURVEY's stated ground truth, no longer only the oracle's.
    # Against the oracle on the same strategy: the exit counts agree again and the per-ray bars hold with no allowance.
    legacy = run_case(20234, hip_opts={"rehit_strategy": 0})
    assert abs(int(legacy["exits"][0]) - int(legacy["exits"][1])) <= 2, legacy["exits"]
    assert legacy["exits"][1] == cpu["exits"][1] and legacy["exits"][0] != cuda["exits"][0]      # (the engine's count moved by the phantom segments)
    frac, pix, path, left_out = legacy["cond"]
    assert frac >= 0.995 and pix >= 0.995 and path >= 0.998, legacy["cond"]
    assert abs(legacy["landed"][0] - legacy["landed"][1]) <= 3e-4 * max(legacy["landed"][1], 1.0) + 1e-3 + legacy["unmatched_weight"]


@pytest.mark.parametrize("seed", [101, 113, 129, 140])
def test_legacy_next_face_strategy_changes_nothing_on_well_formed_crystals(seed):
    """... and on crystals whose entry points lie on their faces' planes the two strategies are the same rays: the engine on the CPU strategy
    (generic kernels, re-hit test on every outgoing child) against the oracle on the CPU strategy, under the suite's ordinary bars — and the
    engine's exit count is its own count on the default (CUDA) strategy to a few rays in 1e5.  (Seed 129: + 19 of 477 641.  A child that leaves
    at grazing incidence has n.d of 1e-5 .. 1e-3 on its own face, so t = -(n.p + d) / n.d passes the +1e-5 threshold when the hit point's
    plane offset is a mere 1e-10 .. 1e-8 INSIDE — which is rounding: the oracle gives + 0 with separately rounded products and + 9 with
    contracted ones (liboracle_fma.so), the engine, which contracts, + 19.  The reference's own CPU result is not defined more finely.)"""
    legacy = run_case(seed, hip_opts={"rehit_strategy": 0})
    check(seed, legacy)
    assert abs(int(legacy["exits"][0]) - int(legacy["exits"][1])) <= 2 + 1e-4 * legacy["exits"][1], legacy["exits"]   # engine vs oracle, both on the CPU strategy
    assert abs(int(legacy["exits"][0]) - int(run_case(seed)["exits"][0])) <= 2e-4 * legacy["exits"][0] + 2
