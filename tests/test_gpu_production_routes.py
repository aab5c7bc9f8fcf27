"""GPU parity of the PRODUCTION kernel instantiations (capture off) at launch sizes where the backend's own route selection
fires: binned accumulation, per-wavelength planes, shape pools, transit-source launches.

The per-ray tests in test_gpu_parity.py set capture_exits=1, which selects the MODE=2 instantiation of halo_trace_kernel and
switches the binned route off.  Here nothing is captured: the kernels that serve BASELINE.json's configs[2] and configs[4]
(`halo_trace_kernel<0, GEOM, MONO, BIN>`) are compared with the oracle through the image, the landed weight and the
per-channel sums, and `halo_last_route` proves which instantiation / accumulation route really ran.

Tolerances (stated): one scattering layer — both sides trace the SAME rays (shared counter-based streams), so landed weight
rel 1e-4, 8x8 block-mean rel L2 <= 3e-3, per-channel sums rel 2e-4.  Two layers — the continuation order is nondeterministic
on a GPU, so layer >= 1 is compared with the reference's statistical battery (4x4 block-mean Pearson >= 0.95, sum-Y within 5 %,
test/e2e/_parity_metrics.py:27-66) tightened by the measured oracle-vs-oracle cross-seed floor: HIP-vs-oracle correlation must
reach the oracle's own seed-A-vs-seed-B correlation minus 0.02 (the battery's G3 margin), and the sum-Y / landed deviations must
stay inside 4x the cross-seed deviation (+ 2e-3).
"""
import os

import numpy as np
import pytest

from ice_halo_sim_amd import abi, scenes
from tests._oracle_backend import OracleBackend, run_session
from tests.test_gpu_parity import block_mean, hip_backend, rel_l2

pytestmark = pytest.mark.gpu

THREADS = max(8, min(os.cpu_count() or 8, 128))
FULL = {"type": "uniform", "mean": 0.0, "std": 360.0}


def _oracle_image(scene, render, wl, n, seed, **opts):
    ob = OracleBackend(seed=seed, threads=THREADS, **opts)
    st = run_session(ob, scene, render, wl, n)
    img, landed = ob.ReadbackXyzAccum()
    ob.close()
    return img, landed, st


def _stoch_pyramid_entry():
    """examples/config_example.json crystal id 5 (pyramid, upper Miller (2,0,3)) with bench_config_stoch's gauss(1, 0.15) face
    distances and its full-sphere axis: BASELINE configs[4]'s "stochastic-geometry pyramidal crystal"."""
    g = {"type": "gauss", "mean": 1.0, "std": 0.15}
    return scenes.entry(scenes.pyramid_crystal(0.1, 1.2, 0.5, upper_miller=(2, 3), face_distance=[g] * 6),
                        scenes.axis(zenith=FULL, azimuth=FULL, roll=FULL), 100.0, 5)


def _check_single_layer(hip, ora):
    ih, lh = hip
    io, lo = ora
    assert abs(lh - lo) <= 1e-4 * lo, (lh, lo)
    err = rel_l2(block_mean(ih), block_mean(io))
    assert err <= 3e-3, err
    for ch in range(3):
        assert ih[..., ch].sum(dtype=np.float64) == pytest.approx(io[..., ch].sum(dtype=np.float64), rel=2e-4)
    return err


@pytest.mark.parametrize("route", ["log_xyz", "log_xyz_overflow", "bin2"])
@pytest.mark.parametrize("pool", [31, 64])
def test_bench_config_stoch_shape_runs_the_prism_pool_production_kernels(pool, route):
    """examples/bench_config_stoch.json as the reference ships it — stochastic prism (six gauss(1, 0.15) face distances), D65
    wavelength pool (31 entries: BASELINE configs[4]'s count; 64: the reference's default pool), rectangular 2048x1024 full sky,
    max_hits 8 — at 9 Mi rays, with device-generated prism records (GEOM 2), on the two routes such a session can take:
      log_xyz  the backend's own selection (>= 8 Mi rays): X, Y, Z planes, halo_trace_kernel<0,2,false,kAccLog> logging
               {slot, CMF code, w} + halo_split_kernel<1024,16,512> + halo_log_accumulate_kernel<3> (CMF in the per-tile pass)
      log_xyz_overflow  the same with log regions and tile lists far too small (option hit_log_cap): most records take the two
               overflow fallbacks (three direct atomics with the code's CMF row)
      bin2     option lambda_planes = 1: one plane per pool entry, the two-level binned route (31 x 128 / 64 x 128 tiles > 512):
               halo_trace_kernel<0,2,true,kAccBin> + halo_split_kernel<256,16,256> + halo_bin_accumulate_range_kernel
    Compared with the oracle on the same rays."""
    sc = scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8)
    rd = scenes.render(abi.LENS_RECTANGULAR, 2048, 1024, el=0.0, visible=abi.VISIBLE_FULL)
    wl = scenes.wl_illuminant("D65", pool)
    n = 9 << 20
    hb = hip_backend(seed=23, **({"lambda_planes": 1} if route == "bin2" else {"hit_log_cap": 4096} if route.endswith("overflow") else {}))
    st = run_session(hb, sc, rd, wl, n)
    r = hb.last_route()
    crystals, orient = hb.last_sample_counts()
    hip = hb.ReadbackXyzAccum()
    hb.close()
    want = (1, 1 << 2, abi.ACCUM_LOG_XYZ, 1, 3) if route.startswith("log_xyz") else (1, 1 << 2, abi.ACCUM_BIN2, 1, pool)
    assert (r.mode_mask, r.geom_mask, r.accum_mask, r.source_mask, r.plane_cnt) == want, (r.mode_mask, r.geom_mask, r.accum_mask, r.source_mask, r.plane_cnt)
    assert r.launches == st[0].launches == 1
    assert crystals == n // 32 and orient == n           # one sampled crystal per 32 rays (simulator.hpp:144-157)
    img_o, landed_o, st_o = _oracle_image(sc, rd, wl, n, 23)
    assert st[0].exit_count == pytest.approx(st_o[0].exit_count, rel=1e-4)
    assert st[0].pixel_hits == pytest.approx(st_o[0].exit_count, rel=1e-4)     # rectangular full sky: every exit lands once
    err = _check_single_layer(hip, (img_o, landed_o))
    print("bench_config_stoch pool %d %s: block-mean rel L2 %.2e, exits/root %.3f" % (pool, route, err, st[0].exit_count / n))


@pytest.mark.parametrize("std", [0.15, 0.6])
def test_sampled_prisms_pick_their_entry_face_slab_by_slab_like_the_walk_over_triangles(std):
    """Logged launches over sampled prisms (halo_trace_kernel<0,2,.,kAccLog>): every half-wave rebuilds the slab-wise entry pick's tables
    (halo_trace.inl SlotFast) from its crystal's fan table, takes the next crystal's record a pass ahead (NextShape) and reads the
    illuminant pool from LDS.  Option pool_entry_fast = 0 walks the fan triangles as before — same uniform, same cumulative order of the
    faces, so the same entry face and point for (all but a rounding's worth of) the rays; 40 passes per workgroup, so every half-wave mostly traces
    crystals whose records came through the mirror.
    std 0.15: bench_config_stoch.json's face distances, every prism full.  std 0.6: two thirds of the prisms lose one or more side faces to their
    neighbours (7, 6, 5 faces, some none at all: their half-waves keep the walk over triangles, the others pick slab by slab in the same launch).
    Both against the oracle as well."""
    g = {"type": "gauss", "mean": 1.0, "std": std}
    e = scenes.entry(scenes.prism_crystal(1.0, [g] * 6), scenes.axis(zenith=FULL, azimuth=FULL, roll=FULL), 100.0, 1)
    sc = scenes.scene([(0.0, [e])], max_hits=8)
    rd = scenes.render(abi.LENS_RECTANGULAR, 2048, 1024, el=0.0, visible=abi.VISIBLE_FULL)   # (above 512 Ki pixels: the X/Y/Z hit log)
    wl = scenes.wl_illuminant("D65", 31)
    n = 5 << 20
    out = {}
    for fast in (1, 0):
        hb = hip_backend(seed=91, pool_entry_fast=fast, blocks_per_cu=2)
        st = run_session(hb, sc, rd, wl, n)
        r = hb.last_route()
        assert (r.mode_mask, r.geom_mask, r.accum_mask) == (1, 1 << 2, abi.ACCUM_LOG_XYZ), (r.mode_mask, r.geom_mask, r.accum_mask)
        img, landed = hb.ReadbackXyzAccum()
        hb.close()
        out[fast] = (img, landed, st[0].exit_count, st[0].pixel_hits)
    (i1, l1, e1, h1), (i0, l0, e0, h0) = out[1], out[0]
    assert abs(e1 - e0) <= 2e-6 * e0 and abs(h1 - h0) <= 2e-6 * h0, (e1, e0, h1, h0)
    assert l1 == pytest.approx(l0, rel=2e-6)
    assert rel_l2(block_mean(i1), block_mean(i0)) <= 1e-4
    img_o, landed_o, st_o = _oracle_image(sc, rd, wl, n, 91)
    assert out[1][2] == pytest.approx(st_o[0].exit_count, rel=1e-4)
    _check_single_layer((i1, l1), (img_o, landed_o))


@pytest.mark.parametrize("kind", ["prism", "pyramid"])
@pytest.mark.parametrize("shape", ["filtered", "second_layer"])
def test_pool_kernels_that_fetch_ahead_under_a_filter_and_behind_a_first_layer(kind, shape):
    """The logged shape-pool kernels (next record a pass ahead, pool entries from LDS, load waits pinned where they are issued) in the two
    places the benchmark configurations do not put them:
      filtered      a sampled crystal with an emit-gate filter on its entry: halo_trace_kernel<kModeFilter, pool, ., kAccLog>
      second_layer  a deterministic plate (prob 0.6) over the sampled crystal: the pool kernel reads its rays from the continuation pool
                    (transit source: five more loads per ray in front of the landing of the next record)
    Each against the SAME session on direct atomics (hit_log 0: the staging copy at the top of every pass, as before) — HIP against HIP, same
    rays for `filtered` (tallies equal, image to float summation order); for `second_layer` the continuation order differs from run to run
    (appends are atomic) and with it WHICH second-layer ray a continuation becomes (its streams go by its place in the pool): two runs are
    two samples of the same scene — exit count to 1e-3 (3e7 exits), landed weight to 3e-3, 32x32 block means to 2e-2 of their norm."""
    g = {"type": "gauss", "mean": 1.0, "std": 0.15}
    crystal = scenes.prism_crystal(1.0, [g] * 6) if kind == "prism" else scenes.pyramid_crystal(0.1, 1.2, 0.5, upper_miller=(2, 3), face_distance=[g] * 6)
    sampled = scenes.entry(crystal, scenes.axis(zenith=FULL, azimuth=FULL, roll=FULL), 100.0, 1, filter_id=1 if shape == "filtered" else 0)
    rd = scenes.render(abi.LENS_RECTANGULAR, 2048, 1024, el=0.0, visible=abi.VISIBLE_FULL)
    wl = scenes.wl_illuminant("D65", 31)
    if shape == "filtered":
        sc, n = scenes.scene([(0.0, [sampled])], max_hits=8), 3 << 20
        filters = [scenes.simple_filter(scenes.filter_term("entry_exit", entry=3, exit=5, min_len=2, max_len=6), "PBD")]
    else:
        plate = scenes.entry(scenes.prism_crystal(0.3), scenes.axis(zenith={"type": "gauss", "mean": 0, "std": 0.8}), 1.0, 6)
        sc, n = scenes.scene([(0.6, [plate]), (0.0, [sampled])], max_hits=7), 2 << 20
        filters = []
    out = {}
    for name, opts in (("log", {}), ("direct", {"hit_log": 0})):
        hb = hip_backend(seed=57, blocks_per_cu=2, **opts)
        if filters:
            hb.set_filters(filters)
        st = run_session(hb, sc, rd, wl, n)
        r = hb.last_route()
        img, landed = hb.ReadbackXyzAccum()
        hb.close()
        out[name] = (st, r, img, landed)
    st, r, img, landed = out["log"]
    st0, r0, img0, landed0 = out["direct"]
    geom_bit = 1 << (2 if kind == "prism" else 1)
    assert r.geom_mask & geom_bit and r.accum_mask & abi.ACCUM_LOG_XYZ, (r.geom_mask, r.accum_mask)
    assert not (r0.accum_mask & abi.ACCUM_LOG_XYZ), r0.accum_mask
    if shape == "filtered":
        assert r.mode_mask == 1 << 1, r.mode_mask
        assert (st[0].exit_count, st[0].pixel_hits) == (st0[0].exit_count, st0[0].pixel_hits)
        assert 0 < st[0].exit_count < n          # the filter lets a fraction through
        assert landed == pytest.approx(landed0, rel=1e-6)
        assert rel_l2(img, img0) <= 2e-5, rel_l2(img, img0)
    else:
        assert st[1].root_count == st[0].continuation_count >= 2 << 20, (st[1].root_count, st[0].continuation_count)   # the second layer's launch takes the log
        assert st[1].root_count == st0[1].root_count
        assert st[1].exit_count == pytest.approx(st0[1].exit_count, rel=1e-3)
        assert landed == pytest.approx(landed0, rel=3e-3)
        assert rel_l2(block_mean(img, 32), block_mean(img0, 32)) <= 2e-2, rel_l2(block_mean(img, 32), block_mean(img0, 32))


@pytest.mark.parametrize("case", ["prism_discrete", "prism_discrete_binned", "pyramid_discrete", "pyramid_discrete_direct", "pyramid_d65", "pyramid_d65_planes"])
def test_stochastic_pool_production_kernels_vs_oracle(case):
    """The other production shape-pool instantiations at sizes where they are what the backend picks:
      prism_discrete         one wavelength, stochastic prism, full sky, 4.5 Mi rays  -> <0,2,true,kAccLog>, hit log over interleaved tiles
      prism_discrete_binned  the same with option bin = 1                             -> <0,2,true,kAccBin>, one-level binned (128 tiles)
      pyramid_discrete       one wavelength, stochastic pyramid (4.1 KB records)      -> <0,1,true,kAccLog>, hit log + split + per-tile sums
      pyramid_discrete_direct  the same with option hit_log = 0                       -> <0,1,true,kAccDirect>, direct scalar plane
      pyramid_d65            D65 pool of 31, stochastic pyramid, 9 Mi rays            -> <0,1,false,kAccLog>, X/Y/Z hit log (3 planes)
      pyramid_d65_planes     the same with option lambda_planes = 1                   -> <0,1,true,kAccDirect>, one plane per entry"""
    prism = case.startswith("prism")
    sc = scenes.scene([(0.0, [scenes.stochastic_prism_entry() if prism else _stoch_pyramid_entry()])], max_hits=8)
    rd = scenes.render(abi.LENS_RECTANGULAR, 2048, 1024, el=0.0, visible=abi.VISIBLE_FULL)
    d65 = "d65" in case
    wl = scenes.wl_illuminant("D65", 31) if d65 else scenes.wl_discrete(550.0)
    n = (9 << 20) if d65 else (9 << 19)
    hb = hip_backend(seed=29, **({"hit_log": 0} if case.endswith("direct") else {"lambda_planes": 1} if case.endswith("planes") else {"bin": 1} if case.endswith("binned") else {}))
    st = run_session(hb, sc, rd, wl, n)
    route = hb.last_route()
    hip = hb.ReadbackXyzAccum()
    hb.close()
    want_geom = (1 << 2) if prism else (1 << 1)
    want_acc = {"prism_discrete": abi.ACCUM_LOG, "prism_discrete_binned": abi.ACCUM_BIN1, "pyramid_discrete": abi.ACCUM_LOG, "pyramid_d65": abi.ACCUM_LOG_XYZ}.get(case, abi.ACCUM_SCALAR)
    assert (route.mode_mask, route.geom_mask, route.accum_mask) == (1, want_geom, want_acc), (route.mode_mask, route.geom_mask, route.accum_mask)
    assert route.plane_cnt == (3 if case == "pyramid_d65" else 31 if d65 else 1)
    img_o, landed_o, st_o = _oracle_image(sc, rd, wl, n, 29)
    assert st[0].exit_count == pytest.approx(st_o[0].exit_count, rel=1e-4)
    err = _check_single_layer(hip, (img_o, landed_o))
    print("%s: block-mean rel L2 %.2e" % (case, err))


def test_bench_config_4d_discrete_31_entry_spectrum_at_its_own_launch_size():
    """bench.py --config 4d — the PRIMARY reading of BASELINE configs[4]'s "31 wavelengths" (SURVEY.md 8(d) item 5): a discrete 31-entry
    spectrum 380..780 nm weighted by the D65 SPD, one session per entry, ceil(25 M / 31) = 806 452 roots each, stochastic prism, full sky.
    The workload object is bench.py's own.  Seven of its 31 sessions (every fifth wavelength: both ends, the SPD's peak, weights from 49.98
    to 117.8) run back to back at THEIR launch size on the route the backend picks for them — sampled-prism pool (GEOM 2), scalar plane of a
    discrete wavelength, the hit log (a full-sky render takes it from 512 Ki rays: every exit lands, round 6), each session's trace kernel
    queued under the closing fold of the session before — and the image, landed weight and exit count are the oracle's for the same seeded
    rays and the same per-session weights."""
    import bench
    wk = bench.workload("4d")
    assert len(wk["wls"]) == 31 and wk["rays"] == -(-25_000_000 // 31)
    ws = [w.weight for w in wk["wls"]]
    assert wk["wls"][0].wavelength == 380.0 and wk["wls"][30].wavelength == pytest.approx(780.0) and min(ws) > 20.0 and max(ws) < 130.0   # D65 SPD, not ones
    sc, rd, n = wk["scene"], wk["render"], wk["rays"]
    pick = wk["wls"][::5]
    hb, ob = hip_backend(seed=61, **{"async": 1}), OracleBackend(seed=61, threads=THREADS, acc64=1)
    exits_h = exits_o = 0
    for wl in pick:
        run_session(hb, sc, rd, wl, n)        # queued: nothing is read between the sessions (what bench.py and the Lumice glue do)
        exits_o += run_session(ob, sc, rd, wl, n)[0].exit_count
    exits_h = hb.collect_stats().exit_count
    route = hb.last_route()
    hip, ora = hb.ReadbackXyzAccum(), ob.ReadbackXyzAccum()
    hb.close(), ob.close()
    assert (route.mode_mask, route.geom_mask, route.accum_mask) == (1, 1 << 2, abi.ACCUM_LOG), (route.mode_mask, route.geom_mask, route.accum_mask)
    assert route.plane_cnt == 1
    assert exits_h == pytest.approx(exits_o, rel=1e-4)
    err = _check_single_layer(hip, ora)
    print("config 4d, 7 of 31 sessions x %d rays: block-mean rel L2 %.2e" % (n, err))


def _pearson_blocks(a, b, k=4):
    x, y = block_mean(a, k).ravel().astype(np.float64), block_mean(b, k).ravel().astype(np.float64)
    return float(np.corrcoef(x, y)[0, 1])


CONFIG3_N, CONFIG3_WLS = 600_000, tuple(scenes.CONFIG_WAVELENGTHS_9[::4])    # 3 of the 9 wavelengths: same launch sizes, a third of the oracle's time


def oracle_config3(seed):
    """the oracle's side of the test below: 4x4 block means of its image (what the Pearson reading takes), Y sum, landed weight, first-layer
    continuation counts — a committed fixture (tests/_oracle_cache.py), rendered by tests/golden/make_oracle_render_fixtures.py"""
    def compute():
        sc, rd = scenes.config3_scene(), scenes.config2_render()
        ob = OracleBackend(seed=seed, threads=THREADS)
        cont = [run_session(ob, sc, rd, scenes.wl_discrete(w), CONFIG3_N)[0].continuation_count for w in CONFIG3_WLS]
        img, landed = ob.ReadbackXyzAccum()
        ob.close()
        return {"blocks4": block_mean(img, 4).astype(np.float32), "sum_y": np.float64(img[..., 1].sum(dtype=np.float64)), "landed": np.float64(landed),
                "cont": np.asarray(cont, np.int64)}
    from tests._oracle_cache import cached
    return cached("config3_two_layer_seed%d" % seed, compute)


def test_config3_two_layer_multi_scatter_at_production_launch_sizes():
    """BASELINE configs[2]: plate (prob 1.0) over a randomly oriented column, fisheye 1920x1080 — with 600 k roots per wavelength (three of
    the nine), so every second-layer launch reads >= 2.5 Mi continuation rays from the sharded pool through the Feistel
    gather (`source_mask` bit 1) and the production kernels run (`mode_mask` == 1).  Layer 0 traces the same rays as the oracle
    (continuation count equal to boundary flips); the second layer is compared statistically against the oracle AND against the
    oracle's own cross-seed floor (module docstring).  The oracle's two renders are committed fixtures (oracle_config3)."""
    sc = scenes.config3_scene()
    rd = scenes.config2_render()
    n = CONFIG3_N
    wls = [scenes.wl_discrete(w) for w in CONFIG3_WLS]
    hb = hip_backend(seed=42)
    cont_h, launches = [], 0
    for wl in wls:
        st = run_session(hb, sc, rd, wl, n)
        route = hb.last_route()
        assert route.mode_mask == 1 and route.source_mask == 0b011 and route.geom_mask == 1 << 3   # regular prisms: the literal-normal instantiation
        assert route.accum_mask == abi.ACCUM_NONE | abi.ACCUM_LOG   # first layer (prob 1): nothing lands, the no-accumulation kernel; >= 2.5 Mi continuations: the hit log
        assert st[1].root_count == st[0].continuation_count >= (5 << 19)
        cont_h.append(st[0].continuation_count)
        launches += st[1].launches
    ih, lh = hb.ReadbackXyzAccum()
    hb.close()
    a, b = oracle_config3(42), oracle_config3(7)
    for ch, co in zip(cont_h, a["cont"]):
        assert ch == pytest.approx(int(co), rel=3e-4)                        # layer 0: same rays on both sides
    pear = lambda x, y: float(np.corrcoef(x.ravel().astype(np.float64), y.ravel().astype(np.float64))[0, 1])
    # cross-seed floor of the oracle itself (the reference's G3 reading, test/e2e/_parity_metrics.py)
    floor_corr = pear(a["blocks4"], b["blocks4"])
    floor_y = abs(float(a["sum_y"]) / float(b["sum_y"]) - 1.0)
    floor_l = abs(float(a["landed"]) / float(b["landed"]) - 1.0)
    corr = pear(block_mean(ih, 4), a["blocks4"])
    dy = abs(ih[..., 1].sum(dtype=np.float64) / float(a["sum_y"]) - 1.0)
    dl = abs(lh / float(a["landed"]) - 1.0)
    print("configs[2] %d x %d roots: corr %.5f (oracle cross-seed %.5f), sumY dev %.2e (floor %.2e), landed dev %.2e (floor %.2e), %d layer-1 launches"
          % (len(wls), n, corr, floor_corr, dy, floor_y, dl, floor_l, launches))
    assert corr >= 0.95 and corr >= floor_corr - 0.02
    assert dy <= 0.05 and dy <= 4.0 * floor_y + 2e-3
    assert dl <= 4.0 * floor_l + 2e-3


def test_chunked_shuffle_equals_the_per_ray_shuffle_in_mean_and_variance():
    """Recombine's shuffle moves chunks of 32 continuation-pool entries (one 128-byte line per plane read) where the reference's
    CUDA backend permutes single rays (cu:1633-1657); option "shuffle_chunk" = 1 restores the latter.  In a two-layer scene with
    stochastic geometry in BOTH layers (rays of one chunk left the same sampled first-layer crystal and, moving together, meet the
    same sampled second-layer crystal) the two must give the same image in the mean and the same block-to-block variance across
    seeds: 8 seeds per mode, rectangular 512x256 full sky, 16x16 blocks."""
    e = scenes.stochastic_prism_entry()
    sc = scenes.scene([(1.0, [e]), (0.0, [e])], max_hits=6)
    rd = scenes.render(abi.LENS_RECTANGULAR, 512, 256, el=0.0, visible=abi.VISIBLE_FULL)
    n = 1 << 20
    seeds = (42, 7, 100, 200, 300, 999, 1234, 5678)          # the reference's stochastic-geometry seed set
    stack = {}
    for chunk in (32, 1):
        imgs = []
        for seed in seeds:
            hb = hip_backend(seed=seed, shuffle_chunk=chunk)
            st = run_session(hb, sc, rd, scenes.wl_discrete(550.0), n)
            assert hb.last_route().shuffle_chunk == chunk and st[1].root_count > 4 * n
            img, landed = hb.ReadbackXyzAccum()
            hb.close()
            imgs.append(block_mean(img, 16)[..., 1].astype(np.float64) * 256.0)     # block sums of Y
        stack[chunk] = np.stack(imgs)
    m32, m1 = stack[32].mean(0), stack[1].mean(0)
    v32, v1 = stack[32].var(0, ddof=1), stack[1].var(0, ddof=1)
    se = np.sqrt((v32 + v1) / len(seeds))
    z = (m32 - m1) / np.maximum(se, 1e-12)
    # mean image: block differences are noise (z-scores ~ N(0,1) over 512 blocks), totals agree to a few 1e-4
    assert abs(m32.sum() / m1.sum() - 1.0) <= 2e-3
    assert abs(z.mean()) <= 0.25 and 0.6 <= z.std() <= 1.5, (z.mean(), z.std())
    # variance: ratio of the per-block variances, averaged over the image (each ratio is an F(7,7) draw: mean 7/5, wide;
    # the log-ratio is symmetric around 0 when the variances are equal)
    lr = np.log(v32 / v1)
    print("chunk-32 vs per-ray shuffle: total ratio %.5f, z mean %.3f std %.3f, mean log variance ratio %.3f" %
          (m32.sum() / m1.sum(), z.mean(), z.std(), lr.mean()))
    assert abs(lr.mean()) <= 0.15, lr.mean()


def test_exits_drain_in_pieces():
    """halo_drain_exits copies at most `cap` records and keeps the rest pending (the reference's DrainExits hands back every
    record, trace_backend.hpp:430-448; a bounded caller drains in pieces): three partial drains return, in order, exactly what
    one full drain returns, and records captured by a later layer append behind a kept tail."""
    sc, rd = scenes.config2_scene(), scenes.config2_render(480, 270)
    n = 20_000
    one = hip_backend(seed=5, capture_exits=1)
    run_session(one, sc, rd, scenes.wl_discrete(550.0), n)
    full = one.DrainExits()
    assert len(one.DrainExits()) == 0
    one.close()
    hb = hip_backend(seed=5, capture_exits=1)
    run_session(hb, sc, rd, scenes.wl_discrete(550.0), n)
    parts = [hb.DrainExits(max_records=1000), hb.DrainExits(max_records=len(full) // 2)]
    assert len(parts[0]) == 1000 and len(parts[1]) == len(full) // 2
    run_session(hb, sc, rd, scenes.wl_discrete(550.0), 1000)         # more exits behind the kept tail
    rest = hb.DrainExits()
    hb.close()
    got = np.concatenate(parts + [rest[: len(full) - 1000 - len(full) // 2]])
    key = lambda e: np.lexsort((e["seq"], e["root"]))
    a, b = full[key(full)], got[key(got)]
    assert len(a) == len(b)
    for f in ("root", "seq", "pixel"):
        assert (a[f] == b[f]).all()
    assert (a["dir"] == b["dir"]).all() and (a["weight"] == b["weight"]).all()
    assert len(rest) > len(full) - 1000 - len(full) // 2                # the second session's records followed


def test_owned_accumulator_is_resized_after_an_external_binding():
    """own buffer at a small size, then a bound external accumulator at a larger size, then unbind and begin at the
    larger size — the owned buffer must be reallocated (its capacity decides, not the last session's width/height)."""
    import torch
    sc = scenes.config2_scene()
    small, big = scenes.config2_render(64, 32), scenes.config2_render(640, 360)
    hb = hip_backend(seed=3)
    run_session(hb, sc, small, scenes.wl_discrete(550.0), 10_000)
    hb.ReadbackXyzAccum()
    ext = torch.zeros(640 * 360 * 3 + 4, dtype=torch.float32, device="cuda")
    hb.bind_accumulator(ext.data_ptr(), ext.numel())
    run_session(hb, sc, big, scenes.wl_discrete(550.0), 50_000)
    hb.sync()
    torch.cuda.synchronize()
    ref = ext[: 640 * 360 * 3].reshape(360, 640, 3).cpu().numpy().copy()
    hb.take_landed()
    hb.bind_accumulator(0, 0)
    fresh = hip_backend(seed=3)
    run_session(fresh, sc, small, scenes.wl_discrete(550.0), 10_000)      # same counter history as hb
    fresh.ReadbackXyzAccum()
    run_session(fresh, sc, big, scenes.wl_discrete(550.0), 50_000)        # skip the rays hb spent on the bound session
    fresh.ReadbackXyzAccum()
    run_session(fresh, sc, big, scenes.wl_discrete(550.0), 50_000)
    want, _ = fresh.ReadbackXyzAccum()
    fresh.close()
    run_session(hb, sc, big, scenes.wl_discrete(550.0), 50_000)           # owned buffer, larger than it was ever allocated for
    img, _ = hb.ReadbackXyzAccum(640, 360)
    hb.close()
    assert ref.sum() > 0 and rel_l2(img, want) <= 1e-5


def test_prism_entry_pick_by_slab_equals_the_walk_over_faces():
    """Option entry_fast: one-shape dispatches of a full prism pick the entry face slab by slab (four dot products held in
    registers, branch-free walk over the seven lit-face candidates) instead of walking all faces twice.  Same uniform, same
    cumulative order, same partial sums — so the two routes must trace the same rays: identical exits but for rays within
    rounding of a triangle boundary, on a regular column AND an irregular prism (unequal face distances)."""
    ax = scenes.axis(zenith={"type": "gauss", "mean": 90, "std": 20}, azimuth=FULL, roll=FULL)
    for cr in (scenes.prism_crystal(1.3), scenes.prism_crystal(0.7, [1.0, 0.8, 1.15, 0.9, 1.1, 0.85])):
        sc = scenes.scene([(0.0, [scenes.entry(cr, ax, 1.0, 1)])], max_hits=5)
        rd = scenes.render(abi.LENS_RECTANGULAR, 512, 256, el=0.0, visible=abi.VISIBLE_FULL)
        ex = {}
        for fast in (1, 0):
            hb = hip_backend(seed=77, capture_exits=1, entry_fast=fast)
            run_session(hb, sc, rd, scenes.wl_discrete(550.0), 200_000)
            e = hb.DrainExits()
            hb.close()
            ex[fast] = e[np.lexsort((e["seq"], e["root"]))]
        a, b = ex[1], ex[0]
        assert len(a) == len(b)
        same = (a["root"] == b["root"]) & (a["seq"] == b["seq"]) & (np.abs(a["dir"] - b["dir"]).max(axis=1) <= 1e-6) & \
               (np.abs(a["weight"] - b["weight"]) <= 1e-6) & (a["path"] == b["path"]).all(axis=1)
        assert same.mean() >= 0.9995, same.mean()


def test_hit_log_route_equals_the_direct_route():
    """Option hit_log: configs[1]'s scene at 3 Mi rays — above the 2 Mi threshold, so the backend's own choice is the hit log
    (halo_trace_kernel<0,3,true,kAccLog> + halo_split_kernel + halo_bin_accumulate_range_kernel) — against the same
    session with direct atomics, and with log regions / tile lists too small for the launch, where most hits take the two
    overflow fallbacks.  The three trace the SAME rays to the SAME pixels; only the order of the float sums differs."""
    sc, rd, wl = scenes.config2_scene(), scenes.config2_render(), scenes.wl_discrete(550.0)
    n = 3 << 20
    out = {}
    for name, opts, acc in (("auto", {}, abi.ACCUM_LOG), ("direct", {"hit_log": 0}, abi.ACCUM_SCALAR), ("overflow", {"hit_log_cap": 2048}, abi.ACCUM_LOG)):
        hb = hip_backend(seed=61, **opts)
        st = run_session(hb, sc, rd, wl, n)
        route = hb.last_route()
        assert (route.mode_mask, route.accum_mask, route.geom_mask) == (1, acc, 1 << 3), (name, route.mode_mask, route.accum_mask, route.geom_mask)
        img, landed = hb.ReadbackXyzAccum()
        hb.close()
        out[name] = (img, landed, st[0].pixel_hits, st[0].exit_count)
    ref = out["direct"]
    assert ref[0].sum() > 0
    for name in ("auto", "overflow"):
        img, landed, hits, exits = out[name]
        assert (hits, exits) == (ref[2], ref[3])
        assert landed == pytest.approx(ref[1], rel=1e-6)
        # float32 sums in three different orders (8 privatised copies of atomics / fp64 tile sums / one copy of atomics): the
        # pixels of the sun's image take ~10^4 hits each and carry most of the norm — measured 1e-6 (auto) and 4e-6 (overflow)
        # (the overflow case adds most of its hits to the planes' fp64 twin since round 4 — halo_device.h DispatchParams::ovf — which the
        # closing fold takes in: no fp32 atomic sees them; it was 4e-6 ... 2e-5 from run to run, under a 5e-5 bar, while they went to ONE plane copy)
        assert rel_l2(img, ref[0]) <= 2e-5, (name, rel_l2(img, ref[0]))
        assert np.abs(img - ref[0]).max() <= 1e-4 * ref[0].max()
    # a small launch (below the threshold) takes the log only when told to
    hb = hip_backend(seed=61, hit_log=1)
    run_session(hb, sc, rd, wl, 100_000)
    assert hb.last_route().accum_mask == abi.ACCUM_LOG
    small, _ = hb.ReadbackXyzAccum()
    hb.close()
    hb = hip_backend(seed=61)
    run_session(hb, sc, rd, wl, 100_000)
    assert hb.last_route().accum_mask == abi.ACCUM_SCALAR
    want, _ = hb.ReadbackXyzAccum()
    hb.close()
    # (100 k rays: the logged launch runs the lens-specialised instantiation, which rounds a pixel coordinate differently now and
    # then — two or three hits of 160 k sit on the other side of a pixel edge, 1e-4 of the norm each)
    assert rel_l2(small, want) <= 1e-3


@pytest.mark.parametrize("case", ["discrete_weight_1000", "one_entry_d65_pool"])
def test_per_tile_sums_hold_heavy_hot_pixels(case):
    """The per-tile passes of the hit log sum in 64-bit fixed point whose scale comes with the launch (halo_kernels.hip FixQ,
    halo_backend.cpp fix_frac_bits).  Round 3 had the scale fixed at 2^32 on the assumption "weight <= 1": a slot that collects more
    than 2^31 of weight in one launch wrapped to a large negative pixel.  The scene that does it: rays of weight ~100-1000
    (a user weight, or the one entry of an illuminant pool), a point sun on a plate held still — every ray goes the same way, a handful
    of pixels take everything: 24 Mi rays of weight 1000 (192 Mi of D65's one-entry pool) put > 3e9 on the brightest.  Checked against what does not depend on the route: no negative
    pixel, the image's Y sum = landed weight (fp64 tally) x cmf_y, and the direct-atomic route's image on everything but the hot pixels
    (fp32 atomics lose the light addends there: 2.4e-7 x 4e9 is a thousand per add)."""
    plate = scenes.entry(scenes.prism_crystal(0.3), scenes.axis(), 1.0, 1)     # every axis fixed: c axis vertical
    sc = scenes.scene([(0.0, [plate])], max_hits=5, sun_altitude=35.0, sun_diameter=0.0)
    rd = scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, 512, 256, visible=abi.VISIBLE_FULL)
    wl = scenes.wl_discrete(550.0, weight=1000.0) if case == "discrete_weight_1000" else scenes.wl_illuminant("D65", 1)
    n = (24 << 20) if case == "discrete_weight_1000" else (192 << 20)
    out = {}
    for name, opts in (("log", {}), ("direct", {"hit_log": 0})):
        hb = hip_backend(seed=3, **opts)
        run_session(hb, sc, rd, wl, n)
        route = hb.last_route()
        assert bool(route.accum_mask & (abi.ACCUM_LOG | abi.ACCUM_LOG_XYZ)) == (name == "log"), (name, route.accum_mask)
        out[name] = hb.ReadbackXyzAccum()
        hb.close()
    img, landed = out["log"]
    ref, landed_ref = out["direct"]
    assert landed == pytest.approx(landed_ref, rel=1e-6) and landed > 50.0 * n      # (per-thread float partial sums in two different instantiations)
    assert img.min() >= 0.0 and np.isfinite(img).all()
    assert img.max() > 2.0 ** 31                                      # the case the fixed 32.32 scale could not hold
    import ctypes as C
    from ice_halo_sim_amd import backend as _be
    pool = np.zeros((4, 5), np.float32)
    k = _be.load_library().halo_host_wl_pool(C.byref(wl), pool.ctypes.data_as(C.POINTER(C.c_float)), 4)
    assert k == 1
    print(case, "landed", landed, "max pixel", img.max(), "sums", img.sum(axis=(0, 1), dtype=np.float64), "direct sums", ref.sum(axis=(0, 1), dtype=np.float64), "cmf", pool[0])
    # image = landed weight x CMF, channel by channel.  (1e-4, measured 2e-5: this scene's hits on a pixel all carry the SAME weight, and a
    # constant addend rounds the same way on every fp32 add of the workgroup's LDS pixel cache until the sum changes its exponent — a bias,
    # where addends that vary give noise; the wrap this test is about was a factor, not a fraction)
    for ch in range(3):
        assert float(img[..., ch].sum(dtype=np.float64)) == pytest.approx(landed * float(pool[0, 2 + ch]), rel=1e-4), ch
    cold = ref < 1e-3 * ref.max()
    assert cold.mean() > 0.99 and np.abs(img - ref)[cold].max() <= 1e-4 * max(float(ref[cold].max()), 1.0)


@pytest.mark.parametrize("lens", list(range(11)))
@pytest.mark.parametrize("visible", [abi.VISIBLE_UPPER, abi.VISIBLE_LOWER, abi.VISIBLE_FULL])
@pytest.mark.parametrize("spectrum", ["discrete", "d65"])
def test_exit_queue_kernel_lands_what_the_emit_site_kernels_land(lens, visible, spectrum):
    """The exit queue of the one-shape production kernels (MODE 0: an interaction pushes only the exits that pass the cheap
    culls of `exit_may_land`, projection and accumulation run on popped batches; scalar-plane kernels for a discrete wavelength,
    X/Y/Z kernels — the popping lane fetches the CMF row — for D65) against the kernels that project at the emit site — the capture instantiation (MODE 2, same rays, same streams) — and against the oracle, for every lens and visibility
    range: a cull that is not conservative for some lens would lose hits here.  Same pixel-hit and exit counts, same landed
    weight, same image up to the order of float sums."""
    sc = scenes.config2_scene()
    overlap = 0.0872 if lens in (4, 5, 6) else 0.0
    rd = scenes.render(lens, 512, 256, fov=120.0 if lens not in (4, 5, 6, 7, 9) else 180.0, el=30.0 if lens != 7 else 0.0, visible=visible, overlap=overlap)
    wl, n = (scenes.wl_discrete(530.0) if spectrum == "discrete" else scenes.wl_illuminant("D65", 31)), 300_000
    out = {}
    for name, kw in (("queue", {}), ("emit_site", {"capture_exits": 1})):
        hb = hip_backend(seed=19, **kw)
        st = run_session(hb, sc, rd, wl, n)
        route = hb.last_route()
        assert route.mode_mask == (1 if name == "queue" else 4)
        if kw:
            hb.DrainExits()
        img, landed = hb.ReadbackXyzAccum()
        hb.close()
        out[name] = (img, landed, st[0].pixel_hits, st[0].exit_count)
    q, e = out["queue"], out["emit_site"]
    # exits are counted before the projection: equal.  Pixel hits: two instantiations may round a coordinate on the frame's edge to
    # different sides (one hit of 1.4 M seen); a cull that is not conservative would lose thousands
    assert q[3] == e[3] and abs(q[2] - e[2]) <= 2 + 1e-5 * e[2], (q[2], e[2], q[3], e[3])
    # The two instantiations trace the same rays to rounding: a handful of 300 k exits end one pixel over (or, on the frame's edge,
    # in / out), each worth ~1e-3 of a dim image's norm — so the image is compared on 4x4 block means, the landed weight to a few
    # rays' worth.  (tools/diag_queue.py prints the per-lens figures.)
    assert q[1] == pytest.approx(e[1], rel=2e-5, abs=1e-3)
    if e[0].sum() > 0:
        assert rel_l2(block_mean(q[0], 4), block_mean(e[0], 4)) <= 1e-3
    img_o, landed_o, st_o = _oracle_image(sc, rd, wl, n, 19)
    assert q[1] == pytest.approx(landed_o, rel=2e-4, abs=1.0)
    assert q[3] == pytest.approx(st_o[0].exit_count, rel=2e-4, abs=20)
    # ... and the oracle's IMAGE: what the queue kernels put on the pixels, channel by channel (same rays: 8x8 block means to 3e-3)
    if img_o.sum() > 0:
        assert rel_l2(block_mean(q[0]), block_mean(img_o)) <= 3e-3
        tot = float(img_o.sum(dtype=np.float64))
        for ch in range(3):
            assert q[0][..., ch].sum(dtype=np.float64) == pytest.approx(img_o[..., ch].sum(dtype=np.float64), rel=3e-4, abs=1e-5 * tot)


@pytest.mark.parametrize("size", [(256, 128), (512, 256), (1024, 512)])
@pytest.mark.parametrize("visible", [abi.VISIBLE_UPPER, abi.VISIBLE_FULL])
def test_hit_log_tiles_on_small_images(size, visible):
    """The hit log keeps 128 tiles whatever the image size — on a 256x128 image that is more tiles than the plane has columns
    (TileMap's second form) — contiguous for `visible: upper`, interleaved for a full-sky render: same image as direct atomics."""
    sc, wl = scenes.config2_scene(), scenes.wl_discrete(550.0)
    rd = scenes.render(abi.LENS_FISHEYE_EQUAL_AREA, size[0], size[1], fov=180.0, el=30.0, visible=visible)
    img = {}
    for name, opts, acc in (("log", {"hit_log": 1}, abi.ACCUM_LOG), ("direct", {"hit_log": 0}, abi.ACCUM_SCALAR)):
        hb = hip_backend(seed=67, **opts)
        st = run_session(hb, sc, rd, wl, 400_000)
        assert hb.last_route().accum_mask == acc
        img[name], landed = hb.ReadbackXyzAccum()
        hb.close()
    assert img["direct"].sum() > 0 and rel_l2(img["log"], img["direct"]) <= 2e-5
    assert np.abs(img["log"] - img["direct"]).max() <= 1e-4 * img["direct"].max()


def test_xyz_hit_log_on_a_deterministic_crystal_equals_direct_xyz_atomics():
    """D65 (pool of 31) on configs[1]'s scene at 3 Mi rays: the X/Y/Z ONE-shape kernel under the hit log
    (halo_trace_kernel<0,3,false,kAccLog>: exit queue with the CMF row fetched by the popping lane, regular-prism search,
    {slot, CMF code, w} records, halo_log_accumulate_kernel<3>) against the same session on direct X/Y/Z atomics and against the
    oracle."""
    sc, rd, wl = scenes.config2_scene(), scenes.config2_render(), scenes.wl_illuminant("D65", 31)
    n = 3 << 20
    out = {}
    for name, opts, acc in (("log", {}, abi.ACCUM_LOG_XYZ), ("direct", {"hit_log": 0}, abi.ACCUM_XYZ), ("overflow", {"hit_log_cap": 2048}, abi.ACCUM_LOG_XYZ)):
        hb = hip_backend(seed=71, **opts)
        st = run_session(hb, sc, rd, wl, n)
        r = hb.last_route()
        assert (r.mode_mask, r.accum_mask, r.geom_mask, r.plane_cnt) == (1, acc, 1 << 3, 3), (name, r.mode_mask, r.accum_mask, r.geom_mask, r.plane_cnt)
        img, landed = hb.ReadbackXyzAccum()
        hb.close()
        out[name] = (img, landed, st[0].pixel_hits, st[0].exit_count)
    ref = out["direct"]
    for name in ("log", "overflow"):
        img, landed, hits, exits = out[name]
        assert (hits, exits) == (ref[2], ref[3])
        assert landed == pytest.approx(ref[1], rel=1e-6)
        assert rel_l2(img, ref[0]) <= 2e-5, (name, rel_l2(img, ref[0]))
    img_o, landed_o, st_o = _oracle_image(sc, rd, wl, n, 71)
    _check_single_layer((out["log"][0], out["log"][1]), (img_o, landed_o))


@pytest.mark.parametrize("prob", [0.0, 0.4, 1.0])
def test_last_layer_gate_in_the_production_kernels(prob):
    """A last layer with prob > 0: candidates that pass the gate are dropped ("continue" with no next layer, simulator.cpp:719-722).
    The production kernels — exit queue with direct atomics, and the last-layer hit-log kernels (the specialised instantiation for
    prob <= 0, the generic one otherwise) — against the emit-site capture kernel on the same rays."""
    sc = scenes.scene([(prob, [scenes.column_crystal_entry()])], max_hits=7)
    rd, wl, n = scenes.config2_render(), scenes.wl_discrete(550.0), 300_000
    out = {}
    for name, kw in (("direct", {}), ("log", {"hit_log": 1}), ("emit_site", {"capture_exits": 1})):
        hb = hip_backend(seed=83, **kw)
        st = run_session(hb, sc, rd, wl, n)
        if "capture_exits" in kw:
            hb.DrainExits()
        img, landed = hb.ReadbackXyzAccum()
        hb.close()
        out[name] = (img, landed, st[0].exit_count, st[0].pixel_hits)
    ref = out["emit_site"]
    if prob >= 1.0:
        assert ref[1] == 0.0 and ref[0].sum() == 0.0
    for name in ("direct", "log"):
        img, landed, exits, hits = out[name]
        assert exits == ref[2] and abs(hits - ref[3]) <= 2 + 1e-5 * ref[3], (name, exits, ref[2], hits, ref[3])
        assert landed == pytest.approx(ref[1], rel=2e-5, abs=1e-3)
        if ref[0].sum() > 0:
            assert rel_l2(block_mean(img, 4), block_mean(ref[0], 4)) <= 1e-3


def test_middle_layer_through_the_hit_log():
    """Two layers, prob 0.5 then 0: the FIRST layer both lands exits and continues rays, at 3 Mi roots — the non-final hit-log
    kernel (exit queue push + continuation append side by side) — and the second layer runs the last-layer kernel on ~7 Mi
    continuations.  Against the same scene on direct atomics: layer 0 traces the same rays (equal continuation count and exits);
    the image agrees statistically (the second layer's ray order is nondeterministic on either route)."""
    col = scenes.column_crystal_entry()
    sc = scenes.scene([(0.5, [col]), (0.0, [col])], max_hits=7)
    rd, wl, n = scenes.config2_render(), scenes.wl_discrete(550.0), 3 << 20
    out = {}
    for name, kw in (("log", {}), ("direct", {"hit_log": 0})):
        hb = hip_backend(seed=89, **kw)
        st = run_session(hb, sc, rd, wl, n)
        r = hb.last_route()
        assert r.mode_mask == 1 and r.accum_mask == (abi.ACCUM_LOG if name == "log" else abi.ACCUM_SCALAR), (name, r.accum_mask)
        img, landed = hb.ReadbackXyzAccum()
        hb.close()
        out[name] = (img, landed, st)
    a, b = out["log"], out["direct"]
    assert a[2][0].continuation_count == b[2][0].continuation_count and a[2][0].exit_count == b[2][0].exit_count
    assert a[2][1].root_count == a[2][0].continuation_count >= (2 << 20)
    assert a[1] == pytest.approx(b[1], rel=2e-3)
    assert a[0].sum(dtype=np.float64) == pytest.approx(b[0].sum(dtype=np.float64), rel=2e-3)
    assert _pearson_blocks(a[0], b[0], 16) >= 0.999


@pytest.mark.parametrize("lens", [abi.LENS_LINEAR, abi.LENS_FISHEYE_EQUAL_AREA, abi.LENS_DUAL_FISHEYE_EQUAL_AREA, abi.LENS_RECTANGULAR])
@pytest.mark.parametrize("visible", [abi.VISIBLE_UPPER, abi.VISIBLE_FULL])
def test_lens_specialised_last_layer_kernels_vs_oracle(lens, visible):
    """The instantiations that carry the lens, the visible range and the closed gate as template constants
    (`halo_trace_kernel<0,3,true,kAccLogFinal,LENS,VIS,true>`: the lenses of the reference's shipped examples) at a launch size where the
    backend picks them (3 Mi rays, prob 0 on the last layer), against the oracle: halo_last_route must report the specialisation
    (spec_mask: last layer | lens | visible range | closed gate, no generic launch), then exits, landed weight, image."""
    sc = scenes.config2_scene()
    dual = lens == abi.LENS_DUAL_FISHEYE_EQUAL_AREA
    rd = scenes.render(lens, 1024, 512, fov=180.0 if lens != abi.LENS_LINEAR else 90.0, el=30.0 if lens != abi.LENS_RECTANGULAR else 0.0, visible=visible,
                       overlap=0.0872 if dual else 0.0)
    wl, n = scenes.wl_discrete(610.0), 3 << 20
    hb = hip_backend(seed=77)
    st = run_session(hb, sc, rd, wl, n)
    r = hb.last_route()
    hip = hb.ReadbackXyzAccum()
    hb.close()
    assert r.mode_mask == abi.MODE_PLAIN and r.geom_mask == 1 << 3 and r.accum_mask == abi.ACCUM_LOG, (r.mode_mask, r.geom_mask, r.accum_mask)
    assert r.spec_mask == abi.SPEC_LAST | abi.SPEC_LENS | abi.SPEC_VIS | abi.SPEC_NOGATE and r.generic_launches == 0, (r.spec_mask, r.generic_launches)
    img_o, landed_o, st_o = _oracle_image(sc, rd, wl, n, 77, acc64=1)
    assert st[0].exit_count == pytest.approx(st_o[0].exit_count, rel=1e-4)
    err = _check_single_layer(hip, (img_o, landed_o))
    print("lens %d visible %d: spec %d, block-mean rel L2 %.2e" % (lens, visible, r.spec_mask, err))


@pytest.mark.parametrize("lens,visible", [(abi.LENS_LINEAR, abi.VISIBLE_UPPER), (abi.LENS_FISHEYE_EQUAL_AREA, abi.VISIBLE_FULL),
                                          (abi.LENS_DUAL_FISHEYE_EQUAL_AREA, abi.VISIBLE_UPPER), (abi.LENS_RECTANGULAR, abi.VISIBLE_FULL)])
def test_lens_specialised_first_layer_logging_kernels_vs_oracle(lens, visible):
    """Round 6: the logging kernels of the layers BEFORE the last take the lens and the visible range as template constants too
    (`halo_trace_kernel<0,3,true,kAccLog,LENS,VIS,false>`: their gate is open).  A two-layer scene whose first layer lets half of its outgoing
    rays land (prob 0.5), 3 Mi roots: the route info must report no generic launch (first layer: lens | visible range; last layer: all four
    bits), layer 0 must trace the oracle's rays (exit and continuation counts), and the image of both layers must agree with the oracle's
    within the second layer's shot noise."""
    col = scenes.column_crystal_entry()
    sc = scenes.scene([(0.5, [col]), (0.0, [col])], max_hits=7)
    dual = lens == abi.LENS_DUAL_FISHEYE_EQUAL_AREA
    rd = scenes.render(lens, 1024, 512, fov=180.0 if lens != abi.LENS_LINEAR else 90.0, el=30.0 if lens != abi.LENS_RECTANGULAR else 0.0, visible=visible,
                       overlap=0.0872 if dual else 0.0)
    wl, n = scenes.wl_discrete(550.0), 3 << 20
    hb = hip_backend(seed=91)
    st = run_session(hb, sc, rd, wl, n)
    r = hb.last_route()
    ih, lh = hb.ReadbackXyzAccum()
    hb.close()
    assert r.mode_mask == abi.MODE_PLAIN and r.geom_mask == 1 << 3 and r.accum_mask == abi.ACCUM_LOG, (r.mode_mask, r.geom_mask, r.accum_mask)
    assert r.spec_mask == abi.SPEC_LAST | abi.SPEC_LENS | abi.SPEC_VIS | abi.SPEC_NOGATE and r.generic_launches == 0, (r.spec_mask, r.generic_launches)
    io, lo, st_o = _oracle_image(sc, rd, wl, n, 91, acc64=1)
    assert st[0].exit_count == pytest.approx(st_o[0].exit_count, rel=3e-4) and st[0].continuation_count == pytest.approx(st_o[0].continuation_count, rel=3e-4)
    assert st[1].root_count == st[0].continuation_count
    assert lh == pytest.approx(lo, rel=2e-3), (lh, lo)   # measured 1e-4 .. 6e-4: the second layer draws its own continuation order
    err = rel_l2(block_mean(ih, 16), block_mean(io, 16))
    print("lens %d visible %d: spec %d generic %d, landed %.6g vs %.6g, 16x16 block-mean rel L2 %.2e" % (lens, visible, r.spec_mask, r.generic_launches, lh, lo, err))
    assert err <= 1e-2, err   # measured 2.2e-3 .. 3.5e-3


def test_unspecialised_lens_runs_the_generic_last_layer_kernel():
    """... and a lens outside that list (fisheye stereographic) reports the generic last-layer kernel — the route info tells the two apart."""
    sc = scenes.config2_scene()
    rd = scenes.render(abi.LENS_FISHEYE_STEREOGRAPHIC, 1024, 512, fov=180.0, el=30.0, visible=abi.VISIBLE_UPPER)
    wl, n = scenes.wl_discrete(610.0), 3 << 20
    hb = hip_backend(seed=78)
    st = run_session(hb, sc, rd, wl, n)
    r = hb.last_route()
    hip = hb.ReadbackXyzAccum()
    hb.close()
    assert r.spec_mask == abi.SPEC_LAST and r.generic_launches == 0 and r.accum_mask == abi.ACCUM_LOG, (r.spec_mask, r.generic_launches, r.accum_mask)
    img_o, landed_o, st_o = _oracle_image(sc, rd, wl, n, 78, acc64=1)
    _check_single_layer(hip, (img_o, landed_o))


@pytest.mark.parametrize("spectrum", ["discrete", "d65"])
def test_panorama_of_two_to_the_25_pixels(spectrum):
    """An 8192x4096 session — 2^25 pixels, what the workgroup cache's key names on its own (halo_begin; the reference has no cap,
    render.cpp takes any resolution) — against the oracle on the same rays: landed weight rel 1e-4, 64x64 block means rel L2 <= 3e-3,
    channel sums rel 2e-4; the pixels above 2^23 (the lower three quarters of the image) must carry their share.  One pixel more is
    HALO_UNAVAILABLE, which the glue turns into the reference's own CPU path for that Run."""
    from ice_halo_sim_amd.backend import BackendUnavailableError
    sc = scenes.config2_scene()
    rd = scenes.render(abi.LENS_RECTANGULAR, 8192, 4096, fov=360.0, el=0.0, visible=abi.VISIBLE_FULL)
    wl = scenes.wl_discrete(550.0) if spectrum == "discrete" else scenes.wl_illuminant("D65", 64)
    n = 8 << 20   # where an illuminant session on a smaller image would take a plane per pool entry
    hb = hip_backend(seed=67)
    st = run_session(hb, sc, rd, wl, n)
    img, landed = hb.ReadbackXyzAccum()
    hb.close()
    oimg, olanded, ost = _oracle_image(sc, rd, wl, n, 67)
    # (a few rays per 10^7 take the other side of a total-reflection threshold in fp32: DESIGN.md section 4)
    assert st[0].exit_count == pytest.approx(ost[0].exit_count, rel=1e-5) and st[0].pixel_hits == pytest.approx(ost[0].pixel_hits, rel=1e-5)
    assert landed == pytest.approx(olanded, rel=1e-4)
    assert img.shape == oimg.shape == (4096, 8192, 3)
    lower = img[1024:].astype(np.float64).sum()
    assert lower > 0.05 * img.astype(np.float64).sum()
    assert lower == pytest.approx(oimg[1024:].astype(np.float64).sum(), rel=2e-4)
    assert rel_l2(block_mean(img, 64), block_mean(oimg, 64)) <= 3e-3
    for c in range(3):
        assert img[..., c].astype(np.float64).sum() == pytest.approx(oimg[..., c].astype(np.float64).sum(), rel=2e-4)
    hb = hip_backend(seed=67)
    with pytest.raises(BackendUnavailableError):
        hb.BeginSession(sc, scenes.render(abi.LENS_RECTANGULAR, 8192, 4097, fov=360.0, el=0.0, visible=abi.VISIBLE_FULL), wl, n)
    hb.close()


def test_many_small_sessions_fold_once_and_equal_the_eager_fold():
    """A server that sends a wavelength's rays as many small sessions (Lumice's CUDA-route dispatch: 2^18 rays per session, server.cpp:151): with
    the backend's own accumulator the closing fold waits for the first reader (option lazy_fold, the default) — the planes of equal sessions
    add up, a session with OTHER planes (another wavelength, another image size) folds what is pending first.  Whatever the order of folds,
    the image is the one the eager fold gives (same rays; only the float summation order differs)."""
    from ice_halo_sim_amd.backend import HipTraceBackend
    from tests._oracle_backend import run_session
    sc = scenes.config2_scene()
    wls = [scenes.wl_discrete(w) for w in (450.0, 450.0, 450.0, 610.0, 610.0, 450.0)]
    sizes = [(480, 270)] * 5 + [(256, 128)]
    out = {}
    for lazy in (1, 0):
        hb = HipTraceBackend(device=0, seed=11, lazy_fold=lazy)
        imgs = []
        for k, (wl, (w, h)) in enumerate(zip(wls, sizes)):
            rd = scenes.config2_render(w, h)
            if k == 5:
                imgs.append(hb.ReadbackXyzAccum(480, 270))     # the 480x270 image, before the size changes
            run_session(hb, sc, rd, wl, 1 << 18)
        imgs.append(hb.ReadbackXyzAccum(256, 128))
        out[lazy] = imgs
        hb.close()
    for (a, la), (b, lb) in zip(out[1], out[0]):
        assert la == pytest.approx(lb, rel=1e-12)
        assert a.sum(dtype=np.float64) == pytest.approx(b.sum(dtype=np.float64), rel=2e-6)
        assert np.abs(a - b).max() <= 2e-5 * max(float(b.max()), 1e-30)
    # the colours of the two wavelengths are both there (a fold with the wrong CMF would tint everything one way)
    big = out[1][0][0]
    assert big[..., 2].sum() > 0.2 * big[..., 1].sum() and big[..., 0].sum() > 0.2 * big[..., 1].sum()


def _stats_tuple(st):
    return int(st.root_count), int(st.exit_count), int(st.pixel_hits)


def test_equal_small_sessions_equal_one_large_session():
    """64 queued sessions of 2^15 rays — what a Lumice server at Metal's dispatch size would send (server.cpp:140) — are the SAME rays as one
    session of 2^21 (monotone ray counters), so exits and pixel hits agree exactly and the image to float summation order, whichever way
    the tables reach the device (cached on the device from the first session on, or uploaded with every dispatch) and however the tallies
    are collected (cumulative device counters, read once at the end)."""
    from ice_halo_sim_amd.backend import HipTraceBackend
    from tests._oracle_backend import run_session
    sc, rd, wl = scenes.config2_scene(), scenes.config2_render(480, 270), scenes.wl_discrete(550.0)
    res = {}
    for name, opts, sizes in (("small", {"async": 1}, [1 << 15] * 64), ("small_uploaded", {"async": 1, "table_cache": 0}, [1 << 15] * 64),
                              ("small_sync", {}, [1 << 15] * 64), ("large", {}, [1 << 21])):
        hb = HipTraceBackend(device=0, seed=5, **opts)
        hb.collect_stats()
        for n in sizes:
            run_session(hb, sc, rd, wl, n)
        st = hb.collect_stats()
        img, landed = hb.ReadbackXyzAccum(480, 270)
        res[name] = (_stats_tuple(st), landed, img)
        hb.close()
    ref_st, ref_landed, ref_img = res["large"]
    assert ref_st[0] == 1 << 21 and ref_st[1] > 4 * ref_st[0] and ref_st[2] > ref_st[0]
    for name in ("small", "small_uploaded", "small_sync"):
        st, landed, img = res[name]
        assert st == ref_st, (name, st, ref_st)
        assert landed == pytest.approx(ref_landed, rel=1e-6)
        assert np.abs(img - ref_img).max() <= 2e-5 * float(ref_img.max())
        assert img.sum(dtype=np.float64) == pytest.approx(ref_img.sum(dtype=np.float64), rel=2e-6)


def test_mixed_session_sizes_on_two_streams_equal_one_stream():
    """Sessions of mixed sizes queued back to back take the two trace streams in turn (<= 2^25 rays), and their routes differ in HOW they add to
    the planes: logged launches (>= 2^21 rays) end in plain read-modify-writes by their per-tile sums, smaller ones add with atomics from the
    trace kernel.  Events keep a plain write-out from running beside any other writer of the same planes (halo_backend.cpp: ev_gate / ev_rmw).
    What this test can show is that the two-stream run of such a mix gives the one-stream run's image pixel by pixel (a lost update would be
    a pixel missing a whole tile sum, tens of per cent; float-order differences are ~1e-5 of a pixel) and its tallies exactly.  It is NOT a race
    detector: a build without the gates (-DHALO_NO_PLANE_GATES) passes it too, the window being a few nanoseconds per slot — the ordering
    holds by construction, not by this test."""
    from ice_halo_sim_amd.backend import HipTraceBackend
    from tests._oracle_backend import run_session
    sc, rd, wl = scenes.config2_scene(), scenes.config2_render(640, 360), scenes.wl_discrete(550.0)
    sizes = [1 << 22, 1 << 20, 1 << 16, 1 << 20, 1 << 16, 1 << 20, 1 << 22, 1 << 21, 1 << 20, 1 << 23, 1 << 18, 1 << 22, 1 << 21, 1 << 21, 1 << 19, 1 << 22]
    res = {}
    for name, opts in (("two_streams", {}), ("one_stream", {"overlap": 0})):
        hb = HipTraceBackend(device=0, seed=29, **{"async": 1}, **opts)
        hb.collect_stats()
        for rep in range(6):
            for n in sizes:
                run_session(hb, sc, rd, wl, n)
        st = hb.collect_stats()
        img, landed = hb.ReadbackXyzAccum(640, 360)
        res[name] = (_stats_tuple(st), landed, img.astype(np.float64))
        hb.close()
    (st2, l2, a), (st1, l1, b) = res["two_streams"], res["one_stream"]
    assert st2 == st1 and st1[0] == 6 * sum(sizes)
    assert l2 == pytest.approx(l1, rel=1e-6)
    worst = np.abs(a - b) / (np.maximum(a, b) + 1e-9 * b.max())
    assert worst.max() <= 1e-3, (worst.max(), np.unravel_index(worst.argmax(), worst.shape))


def test_sampled_crystal_sessions_under_every_scheduling_option():
    """Queued sessions of sampled crystals (device generator -> pool -> trace kernel -> passes) under the scheduling options of round 5: one
    stream (overlap 0), the default, and the generator of a chip-filling launch queued beside the previous launch's kernels on the other
    pool (gen_ahead 1).  Scheduling moves no ray: same tallies, same landed weight, same image to float summation order.  Launches of
    2^22 rays (pool 0, stream 0 by default) and of 2^20 (alternating streams and pools), prisms (hit log on X/Y/Z planes) and pyramids."""
    from ice_halo_sim_amd.backend import HipTraceBackend
    from tests._oracle_backend import run_session
    rd = scenes.config2_render(480, 270)
    for entry_of, wl in ((scenes.stochastic_prism_entry, scenes.wl_illuminant("D65", 31)), (_stoch_pyramid_entry, scenes.wl_discrete(550.0))):
        sc = scenes.scene([(0.0, [entry_of()])], max_hits=8)
        res = {}
        for name, opts in (("default", {}), ("one_stream", {"overlap": 0}), ("gen_ahead", {"gen_ahead": 1})):
            hb = HipTraceBackend(device=0, seed=17, **{"async": 1}, **opts)
            hb.collect_stats()
            for n in (1 << 22, 1 << 22, 1 << 20, 1 << 20, 1 << 22):
                run_session(hb, sc, rd, wl, n)
            st = hb.collect_stats()
            img, landed = hb.ReadbackXyzAccum(480, 270)
            res[name] = (_stats_tuple(st), landed, img)
            hb.close()
        ref_st, ref_landed, ref_img = res["one_stream"]
        assert ref_st[0] == 3 * (1 << 22) + 2 * (1 << 20) and ref_st[2] > ref_st[0]
        for name in ("default", "gen_ahead"):
            st, landed, img = res[name]
            assert st == ref_st, (name, st, ref_st)
            assert landed == pytest.approx(ref_landed, rel=1e-6)
            assert np.abs(img - ref_img).max() <= 2e-5 * float(ref_img.max())


def test_table_cache_follows_scene_wavelength_filters_and_options():
    """The device-resident tables of a crystal entry (round 5) must be replaced whenever anything they were built from changes: the scene
    (another crystal, another axis distribution), the wavelength, the filter table, an option.  A backend with the cache and one that
    uploads its tables with every dispatch trace the same alternating sequence of sessions: same tallies after every session, same image."""
    from ice_halo_sim_amd.backend import HipTraceBackend
    from tests._oracle_backend import run_session
    T = scenes.filter_term
    full = {"type": "uniform", "mean": 0.0, "std": 360.0}
    col = scenes.config2_scene()
    plate = scenes.scene([(0.0, [scenes.entry(scenes.prism_crystal(0.3), scenes.axis(zenith={"type": "gauss", "mean": 0.0, "std": 0.8}, azimuth=full, roll=full), 1.0, 6)])], max_hits=7)
    flt_entry = scenes.column_crystal_entry()
    flt_entry.filter_id = 1
    filtered = scenes.scene([(0.0, [flt_entry])], max_hits=7)
    rd = scenes.config2_render(320, 180)
    f_a = [scenes.simple_filter(T("raypath", raypath=[3, 5]), "P")]
    f_b = [scenes.simple_filter(T("entry_exit", entry=1, exit=3, min_len=2, max_len=5), "PBD")]
    steps = [("scene", col, 550.0), ("scene", col, 550.0), ("scene", plate, 550.0), ("scene", col, 550.0), ("scene", col, 610.0), ("scene", col, 610.0),
             ("filters", f_a, None), ("scene", filtered, 610.0), ("scene", filtered, 610.0), ("filters", f_b, None), ("scene", filtered, 610.0),
             ("option", ("entry_fast", 0), None), ("scene", col, 610.0), ("option", ("entry_fast", 1), None), ("scene", col, 610.0)]
    runs = {}
    for cache in (1, 0):
        hb = HipTraceBackend(device=0, seed=9, table_cache=cache)
        tallies = []
        for kind, what, w in steps:
            if kind == "filters":
                hb.set_filters(what)
            elif kind == "option":
                hb.set_option(*what)
            else:
                st = run_session(hb, what, rd, scenes.wl_discrete(w), 1 << 16)[0]
                tallies.append((int(st.exit_count), int(st.pixel_hits), round(float(st.exit_w_sum), 3)))
        runs[cache] = (tallies, hb.ReadbackXyzAccum(320, 180))
        hb.close()
    assert runs[1][0] == runs[0][0]
    # (a stale table would show here: the two runs trace the same rays session by session, so a session that found its predecessor's
    # crystal, wavelength or filter on the device has other tallies than the one that uploaded its own)
    (img1, l1), (img0, l0) = runs[1][1], runs[0][1]
    assert l1 == pytest.approx(l0, rel=1e-6)
    assert np.abs(img1 - img0).max() <= 2e-5 * float(img0.max())


def test_async_sessions_alternating_wavelengths_and_crystals_keep_their_own_tables():
    """Back-to-back single-layer sessions that nobody reads between (option async: what the Lumice glue does, simulator.cpp:1099 loops over
    the wavelengths with no readback) alternate wavelength and crystal: every session misses the table cache and uploads its tables while the
    kernel of the session before may still be staging ITS tables on a trace stream.  The two slots per cache entry and their reader events
    (HaloBackend::TableCacheEntry) keep them apart: image and landed weight equal a run with no cache, one stream and synchronous sessions.
    Injected rays share the hazard (their device buffers are re-filled per session): the same comparison with host rays."""
    from ice_halo_sim_amd.backend import HipTraceBackend
    from tests._oracle_backend import run_session
    full = {"type": "uniform", "mean": 0.0, "std": 360.0}
    col = scenes.config2_scene()
    plate = scenes.scene([(0.0, [scenes.entry(scenes.prism_crystal(0.3), scenes.axis(zenith={"type": "gauss", "mean": 0.0, "std": 0.8}, azimuth=full, roll=full), 1.0, 6)])], max_hits=7)
    rd = scenes.config2_render(480, 270)
    seq = [(col, 450.0), (plate, 770.0), (col, 610.0), (plate, 450.0), (col, 770.0), (col, 450.0), (plate, 610.0), (col, 530.0)] * 6
    runs = {}
    for tag, opts in (("queued", dict(async_=1, overlap=1, table_cache=1)), ("plain", dict(async_=0, overlap=0, table_cache=0))):
        hb = HipTraceBackend(device=0, seed=31, **{k.rstrip("_"): v for k, v in opts.items()})
        for sc, w in seq:
            run_session(hb, sc, rd, scenes.wl_discrete(w), 1 << 19)
        runs[tag] = hb.ReadbackXyzAccum(480, 270)
        hb.close()
    (img1, l1), (img0, l0) = runs["queued"], runs["plain"]
    assert l1 == pytest.approx(l0, rel=1e-6)
    # a session traced with its neighbour's wavelength puts its refracted light elsewhere: far above the float-order noise of equal rays
    assert np.abs(img1 - img0).max() <= 2e-5 * float(img0.max())
    # host-injected rays: per-session device buffers, re-filled while the session before may still be reading them
    runs = {}
    for tag, opts in (("queued", dict(async_=1, overlap=1)), ("plain", dict(async_=0, overlap=0))):
        hb = HipTraceBackend(device=0, seed=31, **{k.rstrip("_"): v for k, v in opts.items()})
        g = np.random.default_rng(5)
        for k in range(12):
            n = 20000 + 4000 * (k % 3)
            d = g.normal(size=(n, 3)).astype(np.float32)
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            d[:, 2] = -np.abs(d[:, 2]) - 0.2
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            p = np.zeros((n, 3), np.float32)
            p[:, 2] = 0.65   # on the top basal face (h = 1.3)
            p[:, :2] = g.uniform(-0.3, 0.3, size=(n, 2)).astype(np.float32)
            w = np.full(n, 1.0 + k, np.float32)
            hb.BeginSession(col, rd, scenes.wl_discrete(450.0 + 40.0 * (k % 8)), n)
            hb.TraceLayer(n, host_rays=(d, p, w, np.zeros(n, np.uint32)))
            hb.EndSession()
        runs[tag] = hb.ReadbackXyzAccum(480, 270)
        hb.close()
    (img1, l1), (img0, l0) = runs["queued"], runs["plain"]
    assert l0 > 0 and l1 == pytest.approx(l0, rel=1e-6)
    assert np.abs(img1 - img0).max() <= 2e-5 * float(img0.max())


def test_short_logged_sessions_under_the_previous_fold_equal_the_one_stream_order():
    """Round 6: on a full-sky render a scalar session takes the hit log from 512 Ki rays, its trace kernel starts under the closing fold of the
    session before, and the passes alternate between two plane sets (HaloBackend::mono_two) — the fold of session k zeroes one set while the
    passes of session k + 1 add to the other.  A queued sequence that exercises every edge of that — new wavelength each session, the same
    wavelength twice (no fold between: the set stays), a short session under the log's threshold in the middle (direct atomics into the
    current set), a readback half way, another image size — gives the image of the same sequence on ONE stream with synchronous sessions,
    and every logged session reports the hit-log route."""
    from ice_halo_sim_amd.backend import HipTraceBackend
    from tests._oracle_backend import run_session
    sc = scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8)
    big, small = scenes.render(abi.LENS_RECTANGULAR, 2048, 1024, el=0.0, visible=abi.VISIBLE_FULL), scenes.render(abi.LENS_RECTANGULAR, 1024, 512, el=0.0, visible=abi.VISIBLE_FULL)
    n = 600_000
    seq = [(big, 400.0, n), (big, 450.0, n), (big, 450.0, n), (big, 500.0, n), (big, 550.0, 100_000), (big, 600.0, n), (big, 650.0, n), ("read", None, None),
           (big, 700.0, n), (big, 720.0, n), (small, 500.0, n), (small, 520.0, n), (big, 540.0, n), (big, 560.0, n)]
    runs = {}
    for tag, opts in (("queued", {"async": 1, "overlap": 1}), ("plain", {"async": 0, "overlap": 0})):
        hb = HipTraceBackend(device=0, seed=77, **opts)
        imgs, routes = [], []
        cur = None
        for rd, w, m in seq:
            if rd == "read":
                imgs.append(hb.ReadbackXyzAccum(cur.width, cur.height))
                continue
            if cur is not None and (rd.width, rd.height) != (cur.width, cur.height):
                imgs.append(hb.ReadbackXyzAccum(cur.width, cur.height))     # the image changes size: read the old one out first
            cur = rd
            run_session(hb, sc, rd, scenes.wl_discrete(w, 1.0 + w / 500.0), m)
            routes.append((m, hb.last_route().accum_mask))
        imgs.append(hb.ReadbackXyzAccum(cur.width, cur.height))
        runs[tag] = (imgs, routes)
        hb.close()
    for m, mask in runs["queued"][1]:
        assert mask & (abi.ACCUM_LOG if m >= (1 << 19) else abi.ACCUM_SCALAR), (m, mask)
    assert len(runs["queued"][0]) == len(runs["plain"][0]) == 4
    for (iq, lq), (ip, lp) in zip(runs["queued"][0], runs["plain"][0]):
        assert lp > 0 and lq == pytest.approx(lp, rel=1e-6)
        assert np.abs(iq - ip).max() <= 2e-5 * float(ip.max())


def test_options_that_shape_an_open_session_are_refused_inside_it():
    """seed, ray_base and rank move the monotone ray counters, capture_exits / filter_fast / mono_copies the plane layout decided at
    BeginSession: changing any of them between the layers of a session would replay ray indices the session has already consumed or send
    hits to a layout made for another route.  They are refused while a session is open and accepted again after EndSession."""
    from ice_halo_sim_amd.backend import BackendError, HipTraceBackend
    hb = HipTraceBackend(device=0, seed=3)
    sc, rd = scenes.config3_scene(), scenes.config2_render(160, 90)
    hb.BeginSession(sc, rd, scenes.wl_discrete(550.0), 1 << 14)
    hb.TraceLayer(1 << 14)
    for key, val in (("seed", 5), ("ray_base", 1 << 33), ("rank", 2), ("capture_exits", 1), ("filter_fast", 0), ("mono_copies", 4)):
        with pytest.raises(BackendError):
            hb.set_option(key, val)
    hb.Recombine(True)
    st = hb.TraceLayer(0)
    hb.EndSession()
    assert st.root_count > 0
    for key, val in (("seed", 5), ("ray_base", 1 << 33), ("rank", 2), ("capture_exits", 0), ("filter_fast", 1), ("mono_copies", 8)):
        hb.set_option(key, val)
    hb.close()
