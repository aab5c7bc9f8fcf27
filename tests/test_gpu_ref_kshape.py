"""The reference's K-shape-pool structural tests, on this engine's shape pools (reference files under /root/reference/test):
  unit-correctness/backend/test_cuda_kshape_pool_locality.cpp   Test A path[] holds LOCAL faces of the shape a ray was traced on; Test B the
                                                                 second layer of a multi-scatter batch spreads its rays over several pool shapes;
                                                                 Test C what the pool holds when the clock is left alone
  parity-cross-backend/backend/test_cuda_rich_exit.cpp:202-366  CountsStochasticCrystalDrawsAcrossLayers
The reference reaches into its CUDA backend with test hooks (pool_shape_table, d_root_pool_shape_); this engine has no such tables — a ray's shape
is record `ray index / geom_clock` of the launch's pool, by construction — so each invariant is checked on what crosses the seam: exit records
(captured paths, per root), sample counts and continuation counts, against the host builder's shapes for the same stream indices."""

import numpy as np
import pytest

from ice_halo_sim_amd import abi, scenes
from tests._oracle_backend import OracleBackend, run_session
from tests.test_gpu_parity import hip_backend

pytestmark = pytest.mark.gpu

G = {"type": "gauss", "mean": 1.0, "std": 0.15}
FULL = {"type": "uniform", "mean": 0.0, "std": 360.0}


def _stoch_entry(cid=0, axis=None):
    return scenes.entry(scenes.prism_crystal(G, [G] * 6), axis or scenes.axis(zenith=FULL, azimuth=FULL, roll=FULL), 1.0, cid)


def test_paths_hold_local_faces_of_the_shape_the_ray_was_traced_on():
    """Test A (KShapePool_PathIsLocalWithinPolygonFaceCount_AC1, :100-200): clock 8, 64 rays -> 8 pool shapes; every byte of every captured
    path is a face NUMBER of a hexagonal prism (1..8) — never an index into the pool — rays on the later shapes included; and the 64 rays
    did meet more than one shape (the exits of roots 8k .. 8k + 7 are the oracle's for shape k, which differ from shape 0's)."""
    sc = scenes.scene([(0.0, [_stoch_entry()])], max_hits=6, sun_altitude=30.0)
    rd = scenes.render(abi.LENS_FISHEYE_EQUAL_AREA, 64, 64, fov=180.0, el=90.0)
    hb, ob = hip_backend(seed=31, capture_exits=1, geom_clock=8), OracleBackend(seed=31, capture_exits=1, threads=1, geom_clock=8)
    for b in (hb, ob):
        run_session(b, sc, rd, scenes.wl_discrete(550.0), 64)
    crystals, _ = hb.last_sample_counts()
    eh, eo = hb.DrainExits(), ob.DrainExits()
    hb.close(), ob.close()
    assert crystals == 8                                                   # ceil(64 / 8) pool shapes (P_ci, :111-113)
    assert len(eh) == len(eo) > 64
    for rec in eh:
        path = np.asarray(rec["path"])[: int(rec["path_len"])]
        assert len(path) >= 1 and ((path >= 1) & (path <= 8)).all(), path   # local face numbers, whatever shape the ray was on
    assert (eh["root"] >= 8).sum() > 0                                      # rays past the first shape exist and were recorded
    # per root the same exits as the oracle, whose shape k is the host builder's record k: a picker collapsed onto slot 0 would fail here
    key = lambda e: np.lexsort((e["seq"], e["root"]))
    a, b = eh[key(eh)], eo[key(eo)]
    assert np.array_equal(a["root"], b["root"]) and np.array_equal(a["seq"], b["seq"])
    assert np.abs(np.asarray(a["dir"], np.float64) - np.asarray(b["dir"], np.float64)).max() <= 2e-5


def test_second_layer_spreads_its_rays_over_several_pool_shapes():
    """Test B (KShapePool_TransitPicksMultipleShapes_AC1, :202-283): two layers of stochastic prisms, clock 8, 4096 roots — the continuation rays
    that reach layer 1 are dealt over ceil(continuations / 8) freshly sampled shapes (not one), and CountsStochasticCrystalDrawsAcrossLayers
    (test_cuda_rich_exit.cpp:202-366): the session's crystal count is the sum of both layers' draws."""
    sc = scenes.scene([(0.6, [_stoch_entry(0)]), (0.0, [_stoch_entry(1)])], max_hits=6, sun_altitude=30.0)
    rd = scenes.render(abi.LENS_RECTANGULAR, 128, 64, el=0.0, visible=abi.VISIBLE_FULL)
    hb = hip_backend(seed=7, geom_clock=8)
    st = run_session(hb, sc, rd, scenes.wl_discrete(550.0), 4096)
    crystals, orients = hb.last_sample_counts()
    hb.close()
    cont = int(st[0].continuation_count)
    assert cont > 512 and int(st[1].root_count) == cont
    assert crystals == (4096 + 7) // 8 + (cont + 7) // 8                    # both layers draw: 512 shapes + one per 8 continuation rays (>= 64: "several")
    assert orients == 4096 + cont


def test_pool_without_a_clock_setting_follows_the_cpu_paths_32_rays_per_shape():
    """Test C (KShapePool_DefaultKnobUnsetGivesPCiOne_AC2, :285-340) pins what the reference's GPU backends do when LUMICE_GPU_GEOM_CLOCK is
    unset: ONE pool shape per (layer, crystal entry) and batch.  This engine's default is deliberately the legacy CPU path's instead — a new
    shape every 32 rays (simulator.cpp:1244-1275, LUMICE_GEOM_CLOCK's default) — and the reference's knob-off behaviour is
    one option away: geom_clock >= the batch.  Both are pinned: the crystal count of a 4096-ray batch is 128 untouched, 1 with the clock at
    the batch size (and the single shape is then the stream's first: every root sees shape 0)."""
    sc = scenes.scene([(0.0, [_stoch_entry()])], max_hits=6, sun_altitude=30.0)
    rd = scenes.render(abi.LENS_RECTANGULAR, 128, 64, el=0.0, visible=abi.VISIBLE_FULL)
    hb = hip_backend(seed=5)
    run_session(hb, sc, rd, scenes.wl_discrete(550.0), 4096)
    assert hb.last_sample_counts()[0] == 4096 // 32
    hb.set_option("geom_clock", 4096)
    run_session(hb, sc, rd, scenes.wl_discrete(550.0), 4096)
    assert hb.last_sample_counts()[0] == 1
    hb.close()


def test_ray_base_high_word_reaches_the_gen_gate_and_transit_streams():
    """test_cuda_rich_exit.cpp:368-607 (CudaRngHiWiring: GenStreamWireUp, GateStreamWireUp, GateMsMode1StreamWireUp, TransitStreamWireUp): the
    high word of the 64-bit ray base (SplitPcgRayBase, trace_backend.hpp:184; epochs 0, 0, 2^32, 2 x 2^32) must reach every stream's seed.
    The reference needs test hooks to look at each stream's output; here all three streams show in what crosses the seam, and the oracle
    takes the same base (`ray_base`), so each epoch is checked twice: against the other epochs and against the ORACLE at that epoch — a
    stream that ignored its high word would reproduce epoch 0 and fail the second comparison at epochs 1 and 2.
    Scene A (random axes, two layers, prob 0.6): layer 0's exits are the oracle's root for root (gen stream) and the continuation count is
    the oracle's (gate stream); hi == 0 twice is bit-identical, every other pair of epochs moves > 90 % of the entry reflections and
    changes which exits continue.  Scene B, the reference's trick (cuda_test_helpers.hpp:163-182) taken one step further so that the pool's
    ORDER cannot matter: sun at the zenith with no disc over a plate on a fixed vertical axis and max_hits 1 — every continuation ray is the
    same ray, so the second layer's exits, as a sorted set, are a function of the transit stream alone: identical at hi == 0 twice,
    different between epochs, and the oracle's at every epoch."""
    full = {"type": "uniform", "mean": 0.0, "std": 360.0}
    rd = scenes.render(abi.LENS_RECTANGULAR, 128, 64, el=0.0, visible=abi.VISIBLE_FULL)
    n = 4096
    wl = scenes.wl_discrete(550.0)
    e0 = scenes.entry(scenes.prism_crystal(1.0), scenes.axis(zenith=full, azimuth=full, roll=full), 1.0, 0)
    e1 = scenes.entry(scenes.prism_crystal(1.3), scenes.axis(zenith=full, azimuth=full, roll=full), 1.0, 1)
    sc_a = scenes.scene([(0.6, [e0]), (0.0, [e1])], max_hits=8, sun_altitude=30.0)
    plate = scenes.entry(scenes.prism_crystal(0.3), scenes.axis(), 1.0, 0)      # default axis: c vertical, nothing random
    sc_b = scenes.scene([(0.6, [plate]), (0.0, [e1])], max_hits=1, sun_altitude=90.0, sun_diameter=0.0)
    key = lambda e: np.lexsort((e["seq"], e["root"]))

    def sorted_dirs(e):
        d = np.asarray(e["dir"], np.float64).reshape(-1, 3)
        return d[np.lexsort((d[:, 2], d[:, 1], d[:, 0]))]
    got = []
    for base in (0, 0, 1 << 32, 2 << 32):
        out = {}
        for tag, sc in (("a", sc_a), ("b", sc_b)):
            hb, ob = hip_backend(seed=42, capture_exits=1), OracleBackend(seed=42, capture_exits=1, threads=1)
            for b in (hb, ob):
                b.set_option("ray_base", base)
            sh, so = run_session(hb, sc, rd, wl, n), run_session(ob, sc, rd, wl, n)
            eh, eo = hb.DrainExits(), ob.DrainExits()
            hb.close(), ob.close()
            assert int(sh[0].continuation_count) == int(so[0].continuation_count) > 0, (tag, base)          # gate stream, vs the oracle
            l0h, l0o = eh[eh["layer"] == 0], eo[eo["layer"] == 0]
            a, b2 = l0h[key(l0h)], l0o[key(l0o)]
            assert len(a) == len(b2) and np.array_equal(a["root"], b2["root"]) and np.array_equal(a["seq"], b2["seq"]), (tag, base)
            if len(a):
                assert np.abs(np.asarray(a["dir"], np.float64) - np.asarray(b2["dir"], np.float64)).max() <= 2e-5, (tag, base)   # gen stream, vs the oracle
            l1h, l1o = eh[eh["layer"] == 1], eo[eo["layer"] == 1]
            if tag == "b":   # identical continuation rays: the second layer is order-free — the transit stream against the oracle, exit for exit
                assert len(l1h) == len(l1o) == int(sh[0].continuation_count), (base, len(l1h), len(l1o))
                assert np.abs(sorted_dirs(l1h) - sorted_dirs(l1o)).max() <= 2e-5, base
            first = a[a["seq"] == 0]
            out[tag] = (np.asarray(first["dir"], np.float32).copy(), np.asarray(first["root"]).copy(), int(sh[0].continuation_count), sorted_dirs(l1h))
        got.append(out)
    for tag in ("a", "b"):                                                                        # hi == 0 twice: the same rays
        assert np.array_equal(got[0][tag][0], got[1][tag][0]) and got[0][tag][2] == got[1][tag][2]
    assert np.array_equal(got[0]["b"][3], got[1]["b"][3])
    for i, j in ((0, 2), (0, 3), (2, 3)):
        gi, gj = got[i]["a"], got[j]["a"]
        common = np.intersect1d(gi[1], gj[1])
        di, dj = gi[0][np.searchsorted(gi[1], common)], gj[0][np.searchsorted(gj[1], common)]
        assert (np.square(di.astype(np.float64) - dj).sum(axis=1) > 1e-10).mean() > 0.9, (i, j)   # gen stream
        assert gi[2] != gj[2] or got[i]["b"][2] != got[j]["b"][2]                                  # gate stream: other exits continue
        bi, bj = got[i]["b"][3], got[j]["b"][3]
        m = min(len(bi), len(bj))
        assert m > 100 and (np.abs(bi[:m] - bj[:m]).max(axis=1) > 1e-4).mean() > 0.9, (i, j)       # transit stream: another second layer
