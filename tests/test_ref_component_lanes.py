"""The reference's per-colour-class Y-lane contracts (test/unit-correctness/server/test_render_consumer_component_lanes.cpp, cases 1-5, 7, 8;
vectors in tests/golden/ref_test_vectors.json group component_lanes), restated where THIS engine buckets exits into class lanes: in the trace
kernel's emit (halo_trace.inl fan_lanes / fan_lanes_fast — the reference's CPU consumer does it from per-ray masks, render.cpp:398), and on
the oracle's restatement of the same.

A reference "ray straight up with component mask m and weight w" becomes one host-injected ray (HaloHostRays) dropped straight down through
the top face of a unit prism with max_hits 2: its one in-frame exit is the transmitted ray through the bottom face (path 1-2), straight
down = sky-up, at the zenith pixel, carrying w times the plate's transmission — the same factor for every ray, so the lanes' shares of the
main image's Y are the reference's expectations exactly.  The mask comes from raypath-colour predicates: the ray's crystal entry names a
colour set with one `raypath [1, 2]` predicate per bit of m."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from ice_halo_sim_amd import abi, backend, scenes
from tests import _libs
from tests._oracle_backend import OracleBackend

HERE = os.path.dirname(os.path.abspath(__file__))
V = json.load(open(os.path.join(HERE, "golden", "ref_test_vectors.json")))["component_lanes"]
RD = scenes.render(abi.LENS_FISHEYE_EQUAL_AREA, V["render"]["width"], V["render"]["height"], fov=V["render"]["fov"], el=V["render"]["el"], visible=abi.VISIBLE_UPPER)
WL = scenes.wl_discrete(V["wavelength"])
FIXED = scenes.axis()      # no axis object: everything fixed (host rays are crystal-local anyway)


def _make(kind):
    if kind == "oracle":
        return OracleBackend(seed=1, threads=1)
    from ice_halo_sim_amd.backend import HipTraceBackend
    return HipTraceBackend(device=0, seed=1)


def _classes(spec):
    return [scenes.color_class([b for b in range(8) if (bits >> b) & 1], comb) for comb, bits in spec]


def _sets_for(masks):
    """one colour set per distinct non-zero mask: a `raypath [1, 2]` predicate for each of its bits"""
    distinct = sorted({m for m in masks if m})
    sets = [scenes.color_set([(scenes.filter_term("raypath", raypath=[1, 2]), "", b) for b in range(8) if (m >> b) & 1]) for m in distinct]
    return sets, {m: i + 1 for i, m in enumerate(distinct)}


def _drop(b, mask_to_id, mask, weight):
    """one ray straight down through the top face of the unit prism (compact face 0), in its own session"""
    e = scenes.entry(scenes.prism_crystal(1.0), FIXED, 1.0, 1, color_id=mask_to_id.get(mask, 0))
    sc = scenes.scene([(0.0, [e])], max_hits=2)
    b.BeginSession(sc, RD, WL, 1)
    st = b.TraceLayer(1, (np.array([[0.0, 0.0, -1.0]], np.float32), np.array([[0.05, 0.03, 0.5]], np.float32), np.array([weight], np.float32), np.array([0], np.uint32)))
    b.EndSession()
    return st


def _run(kind, classes, rays):
    b = _make(kind)
    sets, ids = _sets_for([m for m, _ in rays])
    b.set_color(sets, _classes(classes))
    for m, w in rays:
        _drop(b, ids, m, w)
    lanes = b.ReadbackClassLanes() if classes else None
    img, landed = b.ReadbackXyzAccum()
    return b, lanes, img, landed


def _check_cases(kind):
    tol = lambda x: x * V["rel_tol"] + V["abs_tol"]
    for case in V["cases"]:
        rays = list(zip(case["masks"], case["weights"]))
        b, lanes, img, landed = _run(kind, case["classes"], rays)
        b.close()
        main_y = float(img[..., 1].sum(dtype=np.float64))
        assert main_y > 0 and (img[..., 1] > 0).sum() == 1, case["name"]          # every ray landed on the one zenith pixel
        kappa = main_y / sum(case["weights"])                                     # Y per unit weight: CMF_y(550) x the plate's transmission
        sums = [float(lanes[c].sum(dtype=np.float64)) for c in range(len(case["classes"]))]
        for c, members in enumerate(case["lanes"]):
            want = kappa * sum(case["weights"][i] for i in members)
            assert abs(sums[c] - want) <= tol(want), (case["name"], c, sums[c], want)
            assert (lanes[c] > 0).sum() == (1 if members else 0)
        if case.get("sum_of_lanes_equals_main_y"):
            assert abs(sum(sums) - main_y) <= tol(main_y), case["name"]
        if case.get("sum_of_lanes_exceeds_main_y"):
            assert sum(sums) > main_y + V["abs_tol"], case["name"]


def _check_main_image_and_reset(kind):
    # 7. the same batch with and without colour classes: the main image does not move, the plain consumer has no lanes
    c = V["colored_mask_does_not_perturb_main_image"]
    rays = list(zip(c["masks"], c["weights"]))
    bp, lp, ip, landed_p = _run(kind, [], rays)
    bc, lc, ic, landed_c = _run(kind, c["classes"], rays)
    assert lp is None and bp.ReadbackClassLanes().shape[0] == 0
    assert np.allclose(ip, ic, rtol=4 * np.finfo(np.float32).eps, atol=0) and landed_p == pytest.approx(landed_c, rel=1e-7)      # ASSERT_FLOAT_EQ: 4 ulp
    bp.close(), bc.close()
    # 8. lanes accumulate across sessions until they are drained; a consumer reset zeroes them
    r = V["reset"]
    b = _make(kind)
    sets, ids = _sets_for([m for m, _ in r["batch_a"] + r["batch_b"]])
    b.set_color(sets, _classes(r["classes"]))
    for m, w in r["batch_a"]:
        _drop(b, ids, m, w)
    after_a = float(b.ReadbackClassLanes()[0].sum(dtype=np.float64))          # (a readback drains: trace_backend.hpp:471-493)
    for m, w in r["batch_a"] + r["batch_b"]:
        _drop(b, ids, m, w)
    after_ab = float(b.ReadbackClassLanes()[0].sum(dtype=np.float64))
    assert after_a > 0 and after_ab == pytest.approx(after_a * (r["batch_a"][0][1] + r["batch_b"][0][1]) / r["batch_a"][0][1], rel=V["rel_tol"])
    for m, w in r["batch_b"]:
        _drop(b, ids, m, w)
    if kind == "hip":
        b.ResetConsumer()                                                        # RenderConsumer::Reset clears the class lanes (render.cpp:598-612)
        assert float(np.abs(b.ReadbackClassLanes()).sum()) == 0.0
    b.close()


def test_class_lane_contracts_on_the_oracle():
    _check_cases("oracle")
    _check_main_image_and_reset("oracle")


@pytest.mark.gpu
def test_class_lane_contracts_on_the_kernels():
    _check_cases("hip")
    _check_main_image_and_reset("hip")


@pytest.mark.gpu
def test_consumed_lanes_are_added_to_the_device_lanes():
    """ConsumeDeviceFused's lane half (render.cpp:150-185) through halo_consumer_consume: lanes drained elsewhere are ADDED to this consumer's,
    class by class; a class-count mismatch is refused."""
    from ice_halo_sim_amd.backend import BackendError, HipTraceBackend
    hb = HipTraceBackend(device=0, seed=1)
    cls = [["any", 1], ["any", 2]]
    sets, ids = _sets_for([1, 2])
    hb.set_color(sets, _classes(cls))
    _drop(hb, ids, 1, 0.5)
    _drop(hb, ids, 2, 0.25)
    own = hb.ReadbackClassLanes()
    hb.LoadClassLanes(own)                                       # (put them back: the readback drained them)
    extra = np.zeros_like(own)
    extra[0, 3, 4], extra[1, 10, 11] = 2.0, 3.0
    img = np.zeros((RD.height, RD.width, 3), np.float32)
    hb.Consume(img, 0.0, extra)
    got = hb.ReadbackClassLanes()
    assert np.array_equal(got, own + extra)
    with pytest.raises(BackendError):
        hb.Consume(img, 0.0, extra[:1])
    hb.close()
