"""Host pyramid builder (short candidate lists + the exhaustive second pass) against the oracle's exhaustive builder: topology disagreements
(face / fan-triangle counts) over an ordinary and a deliberately degenerate sampler of pyramid parameters, and whether the oracle's own table
is a polytope (fan triangles == 2 V - 4) where they disagree.  CPU only.  Round 3: 11 / 30000 ordinary and 2453 / 30000 degenerate
disagreements with tolerance feasibility and tolerance face claims; 0 and 0 with exact feasibility and incidence masks on both sides."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from ice_halo_sim_amd import abi, backend
from tests import _libs
from tests._libs import fptr
L, O = backend.load_library(), _libs.oracle()
def run(sampler, n, seed):
    rng = np.random.default_rng(seed); dis = incons_h = incons_o = 0
    for t in range(n):
        wu, wl, h1, h2, h3, d = sampler(rng)
        a, b = abi.HaloGeomTables(), abi.HaloGeomTables()
        L.halo_host_pyramid_geometry(wu, wl, h1, h2, h3, fptr(d), C.byref(a))
        O.ho_pyramid_geometry(wu, wl, h1, h2, h3, fptr(d), C.byref(b))
        if (a.face_cnt, a.tri_cnt) != (b.face_cnt, b.tri_cnt): dis += 1
    return dis
def degenerate(rng):
    wu, wl = rng.uniform(5, 85, 2)
    h1, h3 = [float(abs(rng.choice([rng.uniform(0, 1), rng.normal(0, 3e-4), rng.uniform(0.99, 1.0)]))) for _ in range(2)]
    h2 = float(abs(rng.choice([rng.uniform(0, 2), rng.normal(0, 3e-4)])))
    d = (1.0 + rng.normal(0, rng.choice([0, 1e-4, 0.1, 0.5]), 6)).astype(np.float32)
    return wu, wl, h1, h2, h3, d
def ordinary(rng):
    wu, wl = rng.uniform(5, 85, 2)
    return wu, wl, float(rng.uniform(0, 1)), float(rng.uniform(0, 2)), float(rng.uniform(0, 1)), (1.0 + rng.normal(0, rng.choice([0, 0.1, 0.3]), 6)).astype(np.float32)
print("degenerate sampler: %d / 30000 topology disagreements with the oracle" % run(degenerate, 30000, 3))
print("ordinary sampler:   %d / 30000" % run(ordinary, 30000, 4))
# of the remaining disagreements: is the oracle's own table a polytope (fan triangles == 2 V - 4)?
import ctypes as C
O.ho_pyramid_face_mask.restype = C.c_int
def detail(sampler, n, seed):
    rng = np.random.default_rng(seed); ok_o = bad_o = 0
    for t in range(n):
        wu, wl, h1, h2, h3, d = sampler(rng)
        a, b = abi.HaloGeomTables(), abi.HaloGeomTables()
        L.halo_host_pyramid_geometry(wu, wl, h1, h2, h3, fptr(d), C.byref(a))
        O.ho_pyramid_geometry(wu, wl, h1, h2, h3, fptr(d), C.byref(b))
        if (a.face_cnt, a.tri_cnt) != (b.face_cnt, b.tri_cnt):
            nv = C.c_int(0)
            O.ho_pyramid_face_mask(wu, wl, h1, h2, h3, fptr(d), C.byref(nv))
            if b.tri_cnt == 2 * nv.value - 4: ok_o += 1
            else: bad_o += 1
    return ok_o, bad_o
print("degenerate sampler, remaining: oracle table a polytope in %d, not in %d" % detail(degenerate, 30000, 3))
print("ordinary sampler, remaining:   oracle table a polytope in %d, not in %d" % detail(ordinary, 30000, 4))
# round 4: a table that is no polytope after the exhaustive pass is REFUSED (the empty crystal) on host, device and oracle alike: how many
def refused(sampler, n, seed):
    rng = np.random.default_rng(seed); e_h = e_o = valid = 0
    for t in range(n):
        wu, wl, h1, h2, h3, d = sampler(rng)
        a, b = abi.HaloGeomTables(), abi.HaloGeomTables()
        L.halo_host_pyramid_geometry(wu, wl, h1, h2, h3, fptr(d), C.byref(a))
        O.ho_pyramid_geometry(wu, wl, h1, h2, h3, fptr(d), C.byref(b))
        e_h += a.face_cnt == 0; e_o += b.face_cnt == 0
        if a.face_cnt:
            valid += 1
            assert a.tri_cnt >= 4
    return e_h, e_o, valid
print("degenerate sampler: empty crystals host %d / oracle %d of 10000 (invalid parameters and refused tables); %d polytopes" % refused(degenerate, 10000, 5))
print("ordinary sampler:   empty crystals host %d / oracle %d of 10000; %d polytopes" % refused(ordinary, 10000, 6))
