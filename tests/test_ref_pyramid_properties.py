"""The reference's structural contracts for the closed-form pyramid (test/golden-analytic/core/test_closed_form_pyramid.cpp), restated on
this repo's geometry: the product's host builder (halo_host_pyramid_geometry, csrc/halo_geom.h), the oracle's exhaustive builder
(ho_pyramid_geometry) and — `-m gpu` half — the device team generator (halo_pyrgen_team_kernel through halo_generate_shapes).

Inputs, tolerances and expectations are the reference's (tests/golden/ref_test_vectors.json group closed_form_pyramid_contracts, each
entry with its file:line; the sample pools in tests/golden/ref_pyramid_goldens.npz).  What is read is what the trace kernels read: the
fan-triangle table of HaloGeomTables, from which each face's polygon is rebuilt (fan (0, k, k+1) -> vertices 0, 1, ..., m-1)."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from ice_halo_sim_amd import abi, backend, scenes
from tests import _libs

HERE = os.path.dirname(os.path.abspath(__file__))
V = json.load(open(os.path.join(HERE, "golden", "ref_test_vectors.json")))["closed_form_pyramid_contracts"]
G = np.load(os.path.join(HERE, "golden", "ref_pyramid_goldens.npz"))
SQRT3 = math.sqrt(3.0)


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _builders():
    L, O = backend.load_library(), _libs.oracle()
    O.ho_pyramid_geometry.restype = None
    O.ho_pyramid_geometry.argtypes = [C.c_float] * 5 + [C.POINTER(C.c_float), C.POINTER(abi.HaloGeomTables)]

    def product(wu, wl, h1, h2, h3, dist):
        g = abi.HaloGeomTables()
        L.halo_host_pyramid_geometry(wu, wl, h1, h2, h3, fptr(np.asarray(dist, np.float32)), C.byref(g))
        return g

    def oracle(wu, wl, h1, h2, h3, dist):
        g = abi.HaloGeomTables()
        O.ho_pyramid_geometry(wu, wl, h1, h2, h3, fptr(np.asarray(dist, np.float32)), C.byref(g))
        return g
    return {"product": product, "oracle": oracle}


class _Lazy(dict):
    """(the libraries are loaded by the first test that asks, not at collection: in a -m gpu session torch's HIP runtime must come up before
    libhalo_hip.so's — tests/conftest.py)"""

    def __missing__(self, key):
        self.update(_builders())
        return self[key]


BUILDERS = _Lazy()


def char_len(wu, wl, h1, h2, h3, dist):
    """ProductionCharLen (test_closed_form_pyramid.cpp:165-177)"""
    def a1(alpha, h_side):
        return SQRT3 / 4.0 / math.tan(math.radians(alpha)) if (h_side > 1e-6 and 0.1 <= alpha <= 89.9) else 0.0
    md = max(abs(float(d)) for d in dist)
    m = max(0.5 * (h1 + h2 + h3), SQRT3 / 8.0 * md, SQRT3 / 8.0 * (0.5 * h2 + max(a1(wu, h1), a1(wl, h3)) * md))
    return max(1.0, m)


def slot_of(face_number):
    """reference face numbers (1, 2 basal; 3-8 prism; 13-18 upper; 23-28 lower: crystal.hpp) -> ClosedFormPyramidResult slot 0..19"""
    n = int(face_number)
    return n - 1 if n <= 8 else (n - 13 + 8 if n <= 18 else n - 23 + 14)


class Census:
    """TakeStructuralCensus (test_closed_form_pyramid.cpp:520-575) on the published polygons."""

    def __init__(self, g):
        self.f = g.face_cnt
        tv = np.frombuffer(g.tri_v, np.float32)[: g.tri_cnt * 9].reshape(-1, 3, 3)
        tf = np.frombuffer(g.tri_face, np.int32)[: g.tri_cnt]
        self.polys = [[] for _ in range(g.face_cnt)]
        for t in range(g.tri_cnt):
            p = self.polys[tf[t]]
            if not p:
                p.extend([tuple(tv[t, 0]), tuple(tv[t, 1]), tuple(tv[t, 2])])
            else:
                assert tuple(tv[t, 0]) == p[0] and tuple(tv[t, 1]) == p[-1]      # a fan: (v0, v_k, v_k+1)
                p.append(tuple(tv[t, 2]))
        self.slots = [slot_of(n) for n in np.frombuffer(g.face_number, np.int32)[: g.face_cnt]]
        self.mask = sum(1 << s for s in self.slots)
        verts, edges = {}, {}
        for p in self.polys:
            for k in range(len(p)):
                a, b = verts.setdefault(p[k], len(verts)), verts.setdefault(p[(k + 1) % len(p)], len(verts))
                edges[(min(a, b), max(a, b))] = edges.get((min(a, b), max(a, b)), 0) + 1
        self.verts = np.array(list(verts.keys()), np.float64).reshape(-1, 3)
        self.v, self.e = len(verts), len(edges)
        self.non_manifold = sum(1 for c in edges.values() if c != 2)
        self.chi = self.v - self.e + self.f
        n = np.frombuffer(g.face_n, np.float32)[: g.face_cnt * 3].reshape(-1, 3).astype(np.float64)
        d = np.frombuffer(g.face_d, np.float32)[: g.face_cnt].astype(np.float64)
        self.n, self.d = n, d
        self.off_plane = 0.0
        self.areas = []
        for f, p in enumerate(self.polys):
            q = np.array(p, np.float64)
            if len(q):
                self.off_plane = max(self.off_plane, float(np.abs(q @ n[f] + d[f]).max()))
                self.areas.append(0.5 * float(np.linalg.norm(np.cross(q, np.roll(q, -1, axis=0)).sum(0))))   # Newell (:185-203)
        self.vtx_per_slot = {s: len(p) for s, p in zip(self.slots, self.polys)}

    @property
    def closed(self):
        return self.f > 0 and self.chi == 2 and self.non_manifold == 0


def _pool(name):
    return [(float(s[0]), float(s[1]), float(s[2]), float(s[3]), float(s[4]), np.ascontiguousarray(s[5:11])) for s in G[name + "_samples"]]


@pytest.mark.parametrize("who", ["product", "oracle"])
def test_vertex_plane_self_consistency(who):
    """(a) every body vertex inside every present face's half-space, (b) >= 3 coplanar body vertices per present face, (c) every entry of a
    face's own polygon on that face's plane — 1e-4 of the characteristic length, the well-conditioned pool (:429-487)."""
    c = V["vertex_plane_self_consistency"]
    build = BUILDERS[who]
    for s in _pool(c["pool"]):
        cs = Census(build(*s))
        tol = c["rel_tol"] * char_len(*s)
        assert cs.f > 0
        sd = cs.verts @ cs.n.T + cs.d[None, :]          # signed distance of every vertex to every present plane (unit normals)
        assert sd.max() <= tol
        assert (np.abs(sd) <= tol).sum(0).min() >= c["min_coplanar"]
        assert cs.off_plane <= tol


@pytest.mark.parametrize("who", ["product", "oracle"])
def test_structural_validity_param_scan(who):
    """Eleven legal crystals — regular / M-shaped / mildly irregular / phase-rotated hexagons x prism_h in {0, 0.3} x apex-reaching and
    truncated, one- and two-sided cones: polygon entries on their planes, V - E + F = 2, every edge shared by exactly two faces, no present
    face without area (:578-668)."""
    c = V["structural_scan"]
    for label, wu, wl, h1, h2, h3, dist in c["cases"]:
        cs = Census(BUILDERS[who](wu, wl, h1, h2, h3, dist))
        cl = char_len(wu, wl, h1, h2, h3, dist)
        assert cs.f > 0, label
        assert cs.off_plane <= c["rel_tol"] * cl, label
        assert cs.chi == 2 and cs.non_manifold == 0, (label, cs.v, cs.e, cs.f, cs.non_manifold)
        assert min(cs.areas) > c["min_area_frac"] * cl * cl, label


@pytest.mark.parametrize("who", ["product", "oracle"])
def test_degenerate_pools_degrade_gracefully(who):
    """DegenerateContractSafe (:1129-1198): the two pools that sit on the merge-tolerance boundary by construction give finite vertices and
    present faces with areas in [0, 4 x the characteristic area]; and — stronger than the reference asks of itself — what is published is
    either the empty crystal or a closed surface (the reference counts its own open surfaces, see the ratchet below)."""
    c = V["degenerate_contract_safe"]
    for pool in c["pools"]:
        for i, s in enumerate(_pool(pool)):
            g = BUILDERS[who](*s)
            assert 0 <= g.face_cnt <= 20 and 0 <= g.tri_cnt <= 64
            if g.face_cnt == 0:
                continue
            cs = Census(g)
            ca = char_len(*s) ** 2
            assert np.isfinite(cs.verts).all()
            assert all(np.isfinite(a) and 0.0 <= a <= c["area_upper_factor"] * ca for a in cs.areas), (pool, i)
            assert cs.closed, (pool, i, cs.v, cs.e, cs.f, cs.non_manifold)


@pytest.mark.parametrize("who", ["product", "oracle"])
def test_specialised_batches_named_shapes_and_slivers_are_closed_solids(who):
    """The shoulder / apex / face-drop batches (:1230-1262), the two named ordinary shapes that must keep all twenty faces (:1304-1345) and
    the three named slivers that must close without being coarsened away (:1579-1663: at least 12 / 19 / 16 faces)."""
    build = BUILDERS[who]
    b = V["specialised_batches"]
    for name in ("shoulder", "apex", "face_drop"):
        for wu, wl, h1, h2, h3, dist in b[name]:
            cs = Census(build(wu, wl, h1, h2, h3, dist))
            assert cs.closed, (name, wu, wl, h1, h2, h3, dist, cs.chi, cs.non_manifold)
    for label, wu, wl, h1, h2, h3, dist in V["named_ordinary_shapes"]["samples"]:
        cs = Census(build(wu, wl, h1, h2, h3, dist))
        assert cs.f == V["named_ordinary_shapes"]["expect_faces"] and cs.closed, (label, cs.f, cs.chi, cs.non_manifold)
    for label, wu, wl, h1, h2, h3, dist, min_faces in V["named_slivers"]["samples"]:
        cs = Census(build(wu, wl, h1, h2, h3, dist))
        assert cs.closed and cs.f >= min_faces, (label, cs.f, cs.chi, cs.non_manifold)


@pytest.mark.parametrize("who", ["product", "oracle"])
def test_combinatorics_are_invariant_under_uniform_scaling(who):
    """Lengths (dist[], h2) times 2^k, k = -6 .. 6 — exact in binary floating point, so the scaled input is a strictly similar solid: which
    faces are present and how many vertices bound each may not change (:1400-1509).  Pools: well-conditioned, both degenerate, flat-tail 89."""
    c = V["scale_invariance"]
    build = BUILDERS[who]
    compared = 0
    for pool in c["pools"]:
        for i, (wu, wl, h1, h2, h3, dist) in enumerate(_pool(pool)):
            def sig(k):
                g = build(wu, wl, h1, float(np.float32(h2) * np.float32(2.0 ** k)), h3, dist * np.float32(2.0 ** k))
                if g.face_cnt == 0:
                    return (0, ())
                cs = Census(g)
                return (cs.mask, tuple(sorted(cs.vtx_per_slot.items())))
            ref = sig(0)
            for k in range(c["min_exp"], c["max_exp"] + 1):
                if k:
                    compared += 1
                    assert sig(k) == ref, (pool, i, k)
    assert compared > 3000


@pytest.mark.parametrize("who", ["product", "oracle"])
def test_truncation_cap_survives_every_legal_upper_h(who):
    """upper_h < 1 is a truncated pyramid: the top cap (slot 0) must be there for wedge angles down to 0.2 degrees, 1 - upper_h down to 1e-3,
    dist 1 and 1000, with and without a prism section (whose six faces must survive too) — and nothing may be left open (:1510-1578)."""
    c = V["truncation_cap"]
    cells = 0
    for ph in c["prism_h"]:
        for alpha in c["alphas"]:
            for d in c["dists"]:
                for h1 in c["upper_h"]:
                    cs = Census(BUILDERS[who](alpha, alpha, h1, ph, 0.0, [d] * 6))
                    cells += 1
                    where = (alpha, h1, d, ph)
                    assert cs.mask & 1, where
                    assert cs.closed, (where, cs.chi, cs.non_manifold)
                    if ph > 0.0:
                        assert (cs.mask >> 2) & 0x3F == 0x3F, where
    assert cells == c["cells"]


@pytest.mark.parametrize("who", ["product", "oracle"])
def test_committed_pools_publish_no_open_surface(who):
    """The reference's ratchet (:1664-1717) admits that four of its committed samples come out as OPEN surfaces (f895#2, deg030#1,
    deg050#0, deg050#19) and only forbids the count to grow.  These builders take every vertex from one plane-triple solve and check
    Euler's count before publishing, so the ceiling here is zero: over the same nine pools every non-empty result is a closed 2-manifold."""
    c = V["open_surface_ratchet"]
    evaluated, opened = 0, []
    for pool in c["pools"]:
        for i, s in enumerate(_pool(pool)):
            g = BUILDERS[who](*s)
            evaluated += 1
            if g.face_cnt and not Census(g).closed:
                opened.append("%s#%d" % (pool, i))
    assert evaluated > c["min_evaluated"]
    assert not opened, opened


@pytest.mark.gpu
def test_device_team_generator_meets_the_same_contracts():
    """The device generator (halo_pyrgen_team_kernel, one team of 32 lanes per crystal) on the named inputs of the contracts above, fed as
    crystals whose nine shape scalars are fixed: its tables must be the host builder's, byte for byte — so every contract above holds on
    the device — and the census is taken on the device's own tables once more for the samples that were defects in the reference."""
    from ice_halo_sim_amd.backend import HipTraceBackend
    hb = HipTraceBackend(device=0, seed=1)
    cases = [tuple(x[1:7]) for x in V["structural_scan"]["cases"]] + [tuple(x[1:7]) for x in V["named_ordinary_shapes"]["samples"]] + \
            [tuple(x[1:7]) for x in V["named_slivers"]["samples"]] + [tuple(x) for k in ("shoulder", "apex", "face_drop") for x in V["specialised_batches"][k]]
    cases += [(s[0], s[1], s[2], s[3], s[4], list(map(float, s[5]))) for tag in ("flat895", "degenerate030", "degenerate050") for s in _pool(tag)]
    closed = 0
    for wu, wl, h1, h2, h3, dist in cases:
        cr = scenes.pyramid_crystal(float(h1), float(h2), float(h3), upper_wedge=float(wu), lower_wedge=float(wl), face_distance=[float(d) for d in dist])
        dev = hb.generate_shapes(cr, 0, 1, on_device=True)[0]
        host = BUILDERS["product"](wu, wl, h1, h2, h3, dist)
        assert bytes(dev) == bytes(host), (wu, wl, h1, h2, h3, dist)
        if dev.face_cnt:
            assert Census(dev).closed
            closed += 1
    hb.close()
    assert closed > 100
