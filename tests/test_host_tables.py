"""Host-side table producers of the PRODUCT (exported through the C ABI, no GPU needed) against the CPU oracle.
Both restate the same reference routines; they are compiled by different compilers (hipcc/clang vs gcc) from
independently written code, so bit-equality here is a real cross-check of geometry, LUT, projection POD,
partition and refractive index.  Also checks that the library loads and exports every declared symbol."""
import ctypes as C
import re
import os

import numpy as np
import pytest

from ice_halo_sim_amd import abi, backend, scenes
from tests import _libs
from tests._libs import fptr

RNG = np.random.default_rng(77)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = backend.load_library()
    header = open(os.path.join(ROOT, "include", "halo_trace.h")).read()
    declared = set(re.findall(r"\b(halo_[a-z0-9_]+)\s*\(", header)) - {"halo_handle_t"}
    assert declared, "no declarations parsed"
    for sym in sorted(declared):
        assert hasattr(L, sym), "libhalo_hip.so does not export " + sym
    assert declared == set(backend.EXPORTED_SYMBOLS), declared ^ set(backend.EXPORTED_SYMBOLS)
    want = int(re.search(r"#define\s+HALO_ABI_VERSION\s+(\d+)", header).group(1))
    assert L.halo_abi_version() == want == 6
    for i, t in enumerate([abi.HaloScene, abi.HaloRender, abi.HaloWl, abi.HaloExitRecord, abi.HaloGeomTables,
                           abi.HaloLayerStats, abi.HaloEntry, abi.HaloColorSet, abi.HaloColorClass, abi.HaloFilter, abi.HaloRouteInfo, abi.HaloComposite]):
        assert L.halo_abi_sizeof(i) == C.sizeof(t), t.__name__


def test_product_fails_loudly_without_gpu():
    L = backend.load_library()
    if L.halo_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(backend.BackendUnavailableError):
        backend.HipTraceBackend(device=0, seed=1)


def geom_equal(a, b):
    assert a.face_cnt == b.face_cnt and a.tri_cnt == b.tri_cnt
    for name in ("face_n", "face_d", "tri_v", "tri_n", "tri_area"):
        x, y = np.frombuffer(getattr(a, name), np.uint32), np.frombuffer(getattr(b, name), np.uint32)
        assert (x == y).all(), name
    assert list(a.face_number) == list(b.face_number) and list(a.tri_face) == list(b.tri_face)


def test_prism_geometry_bit_exact():
    L, O = backend.load_library(), _libs.oracle()
    cases = [(1.0, np.ones(6)), (1.3, np.ones(6)), (0.3, np.ones(6)), (0.0, np.ones(6)), (1e-6, np.ones(6)),
             (1.0, np.array([1, 1, 1, 1, 1, -1.5])), (1.0, np.array([1, 0.2, 1, 1, 0.2, 1])), (2.0, np.array([1, 3, 1, 3, 1, 3]))]
    for _ in range(400):
        cases.append((float(abs(RNG.normal(1.0, 0.5))), 1.0 + RNG.normal(0, RNG.choice([0.05, 0.2, 0.5, 0.8]), 6)))
    empty = 0
    for h, dist in cases:
        d = np.asarray(dist, np.float32)
        a, b = abi.HaloGeomTables(), abi.HaloGeomTables()
        assert L.halo_host_prism_geometry(float(h), fptr(d), C.byref(a)) == 0
        O.ho_prism_geometry(float(h), fptr(d), C.byref(b))
        geom_equal(a, b)
        empty += a.face_cnt == 0
    assert empty >= 3
    # regular hexagon invariants (reference test_closed_form_prism.cpp:164-187)
    g = abi.HaloGeomTables()
    L.halo_host_prism_geometry(1.0, fptr(np.ones(6, np.float32)), C.byref(g))
    assert g.face_cnt == 8 and g.tri_cnt == 20 and list(g.face_number)[:8] == [1, 2, 3, 4, 5, 6, 7, 8]
    d = np.frombuffer(g.face_d, np.float32)[:8]
    assert np.allclose(d[:2], -0.5) and np.allclose(d[2:], -np.sqrt(3) / 4, atol=1e-7)
    area = np.frombuffer(g.tri_area, np.float32)[:20]
    hexa = 3 * np.sqrt(3) / 2 * 0.25
    assert area[:4].sum() == pytest.approx(hexa, rel=1e-6) and area[8:].sum() == pytest.approx(6 * 0.5 * 1.0, rel=1e-6)


@pytest.mark.parametrize("dist", [(abi.DIST_GAUSS, 0.0, 0.3), (abi.DIST_GAUSS, 90.0, 0.8), (abi.DIST_GAUSS, 85.0, 10.0),
                                  (abi.DIST_UNIFORM, 45.0, 30.0), (abi.DIST_UNIFORM, 90.0, 360.0), (abi.DIST_ZIGZAG, 85.0, 30.0),
                                  (abi.DIST_LAPLACIAN, 0.0, 2.0), (abi.DIST_LAPLACIAN, 90.0, 2.0), (abi.DIST_GAUSS, 10.0, 0.0)])
def test_lat_lut_bit_exact_and_monotone(dist):
    L, O = backend.load_library(), _libs.oracle()
    d = abi.HaloDist(*dist)
    a = [np.zeros(257, np.float32) for _ in range(3)]
    b = [np.zeros(257, np.float32) for _ in range(3)]
    assert L.halo_host_build_lat_lut(C.byref(d), *[fptr(x) for x in a]) == 0
    O.ho_build_lat_lut(C.byref(d), *[fptr(x) for x in b])
    for x, y in zip(a, b):
        assert (x.view(np.uint32) == y.view(np.uint32)).all()
    theta, cdf, flip = a
    # BuildLatLut contract (reference lat_lut.hpp:14-30): cdf strictly increasing (or the degenerate ramp), theta
    # non-decreasing inside [0, pi], flip probabilities in [0, 1]
    assert (np.diff(cdf) > 0).all() and (np.diff(theta) >= 0).all()
    assert 0 <= theta[0] and theta[-1] <= np.float32(np.pi) + 1e-6 and ((flip >= 0) & (flip <= 1)).all()


def test_proj_params_bit_exact():
    L, O = backend.load_library(), _libs.oracle()
    for lens in range(11):
        for _ in range(20):
            cfg = scenes.render(lens, int(RNG.choice([512, 1920, 2048])), int(RNG.choice([256, 1080, 1024])), fov=float(RNG.uniform(20, 180)),
                                az=float(RNG.uniform(-180, 180)), el=float(RNG.uniform(-90, 90)), ro=float(RNG.uniform(-45, 45)),
                                visible=int(RNG.integers(0, 3)), overlap=float(RNG.choice([0.0, 0.0872])),
                                lens_shift=(int(RNG.integers(-99, 99)), int(RNG.integers(-99, 99))))
            a, b = abi.ProjParams(), abi.ProjParams()
            assert L.halo_host_build_proj_params(C.byref(cfg), C.byref(a)) == 0
            O.ho_build_proj_params(C.byref(cfg), C.byref(b))
            assert bytes(a) == bytes(b)


def test_partition_sequences_and_refractive_index():
    L, O = backend.load_library(), _libs.oracle()
    for _ in range(200):
        n = int(RNG.integers(1, 9))
        prop = RNG.uniform(-0.2, 5, n).astype(np.float32)
        ca, cb = np.zeros(n), np.zeros(n)
        for _batch in range(6):
            rays = int(RNG.integers(0, 1000))
            oa, ob = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
            L.halo_host_partition(fptr(prop), n, rays, ca.ctypes.data_as(C.POINTER(C.c_double)), oa.ctypes.data_as(C.POINTER(C.c_uint64)))
            O.ho_partition(fptr(prop), n, rays, cb.ctypes.data_as(C.POINTER(C.c_double)), ob.ctypes.data_as(C.POINTER(C.c_uint64)))
            assert (oa == ob).all() and np.array_equal(ca, cb)
            if prop.clip(0).sum() > 0:
                assert oa.sum() == rays  # exact total (reference test_simulator.cpp:38-233)
    for wl in np.linspace(300, 950, 261):
        assert L.halo_host_refractive_index(float(wl)) == O.ho_ice_refractive_index(float(wl))


def test_pyramid_topology_matches_reference_goldens_and_oracle():
    """The reference's fixed pyramid pools (closed_form_samples_generated.hpp) with its own topology goldens
    (pyramid_topology_golden_generated.hpp: vertex count + present-face mask): the oracle must reproduce them, and the
    product's tables must equal the oracle's bit for bit."""
    L, O = backend.load_library(), _libs.oracle()
    O.ho_pyramid_face_mask.restype = C.c_int
    O.ho_pyramid_face_mask.argtypes = [C.c_float] * 5 + [C.POINTER(C.c_float), C.POINTER(C.c_int)]
    O.ho_pyramid_geometry.restype = None
    O.ho_pyramid_geometry.argtypes = [C.c_float] * 5 + [C.POINTER(C.c_float), C.POINTER(abi.HaloGeomTables)]
    G = np.load(os.path.join(ROOT, "tests", "golden", "ref_pyramid_goldens.npz"))
    dropped = []

    def check(wu, wl, h1, h2, h3, dist, golden):
        n = C.c_int()
        mask = O.ho_pyramid_face_mask(wu, wl, h1, h2, h3, fptr(dist), C.byref(n))
        if int(golden[2]) & 0x40:
            # kClosedFormPathTagClaimedFaceDropped (geo3d_closedform.hpp): the reference's own marker that ITS result lost a face it had
            # reached and "the surface they bounded is left open" (geo3d_closedform.cpp:1213-1232; flat-tail 89.5 #2, one of the four open
            # surfaces test_closed_form_pyramid.cpp:1664-1717 counts).  Not a topology to reproduce: here the face must be there —
            # every face of the golden and the dropped ones, on a closed surface (tests/test_ref_pyramid_properties.py checks Euler).
            dropped.append(1)
            assert mask & int(golden[1]) == int(golden[1]) and mask != int(golden[1]) and n.value >= int(golden[0])
        else:
            assert (n.value, mask) == (int(golden[0]), int(golden[1]))
        a, b = abi.HaloGeomTables(), abi.HaloGeomTables()
        assert L.halo_host_pyramid_geometry(wu, wl, h1, h2, h3, fptr(dist), C.byref(a)) == 0
        O.ho_pyramid_geometry(wu, wl, h1, h2, h3, fptr(dist), C.byref(b))
        assert bytes(a) == bytes(b)
        assert a.face_cnt == bin(mask).count("1") and 0 < a.tri_cnt <= 64
        # closed solid: fan areas weighted by outward normals sum to zero (divergence theorem)
        nrm = np.frombuffer(a.tri_n, np.float32)[: a.tri_cnt * 3].reshape(-1, 3)
        area = np.frombuffer(a.tri_area, np.float32)[: a.tri_cnt]
        assert np.abs((nrm * area[:, None]).sum(0)).max() < 1e-4

    for s, t in zip(G["wc_samples"], G["wc_topology"]):
        check(float(s[0]), float(s[1]), float(s[2]), float(s[3]), float(s[4]), np.ascontiguousarray(s[5:11]), t)
    for s, t in zip(G["miller_samples"], G["miller_topology"]):
        wu, wl = scenes.miller_wedge_deg(int(s[0]), int(s[1])), scenes.miller_wedge_deg(int(s[2]), int(s[3]))
        check(wu, wl, float(s[4]), float(s[5]), float(s[6]), np.ascontiguousarray(s[7:13]), t)
    # the six flat-tail pools (wedge 85 .. 89.5 degrees, pyramid_topology_golden_generated.hpp:184-264): the thin-cap regime
    for tag in ("85", "87", "875", "88", "89", "895"):
        for s, t in zip(G["flat%s_samples" % tag], G["flat%s_topology" % tag]):
            check(float(s[0]), float(s[1]), float(s[2]), float(s[3]), float(s[4]), np.ascontiguousarray(s[5:11]), t)
    assert len(dropped) == 1   # of 728 goldens, 727 are reproduced and one is the reference's acknowledged open surface
    # empty / degenerate inputs give the empty crystal on both sides
    e = abi.HaloGeomTables()
    assert L.halo_host_pyramid_geometry(28.0, 28.0, 0.0, 0.0, 0.0, fptr(np.ones(6, np.float32)), C.byref(e)) != 0 and e.face_cnt == 0


def test_graft_entry_build_runs():
    """The driver's build check: __graft_entry__.build() compiles (or finds up to date) every native piece, loads the C-ABI
    library, resolves every exported symbol and checks the ABI version against include/halo_trace.h (no GPU needed)."""
    import __graft_entry__ as g
    g.build()


def test_shape_scalar_draw_plan_equals_the_sequential_sampler():
    """SyncGroupSampler (simulator.cpp:361-393) two ways: the host builder walks the nine scalars in draw order (grouped slots reuse the
    first member's raw value, a Gaussian draw takes two stream slots); the device generators draw scalar q on lane q from the dispatch's draw
    plan (CrystalRecipe::plan_slot / plan_src, geom::PlanShapeScalar).  Every scalar must agree bit for bit over random group assignments,
    distribution types and crystal kinds."""
    L = backend.load_library()
    rng = np.random.default_rng(5)
    kinds = ["fixed", "uniform", "gauss", "zigzag", "laplacian"]

    def dist():
        k = kinds[rng.integers(len(kinds))]
        if k == "fixed":
            return float(rng.uniform(0.2, 1.5))
        return {"type": k, "mean": float(rng.uniform(0.3, 1.3)), "std": float(rng.uniform(0.05, 0.6))}

    n_grouped = 0
    for trial in range(300):
        groups = [int(g) for g in rng.choice([0, 0, 1, 2, 3], 9)]
        if trial % 2:
            cr = scenes.pyramid_crystal(dist(), dist(), dist(), face_distance=[dist() for _ in range(6)], sync_group=groups)
        else:
            cr = scenes.prism_crystal(dist(), [dist() for _ in range(6)], sync_group=groups)
        for idx in (0, 7, 2**32 + 5, 10_000_000_123):
            a, b = np.zeros(9, np.float32), np.zeros(9, np.float32)
            assert L.halo_host_shape_scalars(C.byref(cr), 1234, idx, 0, fptr(a)) == 0
            assert L.halo_host_shape_scalars(C.byref(cr), 1234, idx, 1, fptr(b)) == 0
            assert a.tobytes() == b.tobytes(), (trial, groups, a, b)
            if cr.kind == abi.CRYSTAL_PRISM:
                assert a[1] == 0.0 and a[2] == 0.0          # a prism has one height
            for g in set(groups) - {0}:
                members = [q for q in range(9) if groups[q] == g and (cr.kind != abi.CRYSTAL_PRISM or q not in (1, 2))]
                n_grouped += len(members) > 1
                assert len({float(a[q]) for q in members}) <= 1, (groups, a)   # one raw draw per group
    assert n_grouped > 500
