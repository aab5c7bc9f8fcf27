"""Round-5 vector groups of tests/golden/ref_test_vectors.json — literal inputs and expectations of five more of the reference's test files,
each group with its file:line — run on the oracle, on the product's host side and (`-m gpu` halves) on the HIP kernels:

  ray_num_semantics       test/unit-correctness/server/test_ray_num_semantics.cpp        -> ice_halo_sim_amd.config.TraceJob.per_wavelength_ray_num
  device_filter_fixture   test/unit-correctness/core/test_device_filter_check_host.cpp   -> the ten filter configs x two axes of its fixture on the
                          product's fast filter tables (what the production filter kernels evaluate) and on the oracle's matcher; GPU: the kernels
  post_snapshot_fusion    test/unit-correctness/server/test_render_consumer_post_snapshot_fusion.cpp -> halo_consumer_consume + halo_consumer_snapshot,
                          byte for byte against the chain rebuilt from the reference's own colour primitives (oracle/_ref)
  scatter_outgoing        test/parity-cross-backend/backend/test_cpu_trace_backend.cpp:305-370 -> projection + SpectrumToXyz, bit for bit
  host_injected_crystal   test/parity-cross-backend/backend/test_cpu_trace_backend.cpp:737-777 -> HaloHostRays::crystal consumes no shape draw
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from ice_halo_sim_amd import abi, backend, config, scenes
from tests import _libs
from tests._oracle_backend import OracleBackend

HERE = os.path.dirname(os.path.abspath(__file__))
V = json.load(open(os.path.join(HERE, "golden", "ref_test_vectors.json")))
needs_ref = pytest.mark.skipif(not _libs.have_ref(), reason="oracle/_ref/libref_shared.so is built only where /root/reference exists (build container; .gpurunignore keeps it there)")


# ---- ray_num semantics --------------------------------------------------------------------------------------------------------------------
def test_per_wavelength_ray_num_is_the_ceiling_and_the_identity_for_one_wavelength():
    for total, n_wl, want in V["ray_num_semantics"]["cases"]:
        job = config.TraceJob.__new__(config.TraceJob)
        job.ray_num, job.wavelengths = total, [None] * n_wl
        assert job.per_wavelength_ray_num() == want, (total, n_wl)


# ---- the device filter fixture ------------------------------------------------------------------------------------------------------------
def _fixture_axis(spec):
    kinds = {"none": abi.DIST_NONE, "uniform": abi.DIST_UNIFORM}
    a = abi.HaloAxis()
    for name in ("azimuth", "latitude", "roll"):
        t, c, s = spec[name]
        d = getattr(a, name)
        d.type, d.center, d.spread = kinds[t], float(c), float(s)
    return a


def _fixture_term(t):
    kw = {k: v for k, v in t.items() if k != "type"}
    return scenes.filter_term(t["type"], **kw)


def _fixture_filters():
    out = []
    for f in V["device_filter_fixture"]["filters"]:
        if "or" in f:
            out.append(scenes.complex_filter([[_fixture_term(t) for t in clause] for clause in f["or"]], f["symmetry"], f["action"]))
        else:
            out.append(scenes.simple_filter(_fixture_term(f["term"]), f["symmetry"], f["action"]))
    return out


def _fixture_rays(rng, n, n_filters):
    ln = rng.integers(1, 6, n)
    faces = rng.integers(1, 9, (n, 16)).astype(np.uint8)       # a unit hex prism: face numbers 1..8
    d = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), np.abs(rng.uniform(-1, 1, n)) + 0.1], 1)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return ln, faces, d, rng.integers(0, 16, n), rng.integers(0, n_filters, n)


def test_device_filter_fixture_fast_tables_equal_the_matcher():
    """The fixture's 100000 rays per axis on its ten filters: the product's fast form (FastTables) and the oracle's reduction-based matcher
    (restating filter_shared.h DeviceFilterCheck == FilterSpec::Check) must give the same verdict on every ray — and a few verdicts that
    follow from the configs alone are checked as known answers."""
    L, O = backend.load_library(), _libs.oracle()
    filters = _fixture_filters()
    n = V["device_filter_fixture"]["rays_per_axis"]
    for k, (name, spec) in enumerate(sorted(V["device_filter_fixture"]["axes"].items())):
        ax = _fixture_axis(spec)
        # (by the reference's own rule — crystal.cpp:708-730: azimuth rotationally symmetric and the roll ANCHOR a multiple of 30, whatever the
        # roll's type — both fixture axes are D-applicable; what its two variants really differ in is sigma_a: 0 for the uniform roll
        # centred on 0, 5 for the roll fixed at 30)
        assert O.ho_is_d_applicable(C.byref(ax)) == 1
        assert O.ho_compute_sigma_a(ax.roll.center) == (5 if name == "d_applicable" else 0)
        ln, faces, d, cid, fi = _fixture_rays(np.random.default_rng(0xCAFEBABE + k), n, len(filters))
        passed = np.zeros(len(filters), np.int64)
        for i in range(n):
            p = (C.c_uint8 * 16)(*faces[i])
            dv = (C.c_float * 3)(*d[i])
            got = C.c_int32(-1)
            assert L.halo_host_filter_fast_check(C.byref(filters[fi[i]]), C.byref(ax), p, int(ln[i]), dv, int(cid[i]), C.byref(got)) == 0
            want = O.ho_filter_check(C.byref(filters[fi[i]]), C.byref(ax), C.cast(p, C.POINTER(C.c_uint8)), int(ln[i]), C.cast(dv, C.POINTER(C.c_float)), int(cid[i]))
            assert got.value == int(want != 0), (name, int(fi[i]), list(faces[i][: ln[i]]), d[i], int(cid[i]))
            passed[fi[i]] += got.value
            # known answers: 0 passes everything; 4 is "entry face 3"; 6 is "crystal id 7"; 8 needs crystal 7 AND a two-face path
            if fi[i] == 0:
                assert got.value == 1
            elif fi[i] == 4:
                assert got.value == int(faces[i][0] == 3)
            elif fi[i] == 6:
                assert got.value == int(cid[i] == 7)
            elif fi[i] == 8 and got.value:
                assert cid[i] == 7 and ln[i] == 2
        assert (passed > 0).all() and (passed < np.bincount(fi, minlength=len(filters))).sum() >= 8     # every filter decides something


@pytest.mark.gpu
@pytest.mark.parametrize("which", [7, 9])
def test_device_filter_fixture_complex_filters_on_the_kernels(which):
    """Two of the fixture's complex filters on the production filter kernels against the oracle: P / filter_out over two raypath clauses
    (#7) and the 20-clause PBD big-OR (#9, well above the legacy 8-clause cap), column crystal of configs[1], 2^20 rays."""
    from ice_halo_sim_amd.backend import HipTraceBackend
    from tests._oracle_backend import run_session
    flt = [_fixture_filters()[which]]
    e = scenes.column_crystal_entry()
    e.filter_id = 1
    sc, rd = scenes.scene([(0.0, [e])], max_hits=7), scenes.config2_render(480, 270)
    hb, ob = HipTraceBackend(device=0, seed=13), OracleBackend(seed=13, threads=16)
    for b in (hb, ob):
        b.set_filters(flt)
    sh = run_session(hb, sc, rd, scenes.wl_discrete(550.0), 1 << 20)[0]
    so = run_session(ob, sc, rd, scenes.wl_discrete(550.0), 1 << 20)[0]
    assert hb.last_route().mode_mask == abi.MODE_FILTER
    (ih, lh), (io, lo) = hb.ReadbackXyzAccum(), ob.ReadbackXyzAccum()
    hb.close(), ob.close()
    assert 0 < so.exit_count < 5 * (1 << 20)
    assert abs(int(sh.exit_count) - int(so.exit_count)) <= 3e-4 * so.exit_count + 2
    assert lh == pytest.approx(lo, rel=2e-4)


# ---- PostSnapshot, byte for byte ----------------------------------------------------------------------------------------------------------
def _render_of(spec):
    return scenes.render(abi.LENS_FISHEYE_EQUAL_AREA, spec["width"], spec["height"], fov=spec["fov"], el=spec["el"], visible=abi.VISIBLE_UPPER)


def _scatter_with_ref(rd, dirs, weights, wl):
    """ScatterOutgoingToXyz (scatter_accum.hpp:47-110) from the reference's own pieces in oracle/_ref: ProjectExitToPixel per ray, then
    xyz[pix] += CMF * w in ray order (AccumXyzToPixel / SpectrumToXyz); landed += w for landed hits."""
    R, O = _libs.ref(), _libs.oracle()
    pp = abi.ProjParams()
    O.ho_build_proj_params(C.byref(rd), C.byref(pp))          # (bit-equal to the reference's BuildProjParams: tests/test_oracle_vs_ref.py)
    img = np.zeros((rd.height, rd.width, 3), np.float32)
    flat = img.reshape(-1)
    landed = np.float32(0.0)
    out7 = np.zeros(7, np.int32)
    one = np.zeros(3, np.float32)
    for dvec, w in zip(dirs, weights):
        R.ref_project_exit_to_pixel(C.byref(pp), float(dvec[0]), float(dvec[1]), float(dvec[2]), _libs.i32ptr(out7))
        for k in range(out7[0]):
            px, py, bump = out7[1 + 3 * k], out7[2 + 3 * k], out7[3 + 3 * k]
            if 0 <= px < rd.width and 0 <= py < rd.height:
                one[:] = 0.0
                R.ref_spectrum_to_xyz(float(wl), float(w), _libs.fptr(one))
                flat[(py * rd.width + px) * 3: (py * rd.width + px) * 3 + 3] += one
                if bump:
                    landed = np.float32(landed + np.float32(w))
    return img, float(landed)


def _expected_bytes_from_ref(xyz_raw, scale, ray_color, background):
    """test_render_consumer_post_snapshot_fusion.cpp:91-127, real-colour branch, from GamutClipXyz / XyzToLinearRgb / LinearToSrgb as compiled
    from the reference (oracle/_ref)"""
    R = _libs.ref()
    out = np.zeros(xyz_raw.shape, np.uint8)
    flat, oflat = xyz_raw.reshape(-1, 3), out.reshape(-1, 3)
    clamps = 0
    for i in range(flat.shape[0]):
        xyz = (flat[i] * np.float32(scale)).astype(np.float32)
        clipped, rgb = np.zeros(3, np.float32), np.zeros(3, np.float32)
        R.ref_gamut_clip_xyz(_libs.fptr(xyz), _libs.fptr(clipped))
        R.ref_xyz_to_linear_rgb(_libs.fptr(clipped), _libs.fptr(rgb))
        for j in range(3):
            v = np.float32(rgb[j] + np.float32(background[j]))
            c = np.float32(min(max(v, np.float32(0.0)), np.float32(1.0)))
            clamps += int(c != v)
            oflat[i, j] = int(np.float32(R.ref_linear_to_srgb(float(c))) * np.float32(255))
    return out, clamps


def _xyz_to_rgb_matrix_from_ref():
    """kXyzToRgb as the reference compiled it, read back exactly through XyzToLinearRgb (color_space.cpp:37-45: v = sum_k xyz[k] * M[j][k],
    clamped to [0, 1]): a basis vector scaled by 2^-10 (exact) with the sign that keeps the entry positive."""
    R = _libs.ref()
    M = np.zeros((3, 3), np.float32)
    for k in range(3):
        for sign in (1.0, -1.0):
            xyz, rgb = np.zeros(3, np.float32), np.zeros(3, np.float32)
            xyz[k] = sign * 2.0 ** -10
            R.ref_xyz_to_linear_rgb(_libs.fptr(xyz), _libs.fptr(rgb))
            for j in range(3):
                if rgb[j] > 0:
                    M[j, k] = np.float32(sign) * rgb[j] * np.float32(2.0 ** 10)
    return M


def _expected_bytes_gray_tint(xyz_raw, scale, ray_color, background):
    """the gray + ray_color branch (test_render_consumer_post_snapshot_fusion.cpp:96-110): gray = white point * Y, through the matrix (no
    gamut clip, no clamp before the tint), times ray_color"""
    R = _libs.ref()
    M, wp = _xyz_to_rgb_matrix_from_ref(), np.array(V["post_snapshot_fusion"]["white_point_d65"]["value"], np.float32)
    out = np.zeros(xyz_raw.shape, np.uint8)
    flat, oflat = xyz_raw.reshape(-1, 3), out.reshape(-1, 3)
    for i in range(flat.shape[0]):
        y = np.float32(flat[i, 1] * np.float32(scale))
        gray = (wp * y).astype(np.float32)
        for j in range(3):
            v = np.float32(0.0)
            for k in range(3):
                v = np.float32(v + np.float32(gray[k] * M[j, k]))
            v = np.float32(np.float32(v * np.float32(ray_color[j])) + np.float32(background[j]))
            c = np.float32(min(max(v, np.float32(0.0)), np.float32(1.0)))
            oflat[i, j] = int(np.float32(R.ref_linear_to_srgb(float(c))) * np.float32(255))
    return out


def _snapshot_case(b, case, img, landed):
    b.Consume(img, landed)
    rgb, xyz, total = b.Snapshot(intensity_factor=1.0, ray_color=tuple(case["ray_color"]), background=tuple(case["background"]))
    return rgb, xyz, total


@needs_ref
def test_post_snapshot_chain_byte_for_byte_on_the_oracle():
    """The three cases on the oracle's consumer: the image ScatterOutgoingToXyz makes of four rays straight up (one lit pixel), consumed as a
    drained image, snapshot, and compared BYTE FOR BYTE — never a tolerance — with the chain rebuilt from the reference's primitives."""
    spec = V["post_snapshot_fusion"]
    rd = _render_of(spec["render"])
    img, landed = _scatter_with_ref(rd, [[0.0, 0.0, -1.0]] * len(spec["weights"]), spec["weights"], spec["wavelength"])
    assert (img.reshape(-1, 3).sum(1) > 0).sum() == 1 and landed == pytest.approx(sum(spec["weights"]), rel=1e-6)
    for case in spec["cases"]:
        ob = OracleBackend(seed=1, threads=1)
        rgb, xyz, total = _snapshot_case(ob, case, img, landed)
        ob.close()
        assert np.array_equal(xyz, img) and total == pytest.approx(landed, rel=1e-7)
        scale = np.float32(np.float32(1.0) * np.float32(0.08) * np.float32(rd.width * rd.height) / np.float32(landed))   # ExposureScale render.cpp:96-102
        assert (rgb != 0).any()
        if case["ray_color"][0] < 0:
            want, clamps = _expected_bytes_from_ref(img, scale, case["ray_color"], case["background"])
            assert np.array_equal(rgb, want), case["name"]
            if case.get("expect_clamp"):
                assert clamps > 0 and (rgb.reshape(-1, 3) == rgb.reshape(-1, 3)[0]).all(1).sum() >= 255     # the background lifts every empty pixel
        else:   # gray + tint: Y times the D65 white point through the matrix, times ray_color
            assert np.array_equal(rgb, _expected_bytes_gray_tint(img, scale, case["ray_color"], case["background"])), case["name"]


@pytest.mark.gpu
def test_post_snapshot_chain_byte_for_byte_on_the_device():
    """... and halo_consumer_consume + halo_consumer_snapshot on the device: the same bytes as the oracle's consumer in all three cases
    (the kernel's per-pixel chain reorders nothing: element-wise float operations in the reference's order)."""
    from ice_halo_sim_amd.backend import HipTraceBackend
    spec = V["post_snapshot_fusion"]
    rd = _render_of(spec["render"])
    O = _libs.oracle()
    # the one lit pixel: where the oracle's projection puts a ray straight up, CMF(550) * w summed in ray order (float)
    pp = abi.ProjParams()
    O.ho_build_proj_params(C.byref(rd), C.byref(pp))
    r = O.ho_project_exit_to_pixel(C.byref(pp), 0.0, 0.0, -1.0)
    assert r.count == 1
    cx, cy, cz = np.zeros(1, np.float32), np.zeros(1, np.float32), np.zeros(1, np.float32)
    O.ho_cmf(float(spec["wavelength"]), _libs.fptr(cx), _libs.fptr(cy), _libs.fptr(cz))
    img = np.zeros((rd.height, rd.width, 3), np.float32)
    landed = np.float32(0.0)
    for w in spec["weights"]:
        img[r.hits[0].py, r.hits[0].px] += np.array([cx[0], cy[0], cz[0]], np.float32) * np.float32(w)
        landed = np.float32(landed + np.float32(w))
    for case in spec["cases"]:
        hb, ob = HipTraceBackend(device=0, seed=1), OracleBackend(seed=1, threads=1)
        rh, xh, th = _snapshot_case(hb, case, img, float(landed))
        ro, xo, to = _snapshot_case(ob, case, img, float(landed))
        hb.close(), ob.close()
        assert np.array_equal(xh, xo) and th == to
        assert (ro != 0).any() and np.array_equal(rh, ro), case["name"]


# ---- ScatterOutgoingMatchesReferenceScatter -----------------------------------------------------------------------------------------------
@needs_ref
def test_scatter_of_twelve_outgoing_rays_equals_the_references_scatter():
    """Twelve outgoing directions, weight 0.7, 550 nm, on the 64x64 zenith fisheye: the oracle's projection + CMF accumulate gives the image
    the reference's own ProjectExitToPixel + SpectrumToXyz give, bit for bit; the horizon ray (sky z = 0) falls outside the upper-only render."""
    spec = V["scatter_outgoing"]
    rd = _render_of(spec["render"])
    dirs, w = spec["directions"], [spec["weight"]] * len(spec["directions"])
    want, landed = _scatter_with_ref(rd, dirs, w, spec["wavelength"])
    O = _libs.oracle()
    pp = abi.ProjParams()
    O.ho_build_proj_params(C.byref(rd), C.byref(pp))
    cx, cy, cz = np.zeros(1, np.float32), np.zeros(1, np.float32), np.zeros(1, np.float32)
    O.ho_cmf(float(spec["wavelength"]), _libs.fptr(cx), _libs.fptr(cy), _libs.fptr(cz))
    got = np.zeros_like(want)
    hits = 0
    for d in dirs:
        r = O.ho_project_exit_to_pixel(C.byref(pp), float(d[0]), float(d[1]), float(d[2]))
        for k in range(r.count):
            h = r.hits[k]
            if 0 <= h.px < rd.width and 0 <= h.py < rd.height:
                got[h.py, h.px] += np.array([cx[0], cy[0], cz[0]], np.float32) * np.float32(spec["weight"])
                hits += 1
    assert got.tobytes() == want.tobytes()
    assert landed > 0 and hits == 11          # the horizon ray is the one that does not land


# ---- HostInjectedCrystalIsNotANewSample ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_host_injected_crystal_is_not_a_new_sample():
    """A scene whose prism height is gauss(1, 0.15) — every MakeCrystal call a real draw — traced with 256 host rays that bring their own
    crystal: the crystal is traced as it is, so the session's stochastic crystal sample count stays 0; the same rays WITHOUT a crystal are
    traced in sampled instances (one per 32 rays) and count them.  The exits of the first are those of the deterministic unit prism."""
    from ice_halo_sim_amd.backend import HipTraceBackend
    spec = V["host_injected_crystal"]
    L = backend.load_library()
    unit = abi.HaloGeomTables()
    assert L.halo_host_prism_geometry(1.0, _libs.fptr(np.ones(6, np.float32)), C.byref(unit)) == 0
    n = spec["count"]
    rng = np.random.default_rng(3)
    # rays entering the top basal face (compact face 0) of the unit prism, straight down and slightly tilted
    d = np.tile(np.array([[0.0, 0.0, -1.0]], np.float32), (n, 1))
    d[:, 0] = rng.uniform(-0.2, 0.2, n)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tv = np.frombuffer(unit.tri_v, np.float32)[:9].reshape(3, 3)         # first fan triangle of face 0
    uv = rng.uniform(0.05, 0.45, (n, 2)).astype(np.float32)
    p = (tv[0] + uv[:, :1] * (tv[1] - tv[0]) + uv[:, 1:] * (tv[2] - tv[0])).astype(np.float32)
    w, tf = np.ones(n, np.float32), np.zeros(n, np.uint32)
    full = {"type": "uniform", "mean": 0.0, "std": 360.0}
    stoch = scenes.scene([(0.0, [scenes.entry(scenes.prism_crystal(spec["height"]), scenes.axis(zenith=full, azimuth=full, roll=full), 1.0, 0)])], max_hits=spec["max_hits"])
    fixed = scenes.scene([(0.0, [scenes.entry(scenes.prism_crystal(1.0), scenes.axis(zenith=full, azimuth=full, roll=full), 1.0, 0)])], max_hits=spec["max_hits"])
    rd = scenes.config2_render(64, 64)
    out = {}
    for name, sc, rays in (("injected", stoch, (d, p, w, tf, unit)), ("sampled", stoch, (d, p, w, tf)), ("fixed", fixed, (d, p, w, tf))):
        hb = HipTraceBackend(device=0, seed=spec["seed"], capture_exits=1)
        hb.BeginSession(sc, rd, scenes.wl_discrete(spec["wavelength"]), n)
        st = hb.TraceLayer(n, rays)
        ex = hb.DrainExits()
        hb.EndSession()
        out[name] = (hb.last_sample_counts()[0], int(st.exit_count), np.sort(ex["weight"]))
        hb.close()
    assert out["injected"][0] == 0                              # a host-supplied crystal consumes no MakeCrystal draw
    assert out["sampled"][0] == n // 32                          # the entry's own (stochastic) crystal: one instance per 32 rays
    assert out["injected"][1] == out["fixed"][1] > n and np.array_equal(out["injected"][2], out["fixed"][2])
