"""Pin oracle/halo_oracle.c against the REFERENCE's own shared-math headers (oracle/_ref/libref_shared.so,
compiled from /root/reference/src/core/shared/*.h where they lie).  Bit-exact: both are host fp32 code on
the same libm with FP contraction off.  Skipped where the reference build is absent (GPU box without
a prebuilt _ref) — tests/test_oracle_fixture.py covers the same functions from committed vectors there.
"""
import ctypes as C

import numpy as np
import pytest

from ice_halo_sim_amd import abi
from tests import _libs
from tests._libs import HoGenParams, HoStream, bits, fptr, i32ptr, u32ptr

pytestmark = pytest.mark.skipif(not _libs.have_ref(), reason="oracle/_ref/libref_shared.so not built")

RNG = np.random.default_rng(20260928)


def test_pcg_hash_and_streams():
    O, R = _libs.oracle(), _libs.ref()
    xs = np.concatenate([np.array([0, 1, 2, 0xFFFFFFFF, 0x80000000], dtype=np.uint32),
                         RNG.integers(0, 2**32, 4000, dtype=np.uint32)])
    for x in xs:
        x = int(x)
        assert O.ho_pcg_hash(x) == R.ref_pcg_hash(x)
        assert bits(O.ho_u01_from_hash(x)) == bits(R.ref_u01_from_hash(x))
    for _ in range(2000):
        lo, hi, tid, seed = (int(v) for v in RNG.integers(0, 2**32, 4, dtype=np.uint32))
        if RNG.random() < 0.5:
            lo = 0xFFFFFFFF - int(RNG.integers(0, 1000))
            tid = int(RNG.integers(0, 2000))
        if RNG.random() < 0.3:
            hi = 0
        assert O.ho_pcg_advance_hi(lo, hi, tid) == R.ref_pcg_advance_hi(lo, hi, tid)
        assert O.ho_pcg_seed_with_high(seed, hi) == R.ref_pcg_seed_with_high(seed, hi)
    for _ in range(500):
        seed, gidx = (int(v) for v in RNG.integers(0, 2**32, 2, dtype=np.uint32))
        s = HoStream(seed, gidx, 0)
        r = np.array([seed, gidx, 0], dtype=np.uint32)
        for k in range(12):
            a = O.ho_pcg_uniform(C.byref(s))
            b = R.ref_pcg_uniform(u32ptr(r))
            assert bits(a) == bits(b) and s.slot == r[2]
        a = O.ho_pcg_gaussian(C.byref(s)); b = R.ref_pcg_gaussian(u32ptr(r))
        assert bits(a) == bits(b) and s.slot == r[2]
        for dtype in range(6):
            mean, std = float(RNG.normal()), float(abs(RNG.normal()) + 0.01)
            a = O.ho_pcg_get_dist(C.byref(s), dtype, mean, std)
            b = R.ref_pcg_get_dist(u32ptr(r), dtype, mean, std)
            assert bits(a) == bits(b) and s.slot == r[2]
    # nonce constants (pcg_shared.h:119-120)
    assert R.ref_wl_stream_seed(0) == 0x9E3779B9 and R.ref_geom_shape_stream_seed(0) == 0x94D049BB


def test_normalize_latitude_and_lut_lookup():
    O, R = _libs.oracle(), _libs.ref()
    for phi in np.concatenate([RNG.uniform(-20, 20, 3000), [0, np.pi / 2, -np.pi / 2, np.pi, 3 * np.pi / 2]]):
        a, fa = C.c_float(), C.c_int()
        b, fb = C.c_float(), C.c_int()
        O.ho_normalize_latitude(float(phi), C.byref(a), C.byref(fa))
        R.ref_normalize_latitude(float(phi), C.byref(b), C.byref(fb))
        assert bits(a.value) == bits(b.value) and fa.value == fb.value
    n = 257
    for _ in range(20):
        theta = np.sort(RNG.uniform(0, np.pi, 2)).astype(np.float32)
        th = np.linspace(theta[0], theta[1], n).astype(np.float32)
        cdf = np.sort(RNG.uniform(0, 1, n)).astype(np.float32)
        cdf[0], cdf[-1] = 0.0, 1.0
        cdf = np.maximum.accumulate(np.nextafter(cdf, np.float32(2)) if RNG.random() < 0.5 else cdf).astype(np.float32)
        for xi in np.concatenate([RNG.uniform(0, 1, 300), [0.0, 1.0, float(cdf[5]), float(cdf[100])]]):
            a = O.ho_invert_lat_lut(float(xi), fptr(th), fptr(cdf), n)
            b = R.ref_invert_lat_lut(float(xi), fptr(th), fptr(cdf), n)
            assert bits(a) == bits(b)
            assert O.ho_lat_lut_bin(a, fptr(th), n) == R.ref_lat_lut_bin(b, fptr(th), n)


def _gp_pair(lat_path, lat, az, roll, lut_n):
    gp = HoGenParams(lat_path, lat[0], lat[1], lut_n, az[0], az[1], az[2], roll[0], roll[1], roll[2])
    raw = np.zeros(10, dtype=np.uint32)
    f = raw.view(np.float32)
    raw[0] = lat_path; f[1] = lat[0]; f[2] = lat[1]; raw[3] = lut_n
    raw[4] = az[0]; f[5] = az[1]; f[6] = az[2]; raw[7] = roll[0]; f[8] = roll[1]; f[9] = roll[2]
    return gp, raw


def test_orientation_sampler_and_rotation():
    O, R = _libs.oracle(), _libs.ref()
    n = 257
    d = abi.HaloDist(abi.DIST_GAUSS, 0.0, 0.3)
    th = np.zeros(n, np.float32); cdf = np.zeros(n, np.float32); fl = np.zeros(n, np.float32)
    O.ho_build_lat_lut(C.byref(d), fptr(th), fptr(cdf), fptr(fl))
    for lat_path in (0, 1, 3, 6):
        for trial in range(300):
            az = (int(RNG.integers(0, 6)), float(RNG.uniform(-3, 3)), float(RNG.uniform(0, 6.3)))
            roll = (int(RNG.integers(0, 6)), float(RNG.uniform(-3, 3)), float(RNG.uniform(0, 6.3)))
            gp, raw = _gp_pair(lat_path, (float(RNG.uniform(-1.5, 1.5)), float(RNG.uniform(0, 1))), az, roll, n if lat_path == 6 else 0)
            seed, gidx = (int(v) for v in RNG.integers(0, 2**32, 2, dtype=np.uint32))
            s = HoStream(seed, gidx, 0)
            r = np.array([seed, gidx, 0], dtype=np.uint32)
            a = [C.c_float() for _ in range(3)]
            b = [C.c_float() for _ in range(3)]
            O.ho_sample_lat_lon_roll(C.byref(s), C.byref(gp), fptr(th), fptr(cdf), fptr(fl), *[C.byref(x) for x in a])
            R.ref_sample_lat_lon_roll(u32ptr(r), raw.ctypes.data, fptr(th), fptr(cdf), fptr(fl), *[C.byref(x) for x in b])
            assert [bits(x.value) for x in a] == [bits(x.value) for x in b]
            assert s.slot == r[2]
            ma = np.zeros(9, np.float32); mb = np.zeros(9, np.float32)
            O.ho_build_crystal_rotation_9(a[0].value, a[1].value, a[2].value, fptr(ma))
            R.ref_build_crystal_rotation_9(b[0].value, b[1].value, b[2].value, fptr(mb))
            assert (ma.view(np.uint32) == mb.view(np.uint32)).all()
            dw = RNG.normal(size=3).astype(np.float32)
            oa = np.zeros(3, np.float32); ob = np.zeros(3, np.float32)
            O.ho_apply_inverse_mat9(fptr(ma), fptr(dw), fptr(oa))
            R.ref_apply_inverse_mat9(fptr(mb), fptr(dw), fptr(ob))
            assert (oa.view(np.uint32) == ob.view(np.uint32)).all()


def test_samplers_triangle_cap_categorical_feistel():
    O, R = _libs.oracle(), _libs.ref()
    for _ in range(1500):
        seed, gidx = (int(v) for v in RNG.integers(0, 2**32, 2, dtype=np.uint32))
        s = HoStream(seed, gidx, 3); r = np.array([seed, gidx, 3], dtype=np.uint32)
        v9 = RNG.normal(size=9).astype(np.float32)
        pa = np.zeros(3, np.float32); pb = np.zeros(3, np.float32)
        O.ho_sample_triangle(C.byref(s), fptr(v9), fptr(pa)); R.ref_sample_triangle(u32ptr(r), fptr(v9), fptr(pb))
        assert (pa.view(np.uint32) == pb.view(np.uint32)).all()
        lon, lat, half = float(RNG.uniform(0, 6.3)), float(RNG.uniform(-1.5, 1.5)), float(RNG.uniform(0, 0.1))
        O.ho_sample_sph_cap(C.byref(s), lon, lat, half, fptr(pa)); R.ref_sample_sph_cap(u32ptr(r), lon, lat, half, fptr(pb))
        assert (pa.view(np.uint32) == pb.view(np.uint32)).all() and s.slot == r[2]
        n = int(RNG.integers(1, 65))
        w = RNG.normal(size=n).astype(np.float32)
        if RNG.random() < 0.1:
            w = -np.abs(w)
        u = float(np.float32(RNG.random()))
        assert O.ho_categorical_sample(fptr(w), n, u) == R.ref_categorical_sample(fptr(w), n, u)
    for n in [1, 2, 3, 4, 5, 16, 17, 255, 256, 257, 1000, 4097, 100003]:
        seed = int(RNG.integers(0, 2**32, dtype=np.uint32))
        idx = np.arange(n) if n <= 4097 else RNG.integers(0, n, 3000)
        out = [O.ho_feistel_bijection(int(i), n, seed) for i in idx]
        assert out == [R.ref_feistel_bijection(int(i), n, seed) for i in idx]
        if n <= 4097:
            assert sorted(out) == list(range(n))  # bijection


def test_fresnel_and_slab():
    O, R = _libs.oracle(), _libs.ref()
    for _ in range(5000):
        delta, rr = float(np.float32(RNG.uniform(0, 3))), float(np.float32(RNG.choice([1.31, 1 / 1.31, 1.5, 0.7])))
        assert bits(O.ho_reflect_ratio(delta, rr)) == bits(R.ref_reflect_ratio(delta, rr))
        d = RNG.normal(size=3).astype(np.float32); p = RNG.normal(size=3).astype(np.float32)
        n = RNG.normal(size=3).astype(np.float32); n /= np.linalg.norm(n)
        fd = float(np.float32(RNG.normal()))
        assert bits(O.ho_slab_face_t(fptr(d), fptr(p), fptr(n), fd)) == bits(R.ref_slab_face_t(fptr(d), fptr(p), fptr(n), fd))


def _render(lens, w, h, fov=120.0, az=0.0, el=30.0, ro=0.0, visible=abi.VISIBLE_FULL, overlap=0.0, shift=(0, 0)):
    r = abi.HaloRender()
    r.lens_type, r.fov, r.width, r.height = lens, fov, w, h
    r.lens_shift[0], r.lens_shift[1] = shift
    r.view_az, r.view_el, r.view_ro, r.visible, r.overlap = az, el, ro, visible, overlap
    return r


def test_projection_all_lenses():
    O, R = _libs.oracle(), _libs.ref()
    assert R.ref_proj_params_size() == C.sizeof(abi.ProjParams) == 76
    dirs = RNG.normal(size=(1500, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs = np.concatenate([dirs, np.array([[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, 1, 0], [-1, 0, 0]], np.float32)])
    for lens in range(11):
        for (w, h) in [(1920, 1080), (512, 256), (2048, 1024)]:
            for visible in (0, 1, 2):
                overlap = 0.0872 if lens in (4, 5, 6, 9) and visible == 2 else 0.0
                cfg = _render(lens, w, h, fov=float(RNG.choice([40, 120, 180])), az=float(RNG.uniform(-180, 180)),
                              el=float(RNG.uniform(-60, 60)), ro=float(RNG.uniform(-30, 30)), visible=visible,
                              overlap=overlap, shift=(int(RNG.integers(-50, 50)), int(RNG.integers(-50, 50))))
                pp = abi.ProjParams()
                O.ho_build_proj_params(C.byref(cfg), C.byref(pp))
                out7 = np.zeros(7, np.int32)
                for d in dirs[:: (1 if (w, h) == (512, 256) else 7)]:
                    a = O.ho_project_exit_to_pixel(C.byref(pp), float(d[0]), float(d[1]), float(d[2]))
                    R.ref_project_exit_to_pixel(C.addressof(pp), float(d[0]), float(d[1]), float(d[2]), i32ptr(out7))
                    assert a.count == out7[0]
                    for k in range(a.count):
                        assert (a.hits[k].px, a.hits[k].py, a.hits[k].bump_landed) == tuple(out7[1 + 3 * k: 4 + 3 * k])


def test_cmf_and_prism_exact_oracle():
    O, R = _libs.oracle(), _libs.ref()
    for wl in np.concatenate([np.arange(350, 840, 0.5), [359.4, 359.5, 830.4, 830.5]]):
        x, y, z = C.c_float(), C.c_float(), C.c_float()
        O.ho_cmf(float(wl), C.byref(x), C.byref(y), C.byref(z))
        ref = np.zeros(3, np.float32)
        R.ref_spectrum_to_xyz(float(wl), 1.0, fptr(ref))
        assert (bits([x.value, y.value, z.value]) == ref.view(np.uint32)).all()
    # closed-form prism vs the reference's exact-integer corner oracle (test/support/exact_prism_oracle.hpp):
    # corner count + which faces are present, on well-conditioned random draws (sigma 0.2 like the reference pool)
    agree = 0
    for _ in range(400):
        dist = (1.0 + 0.2 * RNG.normal(size=6)).astype(np.float32)
        out8 = np.zeros(8, np.int32)
        R.ref_exact_prism(fptr(dist), i32ptr(out8))
        cx = np.zeros(12, np.float32); cy = np.zeros(12, np.float32); present = np.zeros(8, np.int32)
        n = O.ho_prism_corner_ring(1.0, fptr(dist), fptr(cx), fptr(cy), i32ptr(present))
        assert out8[1] == 0
        exact_present = [bin(int(m)).count("1") >= 2 for m in out8[2:8]]
        if n == out8[0] and list(present[2:8] != 0) == exact_present:
            agree += 1
    assert agree >= 396  # near-degenerate draws may legitimately differ inside the merge tolerance


def test_consumer_color_pipeline_and_neumaier():
    """util/color_space.cpp (compiled from the reference as a second TU of _ref) and NeumaierAdd (accum_shared.h:70)."""
    O, R = _libs.oracle(), _libs.ref()
    for _ in range(4000):
        x = (RNG.random(3) * RNG.choice([0.01, 1.0, 3.0])).astype(np.float32)
        a, b = np.zeros(3, np.float32), np.zeros(3, np.float32)
        O.ho_gamut_clip_xyz(fptr(x), fptr(a)); R.ref_gamut_clip_xyz(fptr(x), fptr(b))
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
        ra, rb = np.zeros(3, np.float32), np.zeros(3, np.float32)
        O.ho_xyz_to_linear_rgb(fptr(a), fptr(ra)); R.ref_xyz_to_linear_rgb(fptr(b), fptr(rb))
        assert (ra.view(np.uint32) == rb.view(np.uint32)).all()
        v = float(np.float32(RNG.random()))
        assert bits(O.ho_linear_to_srgb(v)) == bits(R.ref_linear_to_srgb(v))
        s1, c1 = C.c_float(float(np.float32(RNG.normal() * 100))), C.c_float(float(np.float32(RNG.normal() * 1e-4)))
        s2, c2 = C.c_float(s1.value), C.c_float(c1.value)
        d = float(np.float32(RNG.normal()))
        O.ho_neumaier_add(C.byref(s1), C.byref(c1), d); R.ref_neumaier_add(C.byref(s2), C.byref(c2), d)
        assert bits(s1.value) == bits(s2.value) and bits(c1.value) == bits(c2.value)
