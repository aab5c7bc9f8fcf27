"""Numeric expectations the REFERENCE'S OWN TESTS hold for the host-side pieces of the path that cannot be compiled here
(lat_lut.cpp, lens_proj_build.hpp, geo3d_closedform.cpp, illuminant.cpp: their translation units need nlohmann-json >= 3.4 /
spdlog, which this image lacks and which may not be stood in for), transcribed and run against BOTH restatements — the
product's host builders (libhalo_hip.so `halo_host_*`) and the oracle's (`ho_*`).  No GPU.

Each test names the reference test it transcribes (file:line under /root/reference/test); the analytic targets, bin layouts,
sample counts and tolerances are the reference's.  Sampling goes through the same inverse-CDF lookup as the device path
(`invert_lat_lut`, pinned bit-exact to the reference's header in test_oracle_vs_ref.py), here vectorised in numpy.
"""
import ctypes as C
import math

import numpy as np
import pytest

from ice_halo_sim_amd import abi, backend
from tests import _libs

DEG = math.pi / 180.0
N_SAMPLES = 500_000   # SphericalSamplingTest::kSampleCount, test_rng.cpp:194


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def build_lut(which, dtype, center, spread):
    """(theta, cdf, flip) float32[257] from the product's or the oracle's BuildLatLut restatement (lat_lut.cpp:74-204)."""
    d = abi.HaloDist(dtype, float(center), float(spread))
    t = [np.zeros(257, np.float32) for _ in range(3)]
    if which == "product":
        assert backend.load_library().halo_host_build_lat_lut(C.byref(d), *[fptr(x) for x in t]) == 0
    else:
        _libs.oracle().ho_build_lat_lut(C.byref(d), *[fptr(x) for x in t])
    return t


def invert(theta, cdf, xi):
    """lm_pcg::invert_lat_lut (pcg_shared.h:311-360), vectorised: clamp, bracket, interpolate."""
    xi = np.clip(xi.astype(np.float32), cdf[0], cdf[-1])
    lo = np.clip(np.searchsorted(cdf, xi, side="right") - 1, 0, len(cdf) - 2)
    c0, c1 = cdf[lo], cdf[lo + 1]
    den = c1 - c0
    w = np.where(den > 0, (xi - c0) / np.where(den > 0, den, 1), 0).astype(np.float32)
    return theta[lo] + w * (theta[lo + 1] - theta[lo])


def sample(which, dtype, center, spread, n, seed):
    """n draws of (colatitude [rad], flip) the way sample_lat_lon_roll's LUT path makes them (pcg_shared.h:392-440)."""
    theta, cdf, flip = build_lut(which, dtype, center, spread)
    rng = np.random.default_rng(seed)
    colat = invert(theta, cdf, rng.random(n, dtype=np.float32))
    span = theta[-1] - theta[0]
    b = np.clip(((colat - theta[0]) / span * 256).astype(np.int64) if span > 0 else np.zeros(n, np.int64), 0, 255)
    flipped = rng.random(n, dtype=np.float32) < flip[b]
    return colat.astype(np.float64), flipped


def folded_gauss_density(centers_deg, zen, sigma):
    """SphericalSamplingTest::TheoreticalDensity, test_rng.cpp:200-224."""
    th, t0, s = np.asarray(centers_deg) * DEG, zen * DEG, sigma * DEG
    g = np.exp(-0.5 * (th - t0) ** 2 / s ** 2) + np.exp(-0.5 * (-th - t0) ** 2 / s ** 2) + np.exp(-0.5 * (2 * np.pi - th - t0) ** 2 / s ** 2)
    d = g * np.sin(th)
    return d / (d.sum() * (centers_deg[1] - centers_deg[0]) * DEG)


WHICH = ["product", "oracle"]


@pytest.mark.parametrize("which", WHICH)
@pytest.mark.parametrize("zen,sigma", [(0, 5), (1, 2), (10, 5), (45, 5), (90, 5)])
def test_gauss_latitude_has_the_jacobian_corrected_density(which, zen, sigma):
    """SphericalSamplingTest.JacobianCorrectedDistribution, test_rng.cpp:255-304: p(theta) ~ G_folded(theta - zen, sigma) sin(theta);
    20 bins over zen +/- 4 sigma, every bin within 2 sigma of the mean within 5 % of the analytic density."""
    lo, hi = max(0.0, zen - 4 * sigma), min(180.0, zen + 4 * sigma)
    edges = np.linspace(lo, hi, 21)
    centers = 0.5 * (edges[:-1] + edges[1:])
    colat, _ = sample(which, abi.DIST_GAUSS, 90.0 - zen, sigma, N_SAMPLES, 42)
    counts, _ = np.histogram(colat / DEG, edges)
    theo = folded_gauss_density(centers, zen, sigma)
    obs = counts / (N_SAMPLES * (edges[1] - edges[0]) * DEG)
    sel = (np.abs(centers - zen) <= 2 * sigma) & (theo >= 1e-6)
    assert sel.any() and (np.abs(obs[sel] - theo[sel]) / theo[sel] < 0.05).all(), np.abs(obs[sel] - theo[sel]) / theo[sel]


@pytest.mark.parametrize("which", WHICH)
@pytest.mark.parametrize("case", ["gauss_0_180", "gauss_90_180", "uniform_full"])
def test_wide_distributions_are_uniform_on_the_sphere(which, case):
    """SphericalSamplingTest.LargeSigmaUniformDistribution (:308-352) and UniformLatitudeJacobianCorrection (:357-408): a Gaussian
    of sigma 180 deg, and a full-range uniform latitude through the PARAMETERISED path, both give p(theta) = sin(theta)/2 —
    18 bins of 10 deg, each within 5 %."""
    if case == "uniform_full":
        colat, _ = sample(which, abi.DIST_UNIFORM, 90.0, 360.0, N_SAMPLES, 77)
    else:
        zen = 0.0 if case == "gauss_0_180" else 90.0
        colat, _ = sample(which, abi.DIST_GAUSS, 90.0 - zen, 180.0, N_SAMPLES, 99)
    edges = np.linspace(0, 180, 19)
    centers = 0.5 * (edges[:-1] + edges[1:])
    counts, _ = np.histogram(colat / DEG, edges)
    obs = counts / (N_SAMPLES * 10 * DEG)
    theo = np.sin(centers * DEG) / 2
    assert (np.abs(obs - theo) / theo < 0.05).all()


@pytest.mark.parametrize("which", WHICH)
@pytest.mark.parametrize("zen,sigma", [(0, 5), (45, 5), (90, 5), (0, 30), (0, 180), (90, 180)])
def test_gauss_latitude_mean_colatitude(which, zen, sigma):
    """SphericalSamplingTest.MeanVarianceAccuracy, test_rng.cpp:412-482: E[theta] under G_folded x sin by quadrature over 7 images;
    sample mean within 0.1 deg (0.5 deg for sigma >= 90)."""
    t0, s = zen * DEG, sigma * DEG
    th = 0.001 + np.arange(10000) * math.pi / 10000
    g = np.zeros_like(th)
    for k in range(-3, 4):
        g += np.exp(-0.5 * (th - t0 - 2 * k * math.pi) ** 2 / s ** 2) + np.exp(-0.5 * (th + t0 - 2 * k * math.pi) ** 2 / s ** 2)
    w = g * np.sin(th)
    expected = (th * w).sum() / w.sum()
    colat, _ = sample(which, abi.DIST_GAUSS, 90.0 - zen, sigma, N_SAMPLES, 123)
    assert abs(colat.mean() - expected) / DEG <= (0.5 if sigma >= 90 else 0.1)


def _raw_draws(dtype, mean, std, n, seed):
    """RandomNumberGenerator::Get (math.cpp:418-444) for the two non-Gaussian latitude proposals, in degrees."""
    u = np.random.default_rng(seed).random(n)
    if dtype == abi.DIST_ZIGZAG:
        return np.abs(std * np.sin(2 * np.pi * u) + mean)
    sgn = np.where(u < 0.5, -1.0, 1.0)
    return mean - std * sgn * np.log(np.maximum(1 - 2 * np.abs(u - 0.5), 1e-30))


@pytest.mark.parametrize("which", WHICH)
def test_zigzag_latitude_range_and_jacobian_shift(which):
    """SphericalSamplingTest.ZigzagBasicSampling (:555-574): zigzag(0, 30) keeps every latitude inside [-35, 35] deg;
    ZigzagJacobianCorrection (:577-678): for (0,30), (45,10), (80,5) at least 90 % of 1 M samples fall into the proposal's
    colatitude range +/- 3 deg and the sin(theta) weighting leaves the mean no farther from the equator than the bare proposal
    (+ 0.5 deg)."""
    colat, _ = sample(which, abi.DIST_ZIGZAG, 0.0, 30.0, 10_000, 42)
    lat = 90.0 - colat / DEG
    assert lat.min() >= -35.0 and lat.max() <= 35.0
    for mean, std in ((0.0, 30.0), (45.0, 10.0), (80.0, 5.0)):
        lat_lo, lat_hi = max(abs(mean) - std, 0.0), abs(mean) + std
        lo, hi = max(0.0, 90.0 - lat_hi - 3.0), min(180.0, 90.0 - lat_lo + 3.0)
        edges = np.linspace(lo, hi, 21)
        centers = 0.5 * (edges[:-1] + edges[1:])
        colat, _ = sample(which, abi.DIST_ZIGZAG, mean, std, 1_000_000, 123)
        counts, _ = np.histogram(colat / DEG, edges)
        assert counts.sum() / 1e6 > 0.90
        observed = (centers * counts).sum() / (counts.sum() + 1e-10)
        proposal = np.clip(90.0 - _raw_draws(abi.DIST_ZIGZAG, mean, std, 100_000, 5), 0, 180).mean()
        assert abs(observed - 90.0) < abs(proposal - 90.0) + 0.5


@pytest.mark.parametrize("which", WHICH)
def test_laplacian_latitude(which):
    """SphericalSamplingTest.LaplacianBasicSampling (:717-739: mean latitude of laplacian(0, 5) within 1 deg of 0, all inside
    [-90, 90]), LaplacianJacobianCorrection (:743-795, the zigzag method on (0,5), (45,3), (80,2)), LaplacianHeavierTailThanGaussian
    (:798-851: more mass beyond 3 scales than the Gaussian of the same scale) and LaplacianTightEnvelopeExactness (:854-946:
    laplacian(90, 5) at the pole has density exp(-theta/b) sin(theta): mean and std within 2 %, eight quantiles within 1 deg)."""
    colat, _ = sample(which, abi.DIST_LAPLACIAN, 0.0, 5.0, 50_000, 42)
    lat = 90.0 - colat / DEG
    assert lat.min() >= -90.0 and lat.max() <= 90.0 and abs(lat.mean()) <= 1.0
    for mean, std in ((0.0, 5.0), (45.0, 3.0), (80.0, 2.0)):
        colat, _ = sample(which, abi.DIST_LAPLACIAN, mean, std, 1_000_000, 999)
        observed = colat.mean() / DEG
        proposal = np.clip(90.0 - _raw_draws(abi.DIST_LAPLACIAN, mean, std, 100_000, 6), 0, 180).mean()
        assert abs(observed - 90.0) < abs(proposal - 90.0) + 0.5
    far = {}
    for name, dt in (("lap", abi.DIST_LAPLACIAN), ("gauss", abi.DIST_GAUSS)):
        colat, _ = sample(which, dt, 0.0, 5.0, 200_000, 777)
        far[name] = int((np.abs(90.0 - colat / DEG) > 15.0).sum())
    assert far["lap"] > far["gauss"]
    colat, _ = sample(which, abi.DIST_LAPLACIAN, 90.0, 5.0, N_SAMPLES, 1234)
    colat.sort()
    b = 5.0 * DEG
    dt = math.pi / 20000
    th = (np.arange(20000) + 0.5) * dt
    w = np.exp(-th / b) * np.sin(th) * dt
    cdf = np.concatenate([[0.0], np.cumsum(w)])
    norm = cdf[-1]
    mean = (th * w).sum() / norm
    std = math.sqrt((th * th * w).sum() / norm - mean * mean)
    assert abs(colat.mean() - mean) <= 0.02 * mean and abs(colat.std() - std) <= 0.02 * std
    for q in (0.05, 0.10, 0.25, 0.50, 0.75, 0.90, 0.95, 0.99):
        idx = int(np.searchsorted(cdf, q * norm, side="left"))
        frac = (q * norm - cdf[idx - 1]) / (cdf[idx] - cdf[idx - 1])
        assert abs(colat[int(q * N_SAMPLES)] - (idx - 1 + frac) * dt) <= 1.0 * DEG, q


@pytest.mark.parametrize("which", WHICH)
def test_fold_flip_balance(which):
    """FoldRollFlipBalanceTest.KGaussianFlipBalance (test_rng.cpp:1028-1080): gauss(latitude 90, sigma 180) folds about half of its
    draws over the pole (flip fraction in [0.30, 0.70]) and the flipped / unflipped groups have the same mean latitude (0.02 rad);
    AxisDistEquivalence (:1087-1150): the mirror-image centre -90 flips the majority (> 40 %) and gives the same latitude
    distribution as +90 (means within 0.02 rad)."""
    colat, flipped = sample(which, abi.DIST_GAUSS, 90.0, 180.0, 900_000, 42)
    lat = math.pi / 2 - colat
    frac = flipped.mean()
    assert 0.30 <= frac <= 0.70
    assert abs(lat[flipped].mean() - lat[~flipped].mean()) <= 0.02
    colat_n, flipped_n = sample(which, abi.DIST_GAUSS, -90.0, 180.0, 900_000, 42)
    assert flipped_n.mean() > 0.40
    assert abs((math.pi / 2 - colat_n).mean() - lat.mean()) <= 0.02


def test_invert_lat_lut_literal_cases():
    """InvertLatLutTest.* (test_rng.cpp:1373-1450) on the oracle's invert_lat_lut: linear CDF -> linear inverse (1e-5), clamping at
    lifted endpoints, a flat CDF bin stays finite, convex CDF t^2 -> sqrt inverse (2e-3) and monotone."""
    O = _libs.oracle()
    half_pi = np.float32(math.pi / 2)
    t = np.arange(257, dtype=np.float32) / np.float32(256)
    theta, cdf = (half_pi * t).astype(np.float32), t.copy()
    for xi in (0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0):
        assert abs(O.ho_invert_lat_lut(xi, fptr(theta), fptr(cdf), 257) - xi * half_pi) <= 1e-5
    th5 = np.array([0.0, 0.25, 0.5, 0.75, 1.0], np.float32)
    c5 = np.array([0.1, 0.3, 0.6, 0.8, 0.95], np.float32)
    assert O.ho_invert_lat_lut(-1.0, fptr(th5), fptr(c5), 5) == 0.0 and O.ho_invert_lat_lut(2.0, fptr(th5), fptr(c5), 5) == 1.0
    assert O.ho_invert_lat_lut(np.float32(0.1), fptr(th5), fptr(c5), 5) == 0.0 and O.ho_invert_lat_lut(np.float32(0.95), fptr(th5), fptr(c5), 5) == 1.0
    flat = np.array([0.0, 0.5, 0.5, 0.5, 1.0], np.float32)
    got = O.ho_invert_lat_lut(0.5, fptr(th5), fptr(flat), 5)
    assert math.isfinite(got) and 0.0 <= got <= 1.0
    cdf2 = (t * t).astype(np.float32)
    prev = -1.0
    for k in range(101):
        got = O.ho_invert_lat_lut(np.float32(k / 100.0), fptr(theta), fptr(cdf2), 257)
        assert got >= prev - 1e-6
        prev = got
    assert abs(O.ho_invert_lat_lut(0.25, fptr(theta), fptr(cdf2), 257) - half_pi * 0.5) <= 2e-3


@pytest.mark.parametrize("which", WHICH)
def test_illuminant_spd_anchors(which):
    """IlluminantSpd.* (test/unit-correctness/config/test_json.cpp:140-185): every D illuminant and A are 100 at 560 nm (0.5 / 0.1),
    E is 1 everywhere in range, all are 0 outside the tabulated range, the D series is positive across the visible range.
    (util/illuminant.cpp itself cannot be compiled here: util/illuminant_data.hpp:4,21 needs nlohmann-json's
    NLOHMANN_JSON_SERIALIZE_ENUM.)"""
    if which == "oracle":
        O = _libs.oracle()
        O.ho_illuminant_spd.restype = C.c_float
        O.ho_illuminant_spd.argtypes = [C.c_int, C.c_float]
        spd = lambda ill, wl: O.ho_illuminant_spd(abi.ILLUM[ill], float(wl))
    else:
        L = backend.load_library()
        spd = lambda ill, wl: L.halo_host_illuminant_spd(abi.ILLUM[ill], float(wl))
    assert abs(spd("D65", 560.0) - 100.0) <= 0.5 and abs(spd("D50", 560.0) - 100.0) <= 0.5 and abs(spd("A", 560.0) - 100.0) <= 0.1
    for wl in (380.0, 550.0, 780.0):
        assert abs(spd("E", wl) - 1.0) <= 1e-5
    for ill, wl in (("D65", 200.0), ("D65", 900.0), ("A", 200.0), ("E", 200.0)):
        assert abs(spd(ill, wl)) <= 1e-5
    for ill in ("D50", "D55", "D65", "D75"):
        assert all(spd(ill, wl) > 0 for wl in range(380, 781, 10))


def test_product_and_oracle_illuminant_and_pool_agree_bit_for_bit():
    """Two independent restatements of GetIlluminantSpd (util/illuminant.cpp:113-134) and ComputeWlPool (wl_pool.hpp:67-91): same
    bits for every illuminant on a 1 nm grid, and the product's pool entries = the oracle's SPD / refractive index / CMF at the
    M mid-point wavelengths of [380, 780] nm."""
    L, O = backend.load_library(), _libs.oracle()
    O.ho_illuminant_spd.restype = C.c_float
    O.ho_illuminant_spd.argtypes = [C.c_int, C.c_float]
    for name, ill in abi.ILLUM.items():
        a = np.array([L.halo_host_illuminant_spd(ill, float(w)) for w in np.arange(290.0, 840.0, 1.0)], np.float32)
        b = np.array([O.ho_illuminant_spd(ill, float(w)) for w in np.arange(290.0, 840.0, 1.0)], np.float32)
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), name
    for m in (31, 64, 255):
        wl = abi.HaloWl(0.0, 0.0, abi.ILLUM["D65"], m)
        buf = np.zeros((m, 5), np.float32)
        assert L.halo_host_wl_pool(C.byref(wl), fptr(buf), m) == m
        lam = np.float32(380.0) + (np.arange(m, dtype=np.float32) + np.float32(0.5)) * np.float32(400.0) / np.float32(m)   # wl_pool.hpp:78, in float
        spd = np.array([O.ho_illuminant_spd(abi.ILLUM["D65"], float(x)) for x in lam], np.float32)
        assert (buf[:, 1].view(np.uint32) == spd.view(np.uint32)).all() and (buf[:, 0] > 1.30).all() and (buf[:, 0] < 1.33).all()


@pytest.mark.parametrize("which", WHICH)
@pytest.mark.parametrize("lens,fov", [(abi.LENS_LINEAR, 60.0), (abi.LENS_LINEAR, 90.0), (abi.LENS_FISHEYE_EQUAL_AREA, 180.0),
                                      (abi.LENS_FISHEYE_EQUAL_AREA, 120.0), (abi.LENS_FISHEYE_EQUIDISTANT, 180.0),
                                      (abi.LENS_FISHEYE_STEREOGRAPHIC, 180.0), (abi.LENS_FISHEYE_ORTHOGRAPHIC, 180.0),
                                      (abi.LENS_FISHEYE_ORTHOGRAPHIC, 90.0)])
def test_build_proj_params_maps_field_angles_to_the_analytic_image_radius(which, lens, fov):
    """BuildProjParams + ProjectExitToPixel against the lens DEFINITIONS the reference's scale formulas encode
    (lens_proj_build.hpp:24-64; doc/configuration.md `fov` = full angle across the short image side): light arriving from the view
    direction lands on the image centre, and light arriving at field angle t off that axis lands at radius
    (short_side / 2) g(t) / g(fov/2) with g = tan (linear), sin(t/2) (equal area), t (equidistant), tan(t/2) (stereographic),
    sin (orthographic) — for any position angle around the axis, under a rotated camera (az 42, el 60, roll 15: the view of
    LmProj.ProjectExitPerTypeRotatedView, test_projection.cpp:704-737) and the default one.  The params come from the product's
    or the oracle's builder; the projection is the reference's own lm_proj::ProjectExitToPixel when oracle/_ref is built."""
    from ice_halo_sim_amd import scenes
    g = {abi.LENS_LINEAR: math.tan, abi.LENS_FISHEYE_EQUAL_AREA: lambda t: math.sin(t / 2), abi.LENS_FISHEYE_EQUIDISTANT: lambda t: t,
         abi.LENS_FISHEYE_STEREOGRAPHIC: lambda t: math.tan(t / 2), abi.LENS_FISHEYE_ORTHOGRAPHIC: math.sin}[lens]
    O = _libs.oracle()
    try:
        R = _libs.ref()
    except Exception:
        R = None
    w, h = 1024, 512
    for az, el, ro in ((42.0, 60.0, 15.0), (0.0, 90.0, 0.0), (-120.0, 10.0, -40.0)):
        cfg = scenes.render(lens, w, h, fov=fov, az=az, el=el, ro=ro, visible=abi.VISIBLE_FULL)
        pp = abi.ProjParams()
        if which == "product":
            assert backend.load_library().halo_host_build_proj_params(C.byref(cfg), C.byref(pp)) == 0
        else:
            O.ho_build_proj_params(C.byref(cfg), C.byref(pp))

        def pixel(d):
            if R is not None:
                out7 = np.zeros(7, np.int32)
                R.ref_project_exit_to_pixel(C.addressof(pp), float(d[0]), float(d[1]), float(d[2]), out7.ctypes.data_as(C.POINTER(C.c_int32)))
                return (out7[1], out7[2]) if out7[0] >= 1 else None
            r = O.ho_project_exit_to_pixel(C.byref(pp), float(d[0]), float(d[1]), float(d[2]))
            return (r.hits[0].px, r.hits[0].py) if r.count >= 1 else None
        a, e = az * DEG, el * DEG
        axis = np.array([math.cos(e) * math.cos(a), math.cos(e) * math.sin(a), math.sin(e)])   # sky position the camera looks at
        # light FROM sky position s travels along -s
        c = pixel(-axis)
        assert c is not None and abs(c[0] - w / 2) <= 1 and abs(c[1] - h / 2) <= 1, (c, az, el)
        u = np.cross(axis, [0.0, 0.0, 1.0] if abs(axis[2]) < 0.9 else [1.0, 0.0, 0.0])
        u /= np.linalg.norm(u)
        v = np.cross(axis, u)
        half = fov / 2 * DEG
        for frac in (0.2, 0.5, 0.8, 0.97):
            t = frac * half
            want = (h / 2) * g(t) / g(half)
            for pa in np.arange(0.0, 2 * math.pi, math.pi / 4):
                s = math.cos(t) * axis + math.sin(t) * (math.cos(pa) * u + math.sin(pa) * v)
                p = pixel(-s)
                if p is None or not (0 <= p[0] < w and 0 <= p[1] < h):
                    continue                       # outside the frame along the short side's diagonal: nothing to measure
                r = math.hypot(p[0] - w / 2, p[1] - h / 2)
                assert abs(r - want) <= 1.5, (lens, fov, az, el, frac, pa, r, want)


def test_product_prism_builder_on_the_reference_pool_and_known_configurations():
    """ClosedFormPrism.* (test/golden-analytic/core/test_closed_form_prism.cpp) on the PRODUCT's host builder:
    RegularHexagonHasExpectedInvariants (:164-187: 8 faces numbered 1..8, basal normals (0,0,+-1), side i at i x 60 deg, corners on
    the circle r = 0.5), ZeroHeightShortCircuit (:189-196: empty crystal), WellConditionedThreeWayAgreement (:246-290: on all 200
    samples of the reference's fixed pool the corner count equals the exact-integer oracle's — the verdicts in
    tests/golden/ref_shared_fixture.npz come from the reference's own test/support/exact_prism_oracle.hpp — and the number of
    present side faces equals the corner count)."""
    import os
    L = backend.load_library()
    g = abi.HaloGeomTables()
    one = np.ones(6, np.float32)
    assert L.halo_host_prism_geometry(1.0, fptr(one), C.byref(g)) == 0
    assert g.face_cnt == 8 and list(g.face_number[:8]) == [1, 2, 3, 4, 5, 6, 7, 8]
    n = np.array(g.face_n[:24], np.float32).reshape(8, 3)
    assert abs(n[0, 2] - 1.0) <= 1e-6 and abs(n[1, 2] + 1.0) <= 1e-6
    for i in range(6):
        assert np.abs(n[2 + i] - [math.cos(i * math.pi / 3), math.sin(i * math.pi / 3), 0.0]).max() <= 1e-6
    v = np.array(g.tri_v[: g.tri_cnt * 9], np.float64).reshape(-1, 3)
    assert np.abs(np.hypot(v[:, 0], v[:, 1]) - 0.5).max() <= 1e-6 and np.abs(np.abs(v[:, 2]) - 0.5).max() <= 1e-6
    assert L.halo_host_prism_geometry(0.0, fptr(one), C.byref(g)) == 0 and g.face_cnt == 0 and g.tri_cnt == 0
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_shared_fixture.npz"))
    assert len(fx["prism_pool_dist"]) == 200
    for dist, verdict in zip(fx["prism_pool_dist"], fx["prism_pool_exact"]):
        assert L.halo_host_prism_geometry(1.0, fptr(np.ascontiguousarray(dist, np.float32)), C.byref(g)) == 0
        side_numbers = [f for f in g.face_number[: g.face_cnt] if f >= 3]
        assert verdict[1] == 0 and len(side_numbers) == verdict[0] == g.face_cnt - 2
        want = [i + 3 for i in range(6) if bin(int(verdict[2 + i])).count("1") >= 2]
        assert side_numbers == want
        # top ring: its fan has corner_count - 2 triangles
        assert sum(1 for t in range(g.tri_cnt) if g.tri_face[t] == 0) == verdict[0] - 2


def test_wavelength_stream_is_decoupled_from_the_orientation_stream():
    """WlStreamDecouple.* (test/unit-correctness/core/test_wl_stream_decouple.cpp:34-99) on the product's constants and the oracle's
    stream: the per-ray wavelength draw lives in its own seed domain (seed ^ kWlStreamNonce) — the nonce is non-zero and differs
    from every other stream nonce of the device path, and the first wavelength draw never equals the orientation stream's draw at
    any of slots 0..20 for the same (seed, ray index)."""
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ice_halo_sim_amd", "csrc", "halo_device.h")).read()
    nonce = {k: int(v, 16) for k, v in re.findall(r"constexpr uint32_t (kNonce\w+) = (0x[0-9A-Fa-f]+)u;", hdr)}
    assert nonce["kNonceWl"] == 0x9E3779B9 != 0                                           # lm_pcg::kWlStreamNonce, pcg_shared.h
    assert (nonce["kNonceTransit"], nonce["kNonceGate"], nonce["kNonceGen"], nonce["kNonceShuffle"]) == (0xA5A5A5A5, 0x5A5A5A5A, 0x3C9A7F11, 0xB17CA3D9)
    assert len(set(nonce.values())) == len(nonce)
    O = _libs.oracle()
    for seed in (0, 1, 0xDEADBEEF, 0xC0FFEE01, 0x12345678):
        for gidx in (0, 42, 1000, 65535, 0x80000000):
            st = _libs.HoStream(seed ^ nonce["kNonceWl"], gidx, 0)
            wl_draw = O.ho_pcg_uniform(C.byref(st))
            assert st.slot == 1
            for slot in range(21):
                so = _libs.HoStream(seed, gidx, slot)
                assert O.ho_pcg_uniform(C.byref(so)) != wl_draw, (seed, gidx, slot)
