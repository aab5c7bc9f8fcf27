"""The reference's parity battery, restated (test infrastructure): the metrics of test/e2e/_parity_metrics.py:27-79 and the
per-projection scaffolding of test/e2e/_projection_battery.py:43-124 — block-mean-downsampled Pearson correlation of the raw XYZ
buffer (G1), the energy ratio (G2), cross-seed self-consistency (G3) and the render's PSNR, with the reference's own thresholds.

Pinned: tests/golden/ref_parity_metrics.json holds the numbers the REFERENCE's modules return on committed inputs (and the eleven
per-lens documents its generator writes), produced by tests/golden/make_parity_metric_fixture.py in the build container;
tests/test_parity_battery.py requires this restatement to reproduce them.
"""
import copy
import math

import numpy as np

DS_BH = DS_BW = 4                       # _parity_metrics.py:22-23
T_RAW_CORR_DS = 0.95                    # _projection_battery.py:96-99 (G1), energy (G2), self-consistency margin (G3), PSNR
T_ENERGY_TOL = 0.05
T_SELF_MARGIN = 0.02
T_PSNR_DB = 13.0
PROJECTION_TYPES = ["linear", "fisheye_equal_area", "fisheye_equidistant", "fisheye_stereographic", "fisheye_orthographic",
                    "dual_fisheye_equal_area", "dual_fisheye_equidistant", "dual_fisheye_stereographic", "dual_fisheye_orthographic",
                    "rectangular", "globe"]
FOV_BY_TYPE = {"linear": 90.0, "fisheye_equal_area": 120.0, "fisheye_equidistant": 120.0, "fisheye_stereographic": 120.0,
               "fisheye_orthographic": 120.0, "dual_fisheye_equal_area": 180.0, "dual_fisheye_equidistant": 180.0,
               "dual_fisheye_stereographic": 180.0, "dual_fisheye_orthographic": 180.0, "rectangular": 90.0, "globe": 90.0}
VIEW_ELEVATION = 20.0
RESOLUTION = [512, 256]
BASE_DOCUMENT = "dual_fisheye_ref"      # test/e2e/configs/dual_fisheye_ref.json (kept in tests/golden/ref_e2e_configs.json)


def block_mean(buf, bh=DS_BH, bw=DS_BW):
    h, w, c = buf.shape
    assert h % bh == 0 and w % bw == 0, (buf.shape, bh, bw)
    return buf.reshape(h // bh, bh, w // bw, bw, c).mean(axis=(1, 3))


def raw_corr_ds(a, b, bh=DS_BH, bw=DS_BW):
    """G1: Pearson over the block means of all three channels flattened; 0.0 on zero variance / NaN"""
    xa = block_mean(np.asarray(a, np.float64), bh, bw).ravel()
    xb = block_mean(np.asarray(b, np.float64), bh, bw).ravel()
    if xa.std() == 0.0 or xb.std() == 0.0:
        return 0.0
    c = float(np.corrcoef(xa, xb)[0, 1])
    return 0.0 if math.isnan(c) else c


def render_psnr(a, b):
    ra, rb = np.asarray(a), np.asarray(b)
    if ra.shape != rb.shape:
        raise ValueError("render shape mismatch: %s vs %s" % (ra.shape, rb.shape))
    diff = ra.astype(float) - rb.astype(float)
    mse = float((diff * diff).mean())
    return float("inf") if mse == 0.0 else 10.0 * math.log10((255.0 * 255.0) / mse)


def energy_dev(a, b):
    """G2: |sum Y_a / sum Y_b - 1| on the raw buffers (same ray_num on both sides)"""
    return abs(float(np.asarray(a, np.float64)[..., 1].sum()) / float(np.asarray(b, np.float64)[..., 1].sum()) - 1.0)


def projection_config(base_doc, lens_type):
    """write_projection_config (_projection_battery.py:102-124): the baseline with only lens, resolution and view elevation swapped"""
    if lens_type not in FOV_BY_TYPE:
        raise ValueError("unknown lens type %r" % (lens_type,))
    cfg = copy.deepcopy(base_doc)
    render = cfg["render"][0]
    render["lens"] = {"type": lens_type, "fov": FOV_BY_TYPE[lens_type]}
    render["resolution"] = list(RESOLUTION)
    render.setdefault("view", {})["elevation"] = VIEW_ELEVATION
    return cfg


def check(backend_a, legacy_a, backend_b=None, legacy_b=None, rgb_backend=None, rgb_legacy=None):
    """The battery on one scene: images of the backend under test and of the yardstick under seed A (and B for G3).  Returns the
    measured values; raises AssertionError like the reference's tests do."""
    out = {"corr": raw_corr_ds(backend_a, legacy_a), "energy": energy_dev(backend_a, legacy_a)}
    assert out["corr"] >= T_RAW_CORR_DS, out
    assert out["energy"] <= T_ENERGY_TOL, out
    if backend_b is not None and legacy_b is not None:
        out["self_backend"], out["self_legacy"] = raw_corr_ds(backend_a, backend_b), raw_corr_ds(legacy_a, legacy_b)
        assert out["self_backend"] >= out["self_legacy"] - T_SELF_MARGIN, out
    if rgb_backend is not None:
        out["psnr"] = render_psnr(rgb_backend, rgb_legacy)
        assert out["psnr"] >= T_PSNR_DB, out
    return out
