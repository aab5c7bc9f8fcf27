"""tests/_parity_battery.py (the repo's restatement of the reference's parity battery, which the GPU parity tests use as their yardstick)
must reproduce what the reference's OWN modules return: tests/golden/ref_parity_metrics.json was written by
tests/golden/make_parity_metric_fixture.py importing test/e2e/_parity_metrics.py and test/e2e/_projection_battery.py in the build
container.  No GPU, no reference at run time."""
import json
import os

import numpy as np
import pytest

from tests import _parity_battery as pb

HERE = os.path.dirname(os.path.abspath(__file__))
FX = json.load(open(os.path.join(HERE, "golden", "ref_parity_metrics.json")))
DOCS = json.load(open(os.path.join(HERE, "golden", "ref_e2e_configs.json")))


def test_constants_are_the_references():
    c = FX["constants"]
    for k in ("DS_BH", "DS_BW", "T_RAW_CORR_DS", "T_ENERGY_TOL", "T_SELF_MARGIN", "T_PSNR_DB", "PROJECTION_TYPES", "FOV_BY_TYPE", "VIEW_ELEVATION",
              "RESOLUTION", "BASE_DOCUMENT"):
        assert getattr(pb, k) == c[k], k
    assert len(pb.PROJECTION_TYPES) == 11


def test_metrics_reproduce_the_references_numbers():
    assert len(FX["pairs"]) >= 6
    for p in FX["pairs"]:
        a, b = np.asarray(p["a"], np.float32), np.asarray(p["b"], np.float32)
        assert np.allclose(pb.block_mean(a.astype(np.float64), 4, 4), np.asarray(p["block_mean_a_4x4"]), rtol=1e-13, atol=0)
        assert pb.raw_corr_ds(a, b) == pytest.approx(p["raw_corr_ds_4x4"], abs=1e-12)
        assert pb.raw_corr_ds(a, b, 2, 8) == pytest.approx(p["raw_corr_ds_2x8"], abs=1e-12)
        ra, rb = np.asarray(p["render_a"], np.uint8), np.asarray(p["render_b"], np.uint8)
        want = float("inf") if p["render_psnr"] == "inf" else p["render_psnr"]
        assert pb.render_psnr(ra, rb) == pytest.approx(want, rel=1e-13)
    # the zero-variance convention: a constant pair reads 0.0, not NaN
    const = [p for p in FX["pairs"] if len(set(np.asarray(p["a"]).ravel())) == 1]
    assert const and all(p["raw_corr_ds_4x4"] == 0.0 for p in const)


def test_projection_documents_are_the_references():
    base = DOCS[pb.BASE_DOCUMENT]
    for lens in pb.PROJECTION_TYPES:
        assert pb.projection_config(base, lens) == FX["projection_documents"][lens], lens
    with pytest.raises(ValueError):
        pb.projection_config(base, "pinhole")


def test_check_applies_the_references_bars():
    rng = np.random.default_rng(1)
    img = rng.random((16, 32, 3)) * 10
    out = pb.check(img, img * 1.01, img * 1.0, img * 0.99)
    assert out["corr"] > 0.999 and out["energy"] == pytest.approx(1 - 1 / 1.01, rel=1e-9)
    with pytest.raises(AssertionError):
        pb.check(img, img * 1.06)                                    # G2: 5 % energy
    with pytest.raises(AssertionError):
        pb.check(img, rng.random((16, 32, 3)) * 10)                  # G1: correlation
