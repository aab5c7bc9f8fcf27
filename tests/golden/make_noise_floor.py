#!/usr/bin/env python3
"""Oracle-vs-oracle cross-seed noise floor for BASELINE configs[1] and configs[2] (SURVEY.md §8d "parity gate": the
reference's battery compares backends against the legacy path's OWN seed-to-seed scatter, test/e2e/_parity_metrics.py, seeds 42 / 7).
Writes tests/golden/noise_floor.json; the statistical thresholds of the multi-layer GPU tests are derived from it
(tests/golden/NOISE_FLOOR.md).  CPU only: `python tests/golden/make_noise_floor.py` (a few minutes on 8 threads)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ice_halo_sim_amd import scenes  # noqa: E402
from tests._oracle_backend import OracleBackend, run_session  # noqa: E402


def block_mean(img, k):
    h, w, c = img.shape
    return img[: h // k * k, : w // k * k].reshape(h // k, k, w // k, k, c).mean(axis=(1, 3))


def render(scene, rd, n, seed, wls):
    ob = OracleBackend(seed=seed, threads=os.cpu_count() or 8)
    cont = 0
    for wl in wls:
        st = run_session(ob, scene, rd, scenes.wl_discrete(wl), n)
        cont += sum(int(s.continuation_count) for s in st[:-1])
    img, landed = ob.ReadbackXyzAccum()
    ob.close()
    return img.astype(np.float64), landed, cont


def floor(name, scene, rd, n, wls):
    a, la, ca = render(scene, rd, n, 42, wls)
    b, lb, cb = render(scene, rd, n, 7, wls)
    out = {"config": name, "rays_per_wavelength": n, "wavelengths": wls, "resolution": [rd.width, rd.height], "seeds": [42, 7]}
    for k in (4, 8, 16):
        x, y = block_mean(a, k).ravel(), block_mean(b, k).ravel()
        out["pearson_block%d" % k] = float(np.corrcoef(x, y)[0, 1])
        out["rel_l2_block%d" % k] = float(np.linalg.norm(x - y) / np.linalg.norm(y))
    out["sum_y_dev"] = float(abs(a[..., 1].sum() / b[..., 1].sum() - 1))
    out["landed_dev"] = float(abs(la / lb - 1))
    out["continuation_dev"] = float(abs(ca / cb - 1)) if cb else 0.0
    print(json.dumps(out))
    return out


def main():
    res = []
    for n in (200_000, 1_000_000):
        res.append(floor("configs[1] single-scatter column, 1920x1080 fisheye upper", scenes.config2_scene(), scenes.config2_render(), n, [550.0]))
        res.append(floor("configs[2] plate over random column, 1920x1080 fisheye upper", scenes.config3_scene(), scenes.config2_render(), n, [550.0]))
        res.append(floor("configs[2] at 512x256 dual fisheye full sky (the e2e battery's resolution)", scenes.config3_scene(),
                         scenes.render(4, 512, 256, visible=2), n, [550.0]))
    with open(os.path.join(ROOT, "tests", "golden", "noise_floor.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
