import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ice_halo_sim_amd import abi, scenes
from tests.test_gpu_parity import run_both
sc = scenes.config2_scene()
rd = scenes.render(1, 512, 256, fov=120.0, el=30.0, visible=abi.VISIBLE_UPPER)
r = run_both(sc, rd, scenes.wl_discrete(530.0), 100_000)
eh, eo = r["eh"], r["eo"]
print("counts", len(eh), len(eo))
kh = (eh["root"].astype(np.int64) << 8) | eh["seq"]; ko = (eo["root"].astype(np.int64) << 8) | eo["seq"]
ih, io = np.argsort(kh), np.argsort(ko)
common, ch, co = np.intersect1d(kh[ih], ko[io], return_indices=True)
a, b = eh[ih][ch], eo[io][co]
print("common", len(common))
dd = np.abs(a["dir"] - b["dir"]).max(axis=1)
dw = np.abs(a["weight"] - b["weight"]) / np.maximum(np.abs(b["weight"]), 1e-12)
for q in (0.5, 0.9, 0.99, 0.999):
    print("q", q, "dd", np.quantile(dd, q), "dw", np.quantile(dw, q))
bad = dd > 2e-5
print("bad dir frac", bad.mean(), "bad w frac", (dw > 2e-4).mean())
print("bad by seq", np.bincount(a["seq"][bad], minlength=16))
print("all by seq", np.bincount(a["seq"], minlength=16))
roots_bad = np.unique(a["root"][bad])
print("n bad roots", len(roots_bad), "of", len(np.unique(a["root"])))
# first exit (seq 0) is the entry reflection: depends only on orientation + sun dir
s0 = a["seq"] == 0
print("seq0 dd quantiles", np.quantile(dd[s0], [0.5, 0.99, 0.999, 1.0]))
i = np.argmax(dd * s0)
print("worst seq0", a[i], b[i])
