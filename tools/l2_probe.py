import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import OracleBackend, run_session
sc, rd = scenes.config2_scene(), scenes.config2_render()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
hb = HipTraceBackend(device=0, seed=2024)
ob = OracleBackend(seed=2024, threads=min(os.cpu_count(), 128))
run_session(hb, sc, rd, scenes.wl_discrete(570.0), n)
run_session(ob, sc, rd, scenes.wl_discrete(570.0), n)
ih, lh = hb.ReadbackXyzAccum(); io, lo = ob.ReadbackXyzAccum()
def rl2(a, b): return float(np.linalg.norm((a - b).ravel().astype(np.float64)) / np.linalg.norm(b.ravel().astype(np.float64)))
y = io[..., 1]
print("landed", lh, lo, "full", rl2(ih, io))
for k in (8, 64, 512, 4096):
    thr = np.partition(y.ravel(), -k)[-k]
    dim = y < thr
    print("without top %d px: %.3e   (top px share of energy %.3f)" % (k, rl2(ih[dim], io[dim]), y[~dim].sum() / y.sum()))
d = np.abs(ih[..., 1] - io[..., 1])
idx = np.argsort(d.ravel())[-8:]
for i in idx[::-1]:
    py, px = divmod(int(i), rd.width)
    print("pix (%d,%d): hip %.3f oracle %.3f diff %.3f" % (px, py, ih[py, px, 1], io[py, px, 1], d[py, px]))
# oracle single-thread determinism check at small n
