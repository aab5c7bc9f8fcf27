"""Scale check: the reference's stochastic benchmark shape at 20 M and 200 M rays in one session — landed weight per ray and
image energy per ray must agree (nothing overflows in the two-level binned route's lists, counters or chunking)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import run_session
sc = scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8)
rd = scenes.render(7, 2048, 1024, el=0, visible=2)
res = {}
for n in (20_000_000, 200_000_000):
    hb = HipTraceBackend(device=0, seed=42)
    hb.sync(); t0 = time.perf_counter()
    st = run_session(hb, sc, rd, scenes.wl_illuminant("D65", 64), n)
    hb.sync(); dt = time.perf_counter() - t0
    img, landed = hb.ReadbackXyzAccum()
    res[n] = (landed / n, float(img[..., 1].astype(np.float64).sum()) / n, st[0].exit_count / n, st[0].launches)
    print("n=%d M: wall %.1f ms, launches %d, landed/ray %.6f, sumY/ray %.6f, exits/ray %.4f" % (n // 1_000_000, dt * 1e3, st[0].launches, *res[n][:3]), flush=True)
    hb.close()
a, b = res[20_000_000], res[200_000_000]
assert abs(a[0] - b[0]) < 2e-3 * a[0] and abs(a[1] - b[1]) < 2e-3 * a[1] and abs(a[2] - b[2]) < 1e-3 * a[2], (a, b)
print("scale check ok")
