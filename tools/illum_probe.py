"""Illuminant (D65) sessions of a deterministic crystal: the X/Y/Z one-shape kernels (exit queue, regular-prism search, X/Y/Z hit log)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import run_session
sc = scenes.config2_scene()
for (w, h, vis) in ((1920, 1080, abi.VISIBLE_UPPER), (1920, 1080, abi.VISIBLE_FULL), (512, 256, abi.VISIBLE_UPPER)):
    rd = scenes.render(1, w, h, fov=180, el=30, visible=vis)
    for n in (1_000_000, 10_000_000, 50_000_000):
        hb = HipTraceBackend(device=0, seed=42)
        best = 1e9
        for r in range(3):
            st = run_session(hb, sc, rd, scenes.wl_illuminant("D65", 31), n)
            best = min(best, sum(s.kernel_ms for s in st))
        r_ = hb.last_route()
        hb.close()
        print("D65 column %dx%d visible %d n=%3dM: kernels %7.3f ms %8.1f M rays/s  (geom %d accum %d planes %d)" % (w, h, vis, n // 1000000, best, n / best / 1e3, r_.geom_mask, r_.accum_mask, r_.plane_cnt), flush=True)
