import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import run_session
sc, rd = scenes.config2_scene(), scenes.config2_render()
wl = scenes.wl_discrete(550.0)
for n in (100_000, 1_000_000, 4_000_000, 16_000_000):
    for bpc in (1, 2, 4, 8):
        hb = HipTraceBackend(device=0, seed=42, blocks_per_cu=bpc)
        best = 1e9
        for rep in range(4):
            st = run_session(hb, sc, rd, wl, n)
            best = min(best, st[0].kernel_ms)
        print("n=%9d blocks_per_cu=%d kernel %.3f ms (%.1f M rays/s)" % (n, bpc, best, n / best / 1e3), flush=True)
        hb.close()
