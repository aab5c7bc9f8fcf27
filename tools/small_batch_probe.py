import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import run_session
sc, rd = scenes.config2_scene(), scenes.config2_render()
wls = [scenes.wl_discrete(w) for w in scenes.CONFIG_WAVELENGTHS_9]
for mode in (0, 1):
    hb = HipTraceBackend(device=0, seed=42, **{"async": mode})
    for n in (10_000, 100_000, 1_000_000, 4_000_000):
        for rep in range(2):
            hb.sync(); t0 = time.perf_counter()
            for k in range(5):
                for wl in wls:
                    run_session(hb, sc, rd, wl, n)
            hb.sync(); dt = time.perf_counter() - t0
        st = hb.collect_stats()
        print("async=%d n=%8d: %.3f ms per session wall, %.3f ms kernel, %.1f M rays/s" % (mode, n, dt * 1e3 / 45, st.kernel_ms / max(st.launches, 1), 45 * n / dt / 1e6), flush=True)
    hb.close()
