"""Do the bandwidth-bound accumulation passes of one engine overlap with the trace kernel of ANOTHER engine on the same GPU?  N engines (own
streams, own buffers), each tracing the same workload from its own host thread; aggregate rays/s against one engine alone.
usage: python tools/two_engines_probe.py [cfg: 1 | light | 4] [engines] [sessions per engine]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import run_session

which = sys.argv[1] if len(sys.argv) > 1 else "1"
n_eng = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
wk = bench.workload({"1": "1", "light": "ref:bench_light_single_ms", "4": "4", "2": "2"}[which])
sc, rd, wls, n = wk["scene"], wk["render"], wk["wls"], wk["rays"]


def worker(hb, out, k):
    t0 = time.perf_counter()
    for _ in range(reps):
        run_session(hb, sc, rd, wls[0], n)
    hb.sync()
    out[k] = time.perf_counter() - t0


for engines in (1, n_eng):
    hbs = [HipTraceBackend(device=0, seed=42 + k, **{"async": 1}) for k in range(engines)]
    for hb in hbs:
        if wk.get("filters"):
            hb.set_filters(wk["filters"])
        run_session(hb, sc, rd, wls[0], n)
        hb.sync()
    out = [0.0] * engines
    th = [threading.Thread(target=worker, args=(hbs[k], out, k)) for k in range(engines)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print("%s: %d engine(s) x %d sessions of %d rays: %.1f ms wall, %.2f G rays/s aggregate" % (which, engines, reps, n, dt * 1e3, engines * reps * n / dt / 1e9), flush=True)
    for hb in hbs:
        hb.close()
