"""What a Lumice server that keeps its CUDA-route dispatch size (server.cpp:151: 262144 rays per SimBatch = one BeginSession / TraceLayer /
EndSession each) would get from this backend, against the large dispatches the glue asks for: wall time per session and rays/s for session
sizes 2^16 ... 2^24 on configs[1]'s scene, sync and async.  usage: python tools/dispatch_size_probe.py [reps]   (under rocprofv3 --kernel-trace
--stats for the per-kernel split)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ice_halo_sim_amd import scenes  # noqa: E402
from ice_halo_sim_amd.backend import HipTraceBackend  # noqa: E402
from tests._oracle_backend import run_session  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
only = int(sys.argv[2]) if len(sys.argv) > 2 else 0
opts = dict(kv.split("=") for kv in sys.argv[3:])   # backend options, e.g. blocks_per_cu=2
sc, rd = scenes.config2_scene(), scenes.config2_render()
wl = scenes.wl_discrete(550.0)
for mode in (0, 1):
    hb = HipTraceBackend(device=0, seed=42, **dict({"async": mode}, **{k: int(v) for k, v in opts.items()}))
    for log2 in ([only] if only else ((16, 18, 20, 22, 24) if not os.environ.get('PROBE_SIZES') else [int(v) for v in os.environ['PROBE_SIZES'].split(',')])):
        n = 1 << log2
        k = max(3, reps >> max(0, log2 - 18))
        for rep in range(2):
            hb.sync()
            t0 = time.perf_counter()
            for _ in range(k):
                run_session(hb, sc, rd, wl, n)
            hb.sync()
            dt = time.perf_counter() - t0
        st = hb.collect_stats()
        print("async=%d  2^%d rays per session: %.3f ms per session wall (%.3f ms of kernels), %.2f G rays/s" %
              (mode, log2, dt * 1e3 / k, st.kernel_ms / max(st.launches, 1), k * n / dt / 1e9), flush=True)
    hb.close()
