import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import run_session
sc, rd = scenes.config3_scene(), scenes.config2_render()
import json
hb = HipTraceBackend(device=0, seed=42, **json.loads(os.environ.get("OPTS", "{}")))
for n in (10_000_000, 50_000_000):
    for rep in range(2):
        hb.sync(); t0 = time.perf_counter()
        st = run_session(hb, sc, rd, scenes.wl_discrete(550.0), n)
        hb.sync(); dt = time.perf_counter() - t0
    img, landed = hb.ReadbackXyzAccum()
    print("configs[2] shape, %d M roots: wall %.1f ms, kernels %.1f ms, continuations %d, exits layer1 %d, landed/root %.4f, per-layer kernel ms %s" % (n // 1_000_000, dt * 1e3, sum(s.kernel_ms for s in st), st[0].continuation_count, st[1].exit_count, landed / (2 * n), [round(x.kernel_ms, 2) for x in st]), flush=True)
