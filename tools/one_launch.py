import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import run_session
hb = HipTraceBackend(device=0, seed=42)
st = run_session(hb, scenes.config2_scene(), scenes.config2_render(), scenes.wl_discrete(550.0), int(os.environ.get("ONE_LAUNCH_RAYS", "10000000")))
print("kernel ms", sum(s.kernel_ms for s in st))
hb.close()
