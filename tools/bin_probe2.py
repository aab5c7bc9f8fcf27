import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import run_session
sc_s = scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8)
rd_s = scenes.render(7, 2048, 1024, el=0, visible=2)
hb = HipTraceBackend(device=0, seed=42, bin=int(sys.argv[1]))
for r in range(3):
    st = run_session(hb, sc_s, rd_s, scenes.wl_illuminant("D65", 64), 16_000_000)
print(sum(s.kernel_ms for s in st))
