import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import run_session
case, b = sys.argv[1], int(sys.argv[2])
sc = scenes.config2_scene()
if case == "full":
    rd, n = scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, 2048, 1024, visible=abi.VISIBLE_FULL), 20_000_000
elif case == "config2":
    rd, n = scenes.config2_render(), 50_000_000
else:
    sc = scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8)
    rd, n = scenes.render(7, 2048, 1024, el=0, visible=2), 16_000_000
hb = HipTraceBackend(device=0, seed=42, bin=b)
for r in range(3):
    st = run_session(hb, sc, rd, scenes.wl_discrete(550.0), n)
print(case, b, sum(s.kernel_ms for s in st), sum(s.pixel_hits for s in st) / n)
