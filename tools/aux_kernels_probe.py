"""Exercise every auxiliary kernel once at configs[1] size so that one rocprofv3 --kernel-trace --stats run times them all."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import run_session
sc, rd = scenes.config2_scene(), scenes.config2_render()
hb = HipTraceBackend(device=0, seed=42)
for rep in range(3):
    for w in (450.0, 550.0, 650.0):
        run_session(hb, sc, rd, scenes.wl_discrete(w), 20_000_000)          # trace + fold
    hb.ConsumeDeviceFused()                                                 # consumer fold
    hb.Snapshot(want_xyz=False)                                             # post-snapshot
full = scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, 2048, 1024, visible=abi.VISIBLE_FULL)
for rep in range(3):
    run_session(hb, sc, full, scenes.wl_discrete(550.0), 20_000_000)        # binned trace + bin accumulate
sc_s = scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8)
for rep in range(3):
    run_session(hb, sc_s, scenes.render(7, 2048, 1024, el=0, visible=2), scenes.wl_discrete(550.0), 4_000_000)   # shapegen + pool trace
for rep in range(3):                                                        # two-level binned route: split + range accumulate, 64-plane fold
    run_session(hb, sc_s, scenes.render(7, 2048, 1024, el=0, visible=2), scenes.wl_illuminant("D65", 64), 20_000_000)
g = {"type": "gauss", "mean": 1.0, "std": 0.1}
full_ax = {"type": "uniform", "mean": 0.0, "std": 360.0}
pyr = scenes.entry(scenes.pyramid_crystal(0.1, 1.2, 0.5, upper_miller=(2, 3), face_distance=[g] * 6), scenes.axis(zenith=full_ax, azimuth=full_ax, roll=full_ax), 1.0, 5)
for rep in range(3):                                                        # pyramid generator + ShapeDev pool trace
    run_session(hb, scenes.scene([(0.0, [pyr])], max_hits=8), scenes.render(7, 2048, 1024, el=0, visible=2), scenes.wl_discrete(550.0), 4_000_000)
hb.close()
