import sys, os
sys.path.insert(0, os.getcwd())
from ice_halo_sim_amd import abi, scenes
from tools.perf_probe import run
sc = scenes.config2_scene()
for name, rd in (("dual fisheye EA 2048x1024 full", scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, 2048, 1024, visible=abi.VISIBLE_FULL)),
                 ("fisheye EA 1920x1080 full", scenes.render(1, 1920, 1080, fov=180, el=30, visible=abi.VISIBLE_FULL))):
    for n in (10_000_000, 50_000_000):
        run("%s n=%dM default(bin)" % (name, n // 1000000), sc, rd, n=n)
        run("%s n=%dM bin=0 (log)" % (name, n // 1000000), sc, rd, n=n, bin=0)
sp = scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8)
rd = scenes.render(abi.LENS_RECTANGULAR, 2048, 1024, el=0.0, visible=abi.VISIBLE_FULL)
for n in (10_000_000, 25_000_000):
    run("stoch prism 550nm rect full n=%dM default(bin)" % (n // 1000000), sp, rd, n=n)
    run("stoch prism 550nm rect full n=%dM bin=0 (log)" % (n // 1000000), sp, rd, n=n, bin=0)
