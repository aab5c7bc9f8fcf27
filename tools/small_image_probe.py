"""Small and large images through the hit log (tile count must not depend on the image size): discrete and D65 sessions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ice_halo_sim_amd import abi, scenes
from tools.perf_probe import run
sc = scenes.config2_scene()
for (w, h) in ((256, 128), (512, 256), (1024, 512), (1920, 1080)):
    for vis in (abi.VISIBLE_UPPER, abi.VISIBLE_FULL):
        rd = scenes.render(1, w, h, fov=180, el=30, visible=vis)
        a = run("fisheye %dx%d visible %d n=10M default" % (w, h, vis), sc, rd)
        b = run("   hit_log=0", sc, rd, hit_log=0)
