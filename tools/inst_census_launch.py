"""One configs[1]-shaped launch with knobs, for tools/inst_census.sh: CENSUS_MAX_HITS (1..7) and CENSUS_VIEW (normal | away — a camera that
looks at the nadir under `visible: upper`: every exit is culled before the exit queue, so nothing is projected or accumulated)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ice_halo_sim_amd import abi, scenes   # noqa: E402
from ice_halo_sim_amd.backend import HipTraceBackend   # noqa: E402
from tests._oracle_backend import run_session   # noqa: E402
H = int(os.environ.get("CENSUS_MAX_HITS", "7"))
view = os.environ.get("CENSUS_VIEW", "normal")
n = int(os.environ.get("CENSUS_RAYS", "20000000"))
sc = scenes.scene([(0.0, [scenes.column_crystal_entry()])], max_hits=H)
rd = scenes.config2_render() if view == "normal" else scenes.render(abi.LENS_FISHEYE_EQUAL_AREA, 1920, 1080, fov=180.0, el=30.0, visible=abi.VISIBLE_LOWER)
hb = HipTraceBackend(device=0, seed=42)
run_session(hb, sc, rd, scenes.wl_discrete(550.0), n)
st = run_session(hb, sc, rd, scenes.wl_discrete(550.0), n)
print("max_hits", H, "view", view, "rays", n, "kernel ms", sum(s.kernel_ms for s in st), "exits", st[0].exit_count, "pixel hits", st[0].pixel_hits)
hb.close()
