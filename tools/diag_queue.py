import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import run_session
from tests.test_gpu_parity import rel_l2
sc = scenes.config2_scene()
for lens, visible in ((9, 0), (2, 1), (8, 1), (2, 2), (1, 0)):
    overlap = 0.0872 if lens in (4, 5, 6) else 0.0
    rd = scenes.render(lens, 512, 256, fov=120.0 if lens not in (4, 5, 6, 7, 9) else 180.0, el=30.0 if lens != 7 else 0.0, visible=visible, overlap=overlap)
    wl, n = scenes.wl_illuminant("D65", 31), 300_000
    out = {}
    for name, kw in (("queue", {}), ("emit", {"capture_exits": 1}), ("queue_nohex", {"hex_fast": 0})):
        hb = HipTraceBackend(device=0, seed=19, **kw)
        st = run_session(hb, sc, rd, wl, n)
        if "capture_exits" in kw: hb.DrainExits()
        img, landed = hb.ReadbackXyzAccum(); hb.close()
        out[name] = (img, landed, st[0].pixel_hits)
    print(os.environ.get("HALO_LIB", "cur")[-12:], lens, visible, "hits", [out[k][2] for k in out], "landed", ["%.2f" % out[k][1] for k in out],
          "l2 q-e %.2e  nohex-e %.2e" % (rel_l2(out["queue"][0], out["emit"][0]), rel_l2(out["queue_nohex"][0], out["emit"][0])))
