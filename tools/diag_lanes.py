"""Raypath-colour class lanes of tests/test_gpu_fuzz.py colour seeds against the oracle's double lanes (seed 5103: the hot-pixel case that read 3.1e-3 low with fp32 lane atomics)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from tests import test_gpu_fuzz as F
r = F.run_color_case(5103)
print("colour seed 5103: lanes hip", r["lanes"][0], "oracle", r["lanes"][1], "rel", np.abs(r["lanes"][0]-r["lanes"][1])/np.maximum(r["lanes"][1],1), "lane_l2", r["lane_l2"])
for s in (5000, 5003, 5007):
    r = F.run_color_case(s)
    print("colour seed", s, "rel", np.abs(r["lanes"][0]-r["lanes"][1])/np.maximum(r["lanes"][1],1), "lane_l2", np.round(r["lane_l2"],6))
