"""Diagnostic: one filtered scene at production size on the fast, generic and capture kernels and on the oracle (one session, and drained
in 1 Mi-ray pieces into float64); pairwise 8x8 block-mean rel L2.  usage: python tools/diag_filter_prod.py [case] [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import OracleBackend, run_session
from tests.test_gpu_parity import _filter_table, _with, block_mean, rel_l2

case = sys.argv[1] if len(sys.argv) > 1 else "direction_out"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6 << 20
col = scenes.column_crystal_entry()
fid = {"direction_out": 3, "complex": 4, "raypath_P": 1, "none": 0}[case]
sc = scenes.scene([(0.0, [_with(col, fid)])], max_hits=7)
rd = scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, 512, 256, visible=abi.VISIBLE_FULL)
wl = scenes.wl_discrete(550.0)
imgs = {}
def hip(tag, **opts):
    hb = HipTraceBackend(device=0, seed=21, **opts)
    hb.set_filters(_filter_table())
    st = run_session(hb, sc, rd, wl, n)
    r = hb.last_route()
    img, landed = hb.ReadbackXyzAccum()
    hb.close()
    imgs[tag] = img.astype(np.float64)
    print("%-16s mode %2d geom %2d accum %2d exits %d landed %.3f sumY %.3f" % (tag, r.mode_mask, r.geom_mask, r.accum_mask, st[0].exit_count, landed, img[..., 1].sum(dtype=np.float64)), flush=True)
hip("fast")
hip("fast_nolog", hit_log=0)
hip("generic", filter_fast=0)
hip("capture", capture_exits=1)
ob = OracleBackend(seed=21, threads=max(8, min(os.cpu_count() or 8, 128)), acc64=int(os.environ.get('ACC64', '1')))
ob.set_filters(_filter_table())
st = run_session(ob, sc, rd, wl, n)
img, landed = ob.ReadbackXyzAccum()
imgs["oracle"] = img.astype(np.float64)
print("oracle           exits %d landed %.3f sumY %.3f" % (st[0].exit_count, landed, img[..., 1].sum(dtype=np.float64)), flush=True)
acc = np.zeros_like(imgs["oracle"])
left = n
while left > 0:   # the same rays (the counters run on), drained every 1 Mi rays
    m = min(left, 1 << 20)
    run_session(ob, sc, rd, wl, m)
    im, _ = ob.ReadbackXyzAccum()
    acc += im
    left -= m
ob.close()
imgs["oracle_pieces(next rays)"] = acc
keys = list(imgs)
for i, a in enumerate(keys):
    for b in keys[i + 1:]:
        print("%-26s vs %-26s block rel L2 %.3e  pixel rel L2 %.3e" % (a, b, rel_l2(block_mean(imgs[a]), block_mean(imgs[b])), rel_l2(imgs[a], imgs[b])))
