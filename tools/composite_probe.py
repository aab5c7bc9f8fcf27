"""Times halo_consumer_composite (three histogram passes of the P99 radix select + the per-pixel composite) on device lanes of a production
size.  usage: python tools/composite_probe.py [width height classes repeats]   (run under rocprofv3 --kernel-trace --stats for the kernels)"""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ice_halo_sim_amd import backend, scenes  # noqa: E402

w, h, n, rep = (int(v) for v in (sys.argv[1:5] + ["2048", "1024", "16", "10"][len(sys.argv) - 1:]))
rng = np.random.default_rng(1)
lanes = np.zeros((n, h, w), np.float32)
m = rng.random((n, h, w)) < 0.4
lanes[m] = np.exp(rng.normal(-3.0, 2.0, size=int(m.sum()))).astype(np.float32)
b = backend.HipTraceBackend(device=0, seed=1)
b.set_color([], [scenes.color_class([i]) for i in range(n)])
b.LoadClassLanes(lanes, float(lanes.sum()))
classes = [{"color": [float(v) for v in rng.random(3)]} for _ in range(n)]
for mode in ("dominant", "additive", "painter"):
    b.CompositeColorClasses(classes, mode)   # warm-up
    t0 = time.perf_counter()
    for _ in range(rep):
        ok, lin, srgb, p99 = b.CompositeColorClasses(classes, mode)
    dt = (time.perf_counter() - t0) / rep
    print("%-9s %dx%d x %d classes: %.3f ms per composite call (select + composite + %.0f MB of readback), P99 %.5g" %
          (mode, w, h, n, dt * 1e3, (lin.nbytes + srgb.nbytes) / 1e6, p99))
b.close()
