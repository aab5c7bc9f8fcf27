"""One seed of tools/route_fuzz.py against the ORACLE (fp64 accumulator): which of the two routes is off when they disagree?
   python tools/diag_route_seed.py <seed>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ice_halo_sim_amd import abi, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import OracleBackend, run_session

seed = int(sys.argv[1])
SIZES = [(64, 48), (333, 211), (512, 256), (1024, 512), (1920, 1080), (2048, 1024), (2048, 2048), (2896, 2896), (4096, 2048), (8192, 1024)]
RAYS = [(2 << 20) - 1, 2 << 20, (2 << 20) + 77, 3 << 20, (8 << 20) - 1, 8 << 20, 9 << 20]
rng = np.random.default_rng(seed)   # (the draws of tools/route_fuzz.py, in its order)
w, h = SIZES[rng.integers(len(SIZES))]
n = int(RAYS[rng.integers(len(RAYS))])
wl = scenes.wl_discrete(float(rng.uniform(400, 700))) if rng.random() < 0.35 else scenes.wl_illuminant(str(rng.choice(["D65", "A"])), int(rng.choice([1, 2, 3, 7, 31, 64, 255])))
lens = int(rng.integers(0, 11))
rd = scenes.render(lens, w, h, fov=float(rng.uniform(30, 110)) if lens == abi.LENS_LINEAR else 180.0, az=float(rng.uniform(0, 360)), el=float(rng.uniform(0, 90)), visible=int(rng.integers(0, 3)))
kind = int(rng.integers(3))
u = lambda m, s: {"type": "uniform", "mean": m, "std": s}
full = u(0.0, 360.0)
if kind == 0:
    e = scenes.column_crystal_entry()
elif kind == 1:
    e = scenes.entry(scenes.prism_crystal(u(1.0, 0.6), [u(1.0, 0.3)] * 6), scenes.axis(zenith=full, azimuth=full, roll=full), 1.0, 1)
else:
    e = scenes.entry(scenes.pyramid_crystal(u(0.3, 0.3), u(1.0, 0.5), 0.2, face_distance=[u(1.0, 0.2)] * 6), scenes.axis(zenith=u(90.0, 20.0), azimuth=full, roll=full), 1.0, 1)
entries = [e] if rng.random() < 0.6 else [e, scenes.entry(scenes.prism_crystal(0.3), scenes.axis(zenith={"type": "gauss", "mean": 0, "std": 1.0}, roll=full), 0.7, 2)]
sc = scenes.scene([(0.0, entries)], max_hits=int(rng.choice([3, 7, 8])))
print("seed %d: %dx%d n=%d wl=(%d,%d) lens=%d vis=%d kind=%d entries=%d" % (seed, w, h, n, wl.illuminant, wl.pool_size, lens, rd.visible, kind, len(entries)))
imgs = {}
for name, opts in (("default route", {}), ("direct atomics", {"hit_log": 0, "bin": 0}), ("direct atomics, 1 plane copy", {"hit_log": 0, "bin": 0, "mono_copies": 1})):
    hb = HipTraceBackend(device=0, seed=seed, **opts)
    run_session(hb, sc, rd, wl, n)
    imgs[name] = hb.ReadbackXyzAccum()[0].astype(np.float64)
    hb.close()
ob = OracleBackend(seed=seed, threads=max(8, min(os.cpu_count() or 8, 128)), acc64=1)
run_session(ob, sc, rd, wl, n)
ref = ob.ReadbackXyzAccum()[0].astype(np.float64)
ob.close()
den = np.linalg.norm(ref)
for name, img in imgs.items():
    d = img - ref
    k = np.unravel_index(np.argmax(np.abs(d)), d.shape)
    print("%-30s vs oracle (fp64 sums): rel L2 %.2e; largest difference %.4g at pixel %s where the oracle has %.6g" % (name, np.linalg.norm(d) / den, d[k], k[:2], ref[k]))
a, b = imgs["default route"], imgs["direct atomics"]
d = a - b
k = np.unravel_index(np.argmax(np.abs(d)), d.shape)
print("default vs direct: rel L2 %.2e; largest difference %.4g at pixel %s: default %.6f direct %.6f oracle %.6f; pixels differing by > 1e-3 of the brightest: %d; brightest pixel %.6g (default) %.6g (direct) %.6g (oracle)" %
      (np.linalg.norm(d) / np.linalg.norm(b), d[k], k[:2], a[k], b[k], ref[k], int((np.abs(d) > 1e-3 * b.max()).sum()), a.max(), b.max(), ref.max()))
order = np.argsort(-np.abs(d).ravel())[:6]
for o in order:
    kk = np.unravel_index(o, d.shape)
    print("   pixel %s ch %d: default %.6f direct %.6f oracle %.6f" % (kk[:2], kk[2], a[kk], b[kk], ref[kk]))
