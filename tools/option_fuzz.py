"""Backend options, HIP against HIP: the random scenes of tests/test_gpu_fuzz.py (filters included) at 3 Mi rays with the default options
against a random setting of the options that must not change what is rendered — launch chunking, dispatch-ahead, workgroups per CU, plane
copies, coarse-list count, the fast entry pick / regular-prism search / fast filter tables, per-entry planes.  Exit count, landed weight and the
image must agree (different kernels may put one hit in ~1e5 into a neighbouring pixel).   python tools/option_fuzz.py [first] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests._oracle_backend import run_session
from tests.test_gpu_fuzz import make_case
from tests.test_gpu_parity import block_mean, hip_backend, rel_l2

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 60)
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed + 99991)
    sc, rd, wl, filters, clock = make_case(seed)
    opts = {}
    if rng.random() < 0.5: opts["chunk"] = int(rng.choice([1 << 20, 3 << 19, 1 << 21]))   # (multiples of every shape clock: a chunk boundary inside a clock group would re-deal the crystals)
    if rng.random() < 0.4: opts["stoch_chunk"] = int(rng.choice([1 << 19, 1 << 21]))
    if rng.random() < 0.3: opts["async"] = 1
    if rng.random() < 0.4: opts["blocks_per_cu"] = int(rng.choice([1, 5, 64]))
    if rng.random() < 0.3: opts["mono_copies"] = int(rng.choice([1, 2, 16]))
    if rng.random() < 0.3: opts["bin_l1"] = int(rng.choice([8, 64, 512]))
    if rng.random() < 0.3: opts["entry_fast"] = 0
    if rng.random() < 0.3: opts["hex_fast"] = 0
    if rng.random() < 0.3: opts["filter_fast"] = 0
    if rng.random() < 0.3: opts["lambda_planes"] = int(rng.choice([0, 1]))
    print("seed %d opts %s" % (seed, opts), end=" ", flush=True)
    base_clock = opts.get("geom_clock", clock)
    res = []
    for o in ({"geom_clock": base_clock}, dict(opts, geom_clock=base_clock)):
        hb = hip_backend(seed=seed, **o)
        hb.set_filters(filters)
        st = run_session(hb, sc, rd, wl, 3 << 20)
        if o.get("async"):
            tot = hb.collect_stats()
            exits = tot.exit_count
        else:
            exits = st[0].exit_count
        img, landed = hb.ReadbackXyzAccum()
        hb.close()
        res.append((exits, landed, img))
    (xa, la, ia), (xb, lb, ib) = res
    err = rel_l2(block_mean(ia, 4), block_mean(ib, 4)) if ib.any() else 0.0
    ok = abs(xa - xb) <= 3 and abs(la - lb) <= 1e-5 * max(lb, 1.0) + 1e-4 and err <= 1e-3
    bad += not ok
    print("exits %d/%d landed rel %.1e block-mean rel L2 %.1e %s" % (xa, xb, abs(la - lb) / max(lb, 1.0), err, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
