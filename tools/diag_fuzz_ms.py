"""A multi-scatter seed of tests/test_gpu_fuzz.py: landed weight and exit counts of HIP and the oracle under several RNG seeds (is a difference inside the seed-to-seed scatter?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests._oracle_backend import OracleBackend, run_session
from tests.test_gpu_fuzz import make_ms_case
from tests.test_gpu_parity import hip_backend
for seed in [int(a) for a in sys.argv[1:]]:
    sc, rd, wl, filters, clock = make_ms_case(seed)
    print("=== seed", seed, "layers", sc.layer_count, "probs", [round(sc.layers[l].prob, 2) for l in range(sc.layer_count)], "max_hits", sc.max_hits, "lens", rd.lens_type, "vis", rd.visible, "wl", wl.illuminant, wl.pool_size, "filters", len(filters))
    for who in ("hip", "ora"):
        row = []
        for s in (seed, seed + 1000, seed + 2000, seed + 3000):
            b = hip_backend(seed=s, geom_clock=clock) if who == "hip" else OracleBackend(seed=s, threads=16, geom_clock=clock)
            b.set_filters(filters)
            st = run_session(b, sc, rd, wl, 100_000)
            img, landed = b.ReadbackXyzAccum(); b.close()
            row.append((round(landed, 1), sum(x.exit_count for x in st), [x.continuation_count for x in st]))
        print("  ", who, row)
