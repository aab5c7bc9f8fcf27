"""The fixed-orientation scenes of tests/test_gpu_fuzz.py whose bars the Fresnel split's hardware reciprocal / square root had forced down
(seeds 247, 411, 702, 11584 single-layer; 3201 two-layer), per library build (HALO_LIB): matched-exit fraction at the per-ray bars
(direction 2e-5, weight 2e-4), and the production-kernel image / landed-weight distances.  gpurun -- 'for t in f0 f1 -; do ...; done'"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fuzz as F

single = [int(x) for x in os.environ.get("SEEDS", "247 411 702 11584").split()]
ms = [int(x) for x in os.environ.get("MS_SEEDS", "3201").split()]
print("lib", os.environ.get("HALO_LIB", "(product)"))
for s in single:
    r = F.run_case(s)
    print("  seed %5d fixed %d match %.5f (the oracle's two roundings between themselves: %.5f) | within the bars + 4 x the oracles' own difference per exit: %.5f (%d exits only one oracle emits) | landed %.6g / %.6g l2 %.2e" % (
        s, r["fixed_axes"], r["match"][0], r["oracle_pair"], r["cond"][0], r["cond"][3], r["landed"][0], r["landed"][1], r["l2"]))
    if os.environ.get("PROD", "1") != "0":
        p = F.run_production_case(s)
        print("      production: exits %d / %d  landed rel %.2e  l2 %.2e  sums rel %s" % (p["exits"][0], p["exits"][1], abs(p["landed"][0] - p["landed"][1]) / max(p["landed"][1], 1.0), p["l2"],
              ["%.1e" % (abs(a - b) / max(b, 1e-30)) for a, b in zip(p["sums"][0], p["sums"][1])]))
for s in ms:
    r = F.run_ms_case(s)
    print("  ms seed %5d fixed %d first-layer match %.5f, conditioned %.5f cont %s" % (s, r["fixed_axes"], r["first"][0], r["first_cond"][0], r["cont"][0]))
