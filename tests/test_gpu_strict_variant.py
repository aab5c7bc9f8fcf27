"""The build with the reference's roundings, kept alive as a TESTED variant (round-4 review): `libhalo_hip_strict.so` = HALO_STRICT (every fma
chain of the ray's way spelled as separately rounded products and sums, in the same order) + HALO_FRESNEL=1 (IEEE division and square root in the
Fresnel split) + -ffp-contract=off, built by __graft_entry__.build() / `HALO_BUILD_TAG=strict python -m ice_halo_sim_amd.build`.

The product contracts products into FMAs, and on fixed-orientation scenes — every ray meets the crystal the same way — a few percent of its exits
then differ from the uncontracted oracle by more than the per-ray bars (direction 2e-5, weight 2e-4); tests/test_gpu_parity.py's
match_exits_conditioned therefore widens each exit's bar by what the oracle's own two roundings differ by on that exit.  That yardstick is only
as good as the claim behind it: "a build that rounds like the reference agrees with the oracle exit by exit".  This file keeps the claim
falsifiable — the strict build is held to the UNCONDITIONED bars on the seeds that motivated the conditioning."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRICT = os.path.join(ROOT, "ice_halo_sim_amd", "libhalo_hip_strict.so")
# unconditioned match_exits of the strict build, measured (profiles/r05_strict_variant.txt): 0.99887 / 0.99962 / 0.99444 / 0.99957 — against
# 0.9948 / 0.9968 / 0.9811 / 0.9096 of the product and 0.945 / 0.951 / 0.976 / 0.918 between the oracle's own two roundings
BARS = {247: 0.998, 411: 0.998, 702: 0.993, 11584: 0.998}
# Round 6: the strict build over ALL seeds (49 then; 52 with the sweep's three at the end of the round) of test_random_scene_traces_the_same_rays_as_the_oracle against the unconditioned bars
# (tools/strict_sweep.py, profiles/r06_strict_sweep.txt): every seed >= 0.99915 (the product build >= 0.9985 on the same list — the conditioning is
# only ever needed on fixed-orientation scenes like the four above).  The worst eight of that sweep stay pinned here at 0.998.
SWEEP_WORST = {137: 0.998, 20234: 0.998, 104: 0.998, 121: 0.998, 100: 0.998, 130: 0.998, 129: 0.998, 122: 0.998}
# Round 6, sweep of seeds 40000..40399: seed 40254 (sun 0.28 degrees under a FIXED pyramid's basal plane, sun diameter 0 — 20 k identical rays, so the
# oracle pair is one sample of the rounding): oracle pair 0.98403, product 0.98070 (0.98300 conditioned — the product differs on other exits
# than the pair does, so the widening misses them), strict build 0.99991 unconditioned.  The claim stays falsifiable on it: the strict build
# must meet the plain bar, and the product must be no further from the oracle than the oracle's two roundings are from each other.
SINGLE_SAMPLE = {40254: 0.998}

DRIVER = r"""
import json, sys
sys.path.insert(0, %r)
from tests import test_gpu_fuzz as F
out = {}
for s in %r:
    r = F.run_case(s)
    out[s] = {"match": float(r["match"][0]), "cond": float(r["cond"][0]), "pair": float(r["oracle_pair"]), "fixed": int(r["fixed_axes"]),
              "landed": [float(r["landed"][0]), float(r["landed"][1])]}
print("RESULT " + json.dumps(out))
"""


def _start(lib):
    env = dict(os.environ)
    if lib:
        env["HALO_LIB"] = lib
    else:
        env.pop("HALO_LIB", None)
    return subprocess.Popen([sys.executable, "-c", DRIVER % (ROOT, sorted(BARS) + sorted(SWEEP_WORST) + sorted(SINGLE_SAMPLE))], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _finish(proc):
    out, err = proc.communicate(timeout=900)
    assert proc.returncode == 0, err[-2000:]
    line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
    return {int(k): v for k, v in json.loads(line[7:]).items()}


def _run_both(lib_a, lib_b):
    """the two libraries' runs side by side (each is mostly the CPU oracle's time: two roundings x four scenes)"""
    a, b = _start(lib_a), _start(lib_b)
    try:
        return _finish(a), _finish(b)
    finally:
        for p in (a, b):
            if p.poll() is None:
                p.kill()


def test_strict_build_meets_the_unconditioned_per_ray_bars():
    assert os.path.exists(STRICT), "libhalo_hip_strict.so is not built: HALO_BUILD_TAG=strict python -m ice_halo_sim_amd.build (__graft_entry__.build() does it)"
    strict, product = _run_both(STRICT, None)
    for seed, bar in BARS.items():
        s, p = strict[seed], product[seed]
        assert s["fixed"] >= 1                                    # the scenes that motivated the conditioning: fixed orientation axes
        assert s["match"] >= bar, (seed, s)
        assert s["landed"][0] == pytest.approx(s["landed"][1], rel=1e-4)
        # ... and the comparison the conditioning rests on: the reference-rounding build is at least as close to the oracle as the product, which
        # in turn sits inside the oracle's own two roundings' spread once conditioned
        assert s["match"] >= p["match"] - 1e-3 and p["cond"] >= 0.995, (seed, s, p)
    assert sum(strict[k]["match"] - product[k]["match"] for k in BARS) > 0.05      # (the product really does differ there: seed 11584 by 0.09)
    for seed, bar in SWEEP_WORST.items():                                          # the whole seed list's worst eight, no conditioning
        assert strict[seed]["match"] >= bar, (seed, strict[seed])
        assert product[seed]["match"] >= 0.995, (seed, product[seed])               # (random orientations: the product meets the plain bar too)
    for seed, bar in SINGLE_SAMPLE.items():
        s, p = strict[seed], product[seed]
        assert s["fixed"] >= 1 and s["match"] >= bar, (seed, s)
        assert p["cond"] >= p["pair"] - 0.01 and p["match"] >= p["pair"] - 0.01, (seed, p)
