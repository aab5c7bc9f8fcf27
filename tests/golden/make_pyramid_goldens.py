#!/usr/bin/env python3
"""Extract the reference's fixed pyramid sample pools and their topology goldens (DATA held by the reference's own
tests) into tests/golden/ref_pyramid_goldens.npz.  Build container only.
  inputs : test/golden-analytic/core/closed_form_samples_generated.hpp  kPyramidWellConditionedSamples, kPyramidMillerSamples, the six
           flat-tail pools kPyramidFlatTailAlpha{85,87,875,88,89,895}Samples (wedge angles 85 .. 89.5 degrees: caps a few hundredths high)
           and — inputs only, the reference snapshots no topology for them — the two degenerate pools kPyramidDegenerateSigma0{30,50}Samples
  goldens: test/golden-analytic/core/pyramid_topology_golden_generated.hpp  (vtx_cnt, face_present_mask, path_tag_union)"""
import os
import re

import numpy as np

REF = "/root/reference/test/golden-analytic/core/"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def num(tok):
    tok = tok.strip()
    low = tok.lower()
    if low.startswith(("0x", "-0x")):
        if "p" in low:
            return float.fromhex(tok)
        return int(low.rstrip("u"), 16)
    tok = tok.rstrip("fFuU")
    return float(tok) if ("." in tok or "e" in tok.lower()) else int(tok)


def array_body(txt, name):
    m = re.search(re.escape(name) + r"\s*\[[^\]]*\]\s*=\s*\{(.*?)\n\};", txt, re.S)
    assert m, name
    return re.sub(r"//[^\n]*", "", m.group(1))


def flat_numbers(body):
    return [num(t) for t in re.findall(r"-?0x[0-9a-fA-F.]+(?:p[-+]?\d+)?u?|-?\d+\.?\d*(?:[eE][-+]?\d+)?[fu]?", body)]


def main():
    samples = open(REF + "closed_form_samples_generated.hpp").read()
    topo = open(REF + "pyramid_topology_golden_generated.hpp").read()
    wc = np.asarray(flat_numbers(array_body(samples, "kPyramidWellConditionedSamples")), np.float64).reshape(-1, 11)
    ml = np.asarray(flat_numbers(array_body(samples, "kPyramidMillerSamples")), np.float64).reshape(-1, 13)
    wc_t = np.asarray(flat_numbers(array_body(topo, "kPyramidWellConditionedTopology")), np.int64).reshape(-1, 3)
    ml_t = np.asarray(flat_numbers(array_body(topo, "kPyramidMillerTopology")), np.int64).reshape(-1, 3)
    assert len(wc) == len(wc_t) == 200 and len(ml) == len(ml_t) == 288, (wc.shape, wc_t.shape, ml.shape, ml_t.shape)
    extra = {}
    for tag in ("85", "87", "875", "88", "89", "895"):   # pyramid_topology_golden_generated.hpp:184-264
        smp = np.asarray(flat_numbers(array_body(samples, "kPyramidFlatTailAlpha%sSamples" % tag)), np.float64).reshape(-1, 11)
        top = np.asarray(flat_numbers(array_body(topo, "kPyramidFlatTailAlpha%sTopology" % tag)), np.int64).reshape(-1, 3)
        assert len(smp) == len(top) == 40, (tag, smp.shape, top.shape)
        extra["flat%s_samples" % tag] = smp.astype(np.float32)
        extra["flat%s_topology" % tag] = top
    for tag in ("030", "050"):   # closed_form_samples_generated.hpp:5353,5601 — no golden: test_closed_form_pyramid.cpp DegenerateContractSafe only
        smp = np.asarray(flat_numbers(array_body(samples, "kPyramidDegenerateSigma%sSamples" % tag)), np.float64).reshape(-1, 11)
        extra["degenerate%s_samples" % tag] = smp.astype(np.float32)
    out = os.path.join(ROOT, "tests", "golden", "ref_pyramid_goldens.npz")
    np.savez_compressed(out, wc_samples=wc.astype(np.float32), wc_topology=wc_t, miller_samples=ml.astype(np.float32), miller_topology=ml_t, **extra)
    print("wrote", out, wc.shape, ml.shape, {k: v.shape for k, v in extra.items()})


if __name__ == "__main__":
    main()
