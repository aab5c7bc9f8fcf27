#!/usr/bin/env python3
"""Extract the reference's fixed pyramid sample pools and their topology goldens (DATA held by the reference's own
tests) into tests/golden/ref_pyramid_goldens.npz.  Build container only.
  inputs : test/golden-analytic/core/closed_form_samples_generated.hpp  kPyramidWellConditionedSamples, kPyramidMillerSamples
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
    out = os.path.join(ROOT, "tests", "golden", "ref_pyramid_goldens.npz")
    np.savez_compressed(out, wc_samples=wc.astype(np.float32), wc_topology=wc_t, miller_samples=ml.astype(np.float32), miller_topology=ml_t)
    print("wrote", out, wc.shape, ml.shape)


if __name__ == "__main__":
    main()
