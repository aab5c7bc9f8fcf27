"""Collects the config documents of the reference's own end-to-end tests into one input fixture.

  inputs : /root/reference/test/e2e/configs/*.json   (data files the reference's tests hold: crystal / filter / scene / render
           documents, no code)
           /root/reference/examples/*.json           (the documents the reference ships as its examples: config_example.json — the README's
           quick-start run — bench_config.json, bench_config_stoch.json, lens_orthographic.json)
  output : tests/golden/ref_e2e_configs.json          {"<name>": <document>, ...}
           tests/golden/ref_example_configs.json      {"<name>": <document>, ...}   (bench.py --config ref:config_example times the first)

Run in the build container (needs /root/reference); the GPU box only reads the committed output.
tests/test_config_json.py parses every document through ice_halo_sim_amd.config;
tests/test_gpu_parity.py::test_reference_e2e_configs_parity traces each on the HIP backend and on the oracle.
"""
import glob
import json
import os

SRC = "/root/reference/test/e2e/configs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_e2e_configs.json")

if __name__ == "__main__":
    docs = {}
    for path in sorted(glob.glob(os.path.join(SRC, "*.json"))):
        with open(path) as f:
            docs[os.path.basename(path)[:-5]] = json.load(f)
    with open(OUT, "w") as f:
        json.dump(docs, f, indent=1, sort_keys=True)
    print("wrote %d documents to %s" % (len(docs), OUT))
    ex = {}
    for path in sorted(glob.glob("/root/reference/examples/*.json")):
        with open(path) as f:
            ex[os.path.basename(path)[:-5]] = json.load(f)
    out2 = os.path.join(os.path.dirname(OUT), "ref_example_configs.json")
    with open(out2, "w") as f:
        json.dump(ex, f, indent=1, sort_keys=True)
    print("wrote %d documents to %s" % (len(ex), out2))
