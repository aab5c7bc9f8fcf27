#!/usr/bin/env python3
"""Renders the oracle halves of the slowest `-m gpu` comparisons once and commits their summaries under tests/golden/oracle_renders/
(tests/_oracle_cache.py).  CPU only — liboracle.so; run it in the build container after a change to the oracle or to one of the scenes:

    python tests/golden/make_oracle_render_fixtures.py            # render everything, write fixtures + MANIFEST.json
    python tests/golden/make_oracle_render_fixtures.py --stale    # render only the fixtures whose fingerprint moved (tests/_oracle_cache.py)
    python tests/golden/make_oracle_render_fixtures.py --stamp    # record fingerprints of the files as they are — ONLY after proving from
                                                                  # git history that no fingerprinted input changed since they were rendered
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
MODE = sys.argv[1] if len(sys.argv) > 1 else ""
if MODE == "--stamp":
    # the proof: the last commit of every input file is not younger than the commit that wrote the fixtures, and the work tree has no edit of them
    from tests import _oracle_cache as _C
    ins = list(_C.BASE_INPUTS) + ["ice_halo_sim_amd/config.py", "tests/golden/ref_e2e_configs.json", "tests/test_gpu_parity.py", "tests/test_gpu_filter_production.py",
                                  "tests/test_gpu_production_routes.py"]

    def ct(path):
        return int(subprocess.check_output(["git", "log", "-1", "--format=%ct", "--", path], cwd=ROOT).decode().strip() or 0)
    t_fix = ct("tests/golden/oracle_renders")
    young = [f for f in ins[: len(_C.BASE_INPUTS) + 2] if ct(f) > t_fix]
    dirty = subprocess.check_output(["git", "status", "--porcelain", "--"] + ins[: len(_C.BASE_INPUTS) + 2], cwd=ROOT).decode().strip()
    if young or dirty:
        sys.exit("refusing to stamp: inputs changed since the fixtures were rendered: %s %s" % (young, dirty))
    # (the compute functions live in the three test modules: their source is part of each fingerprint, and a stamp is only honest while the text
    #  of those functions is what rendered the files — compared against the fixtures' commit below)
    for mod in ins[-3:]:
        old = subprocess.check_output(["git", "show", "%s:%s" % (subprocess.check_output(["git", "log", "-1", "--format=%H", "--", "tests/golden/oracle_renders"], cwd=ROOT).decode().strip(), mod)], cwd=ROOT).decode()
        new = open(os.path.join(ROOT, mod)).read()
        import re

        def computes(text):   # the bodies of the nested compute() functions, up to the `from tests._oracle_cache import cached` that follows each
            return re.findall(r"    def compute\(\):\n(.*?)\n    from tests\._oracle_cache import cached", text, re.S)
        if computes(old) != computes(new):
            sys.exit("refusing to stamp: a compute() in %s differs from the one that rendered the fixtures" % mod)
    os.environ["HALO_STAMP_ORACLE_FIXTURES"] = "1"
elif MODE == "--stale":
    os.environ["HALO_WRITE_ORACLE_FIXTURES"] = ""
else:
    os.environ["HALO_WRITE_ORACLE_FIXTURES"] = "1"
if MODE == "--stale":   # cached() renders live what is stale; to have it WRITTEN, the write flag is raised per stale key
    from tests import _oracle_cache as _C
    _orig = _C.cached

    def _cached(key, compute, inputs=()):
        fp = _C.fingerprint(compute, inputs)
        stale = not os.path.exists(os.path.join(_C.DIR, key + ".npz")) or _C.load_manifest().get(key) != fp
        os.environ["HALO_WRITE_ORACLE_FIXTURES"] = "1" if stale else ""
        if stale:
            print("re-rendering", key)
        try:
            return _orig(key, compute, inputs)
        finally:
            os.environ["HALO_WRITE_ORACLE_FIXTURES"] = ""
    _C.cached = _cached

from tests import test_gpu_production_routes as R   # noqa: E402

for seed in (42, 7):
    s = R.oracle_config3(seed)
    print("config3 seed %d: landed %.1f, continuations %s" % (seed, float(s["landed"]), list(s["cont"])))
from tests import test_gpu_parity as P   # noqa: E402

for name in P.e2e_multilayer_names():
    for seed in (42, 7):
        s = P.oracle_e2e_multilayer(name, seed)
        print("e2e %s seed %d: continuations %s exits %d landed %.1f" % (name, seed, list(s["cont"]), int(s["n_exits"]), float(s["landed"])))
from tests import test_gpu_filter_production as F   # noqa: E402

for name in F.multilayer_docs():
    for seed in (42, 7):
        s = F.oracle_doc_multilayer(name, seed)
        print("filter doc %s seed %d: continuations %s landed %.1f" % (name, seed, list(s["cont"]), float(s["landed"])))
print(sorted(os.listdir(os.path.join(ROOT, "tests", "golden", "oracle_renders"))))
