#!/usr/bin/env python3
"""Renders the oracle halves of the slowest `-m gpu` comparisons once and commits their summaries under tests/golden/oracle_renders/
(tests/_oracle_cache.py).  CPU only — liboracle.so; run it in the build container after a change to the oracle or to one of the scenes:

    python tests/golden/make_oracle_render_fixtures.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["HALO_WRITE_ORACLE_FIXTURES"] = "1"

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
