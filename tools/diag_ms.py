import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
from ice_halo_sim_amd import config
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import OracleBackend, run_session
name = sys.argv[1]
doc = json.load(open("tests/golden/ref_e2e_configs.json"))[name]
job = config.load_config(doc)
rd = job.renders[sorted(job.renders)[0]]
wl = job.wavelengths[0]
n = 200_000
def run(b):
    if job.geom_clock: b.set_option("geom_clock", job.geom_clock)
    b.set_filters(job.filters)
    if job.color_classes: b.set_color(job.color_sets, job.color_classes)
    s = run_session(b, job.scene, rd, wl, n)
    e = b.DrainExits(); im, l = b.ReadbackXyzAccum()
    r = b.last_route() if hasattr(b, "last_route") else None
    b.close()
    return [x.continuation_count for x in s], len(e), l, (r.geom_mask, r.accum_mask, r.mode_mask) if r else None
for seed in (42, 7):
    print("oracle seed", seed, run(OracleBackend(seed=seed, capture_exits=1, threads=8)))
for opts in ({}, {"hex_fast": 0}, {"entry_fast": 0}, {"hit_log": 0}):
    for seed in (42, 43):
        print("hip", opts, seed, run(HipTraceBackend(device=0, seed=seed, capture_exits=1, **opts)))
