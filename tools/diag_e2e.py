import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
from ice_halo_sim_amd import config, scenes
from ice_halo_sim_amd.backend import HipTraceBackend
from tests._oracle_backend import OracleBackend, run_session
from tests.test_gpu_parity import match_exits
name = sys.argv[1]
opts = dict(kv.split("=") for kv in sys.argv[2:])
doc = json.load(open("tests/golden/ref_e2e_configs.json"))[name]
job = config.load_config(doc)
rd = job.renders[sorted(job.renders)[0]]
wl = job.wavelengths[0]
n = 120_000
hb = HipTraceBackend(device=0, seed=42, capture_exits=1, **{k: int(v) for k, v in opts.items()})
ob = OracleBackend(seed=42, capture_exits=1, threads=8)
for b in (hb, ob):
    b.set_filters(job.filters)
run_session(hb, job.scene, rd, wl, n); run_session(ob, job.scene, rd, wl, n)
eh, eo = hb.DrainExits(), ob.DrainExits()
print(os.environ.get("HALO_LIB", "product"), opts, "exits", len(eh), len(eo), "match", match_exits(eh, eo))
kh = set(zip(eh["root"].tolist(), eh["seq"].tolist())); ko = set(zip(eo["root"].tolist(), eo["seq"].tolist()))
print("only hip", len(kh - ko), "only oracle", len(ko - kh))
for r, s in list(kh - ko)[:5]:
    e = eh[(eh["root"] == r) & (eh["seq"] == s)][0]; print(" hip-only", r, s, e["dir"], e["weight"], list(e["path"][:e["path_len"]]))
for r, s in list(ko - kh)[:5]:
    e = eo[(eo["root"] == r) & (eo["seq"] == s)][0]; print(" ora-only", r, s, e["dir"], e["weight"], list(e["path"][:e["path_len"]]))
