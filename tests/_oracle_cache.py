"""Oracle renders that take the CPU oracle tens of seconds (multi-layer scenes at production launch sizes) as committed fixtures.

The `-m gpu` suite compares the HIP path with the oracle on the same seeded inputs; three of those comparisons spent 150 s of every run
re-rendering the ORACLE side, which depends on nothing that changes between runs.  Their oracle halves are now functions that return a
summary (block-mean images, landed weight, continuation and exit counts — exactly what the test reads), computed once by
tests/golden/make_oracle_render_fixtures.py (build container; runs liboracle.so only) and committed under tests/golden/oracle_renders/.
The fixtures are data (inputs are the test's own seeded scene; expected outputs are the oracle's), and the script that made them is
committed with them.

Staleness guard (round 6).  A fixture is only as good as the oracle that rendered it.  MANIFEST.json beside the fixtures holds, per key, the
sha256 of everything that decides the render: the oracle's sources, the ctypes face the tests drive it through, the scene builders, the
source text of the function that computed the summary, and whatever the caller names as extra inputs (the JSON document, the reader that
turns it into a scene).  `cached()` hands a fixture out only while that fingerprint is the one in the manifest; a fixture whose inputs
moved is rendered live instead (correct, slow) — and tests/test_oracle_render_cache.py fails in the CPU suite, so the builder re-renders
before the GPU suite ever meets a stale file."""
import hashlib
import inspect
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "tests", "golden", "oracle_renders")
MANIFEST = os.path.join(DIR, "MANIFEST.json")
# what every oracle render depends on: the restatement itself and the path the tests reach it by
BASE_INPUTS = ("oracle/halo_oracle.c", "oracle/halo_oracle.h", "oracle/cie_tables_oracle.inc", "oracle/Makefile", "tests/_oracle_backend.py",
               "ice_halo_sim_amd/scenes.py")


def _sha(data):
    return hashlib.sha256(data if isinstance(data, bytes) else data.encode()).hexdigest()


def _file_sha(rel):
    with open(os.path.join(ROOT, rel), "rb") as f:
        return _sha(f.read())


def fingerprint(compute, inputs=()):
    """sha256 over the base inputs, the source of `compute` and the caller's extra inputs (repo-relative file paths, or ("name", text) pairs)"""
    parts = [(rel, _file_sha(rel)) for rel in BASE_INPUTS]
    parts.append(("compute", _sha(inspect.getsource(compute))))
    for item in inputs:
        if isinstance(item, tuple):
            parts.append((item[0], _sha(item[1])))
        else:
            parts.append((item, _file_sha(item)))
    return _sha(json.dumps(parts))


def load_manifest():
    if not os.path.exists(MANIFEST):
        return {}
    with open(MANIFEST) as f:
        return json.load(f)


def _store_manifest(m):
    os.makedirs(DIR, exist_ok=True)
    with open(MANIFEST, "w") as f:
        json.dump(m, f, indent=0, sort_keys=True)
        f.write("\n")


PROBE = False   # True: cached() only records whether the fixture of `key` is current and returns None (tests/test_oracle_render_cache.py)
FORCE_LIVE = False   # True: cached() renders whatever the file says (the live half of tests/test_oracle_render_cache.py)
_seen = {}   # key -> (fingerprint now, fresh?) of every cached() call of this process (tests/test_oracle_render_cache.py reads it)


def cached(key, compute, inputs=()):
    """summary dict {name: ndarray} of oracle render `key`: the committed fixture while its fingerprint is current, else compute()
    (written out, manifest entry included, when HALO_WRITE_ORACLE_FIXTURES is set — the generator script; HALO_STAMP_ORACLE_FIXTURES only
    records the fingerprint of an existing file: the generator's --stamp mode, which proves from git history that no input moved)."""
    path = os.path.join(DIR, key + ".npz")
    fp = fingerprint(compute, inputs)
    if os.environ.get("HALO_STAMP_ORACLE_FIXTURES"):
        assert os.path.exists(path), path
        m = load_manifest()
        m[key] = fp
        _store_manifest(m)
    fresh = os.path.exists(path) and load_manifest().get(key) == fp
    _seen[key] = (fp, fresh)
    if PROBE:
        return None
    if fresh and not FORCE_LIVE and not os.environ.get("HALO_WRITE_ORACLE_FIXTURES"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    out = {k: np.asarray(v) for k, v in compute().items()}
    if os.environ.get("HALO_WRITE_ORACLE_FIXTURES"):
        os.makedirs(DIR, exist_ok=True)
        np.savez_compressed(path, **out)
        m = load_manifest()
        m[key] = fp
        _store_manifest(m)
    return out
