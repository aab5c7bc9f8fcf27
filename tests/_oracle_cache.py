"""Oracle renders that take the CPU oracle tens of seconds (multi-layer scenes at production launch sizes) as committed fixtures.

The `-m gpu` suite compares the HIP path with the oracle on the same seeded inputs; three of those comparisons spent 150 s of every run
re-rendering the ORACLE side, which depends on nothing that changes between runs.  Their oracle halves are now functions that return a
summary (block-mean images, landed weight, continuation and exit counts — exactly what the test reads), computed once by
tests/golden/make_oracle_render_fixtures.py (build container; runs liboracle.so only) and committed under tests/golden/oracle_renders/.
A test finds its summary there; if the file is missing it renders live, as before.  The fixtures are data (inputs are the test's own
seeded scene; expected outputs are the oracle's), and the script that made them is committed with them."""
import os

import numpy as np

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_renders")


def cached(key, compute):
    """summary dict {name: ndarray} of oracle render `key`: the committed fixture, else compute() (written out when
    HALO_WRITE_ORACLE_FIXTURES is set — the generator script)."""
    path = os.path.join(DIR, key + ".npz")
    if os.path.exists(path) and not os.environ.get("HALO_WRITE_ORACLE_FIXTURES"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    out = {k: np.asarray(v) for k, v in compute().items()}
    if os.environ.get("HALO_WRITE_ORACLE_FIXTURES"):
        os.makedirs(DIR, exist_ok=True)
        np.savez_compressed(path, **out)
    return out
