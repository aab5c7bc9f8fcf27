"""The committed oracle renders (tests/golden/oracle_renders/, tests/_oracle_cache.py) are only as good as the oracle that rendered them:
this CPU test fails when a fixture's fingerprint (oracle sources, scene builders, the compute function's text, the JSON document and its
reader) is no longer the one in MANIFEST.json — before the `-m gpu` suite would compare the HIP path with a stale oracle — and re-renders
the two cheapest fixtures live to show that a fixture is what the oracle says today."""
import os

import numpy as np

from tests import _oracle_cache as C


def _probe_all():
    from tests import test_gpu_filter_production as F
    from tests import test_gpu_parity as P
    from tests import test_gpu_production_routes as R
    C._seen.clear()
    C.PROBE = True
    try:
        for seed in (42, 7):
            R.oracle_config3(seed)
            for name in P.e2e_multilayer_names():
                P.oracle_e2e_multilayer(name, seed)
            for name in F.multilayer_docs():
                F.oracle_doc_multilayer(name, seed)
    finally:
        C.PROBE = False
    return dict(C._seen)


def test_every_committed_oracle_render_is_current():
    seen = _probe_all()
    stale = sorted(k for k, (_, fresh) in seen.items() if not fresh)
    assert not stale, "oracle renders older than their inputs: %s — run python tests/golden/make_oracle_render_fixtures.py --stale" % stale
    on_disk = {f[:-4] for f in os.listdir(C.DIR) if f.endswith(".npz")}
    assert on_disk == set(seen), (sorted(on_disk - set(seen)), sorted(set(seen) - on_disk))   # no orphan file, no missing one
    assert set(C.load_manifest()) == on_disk


def test_a_changed_input_makes_a_fixture_stale(monkeypatch):
    """the guard itself: another oracle source (here: one more byte) and the same key is no longer handed out from the file"""
    from tests import test_gpu_production_routes as R
    real = C._file_sha
    monkeypatch.setattr(C, "_file_sha", lambda rel: real(rel) if rel != "oracle/halo_oracle.c" else C._sha(b"edited"))
    C.PROBE = True
    try:
        R.oracle_config3(42)
    finally:
        C.PROBE = False
    assert C._seen["config3_two_layer_seed42"][1] is False


def test_cheapest_fixtures_rerendered_live():
    """two multi-layer e2e documents at their fixture's own size (200 k roots): the first layer's continuation count is a function of the
    seeded rays alone (exact); what depends on the oracle's thread schedule (the order continuations reach the next layer in) agrees within
    the cross-realisation spread"""
    from tests import test_gpu_parity as P
    for name in ("crystal_sample_count_zero_proportion", "ms_prob05"):
        fix = P.oracle_e2e_multilayer(name, 42)
        C.FORCE_LIVE = True
        try:
            live = P.oracle_e2e_multilayer(name, 42)
        finally:
            C.FORCE_LIVE = False
        assert int(live["cont"][0]) == int(fix["cont"][0]), name
        assert abs(float(live["landed"]) - float(fix["landed"])) <= 2e-2 * float(fix["landed"]), name
        assert abs(int(live["n_exits"]) - int(fix["n_exits"])) <= 2e-2 * int(fix["n_exits"]), name
        a, b = np.asarray(live["y16"], np.float64).ravel(), np.asarray(fix["y16"], np.float64).ravel()
        if b.std() > 0:
            assert np.corrcoef(a, b)[0, 1] > 0.9, name
