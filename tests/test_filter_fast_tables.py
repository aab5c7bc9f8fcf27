"""The fast form of the emit-gate filter (csrc/halo_device.h FastTables — what the production filter kernels evaluate) against the
oracle's reduction-based predicate (oracle/halo_oracle.c ho_filter_check, restating shared/filter_shared.h:53-315), on the host.

The fast form does no symmetry reduction per exit: the host lists, per raypath term, every face sequence whose reduction equals the
term's canonical form (an image of that form under the P / B / D group that the reference's own reduction maps back onto it), and
turns an entry/exit term into a bit matrix.  Whether that is the SAME predicate is a finite question for short paths, so it is
answered exhaustively here: every sequence of up to 5 prism faces and up to 3 pyramid faces, every symmetry subset, D applicable
and not, several roll centres (sigma_a), filter_in / filter_out, simple and complex filters.  No GPU.
"""
import ctypes as C
import itertools

import numpy as np
import pytest

from ice_halo_sim_amd import abi, backend, scenes

from _libs import oracle

PRISM_FACES = [1, 2, 3, 4, 5, 6, 7, 8]
PYRAMID_FACES = [1, 2] + list(range(3, 9)) + list(range(13, 19)) + list(range(23, 29))
SYMS = ["", "P", "B", "D", "PB", "PD", "BD", "PBD"]
UNI360 = {"type": "uniform", "mean": 0, "std": 360}


def axes():
    """orientation distributions: D applicable (azimuth symmetric, roll centre a multiple of 30) with several sigma_a, and not applicable"""
    out = []
    for roll_mean in (0.0, 30.0, 90.0, 150.0, -60.0):
        out.append(scenes.axis(zenith={"type": "gauss", "mean": 90, "std": 0.3}, azimuth=UNI360, roll={"type": "gauss", "mean": roll_mean, "std": 1.0}))
    out.append(scenes.axis(zenith={"type": "gauss", "mean": 90, "std": 0.3}, azimuth=UNI360, roll={"type": "gauss", "mean": 10.0, "std": 1.0}))   # not a multiple of 30
    out.append(scenes.axis(zenith={"type": "gauss", "mean": 0, "std": 1.0}, roll=UNI360))   # azimuth not symmetric
    return out


def both(f, ax, path, d=(0.0, 0.0, -1.0), cid=3):
    L, O = backend.load_library(), oracle()
    p = (C.c_uint8 * 16)(*path)
    dv = (C.c_float * 3)(*d)
    got = C.c_int32(-1)
    rc = L.halo_host_filter_fast_check(C.byref(f), C.byref(ax), p, len(path), dv, cid, C.byref(got))
    assert rc == 0
    want = O.ho_filter_check(C.byref(f), C.byref(ax), C.cast(p, C.POINTER(C.c_uint8)), len(path), C.cast(dv, C.POINTER(C.c_float)), cid)
    return got.value, int(want != 0)


def sweep(f, ax, alphabet, max_len):
    bad = []
    n = matched = 0
    for ln in range(1, max_len + 1):
        for path in itertools.product(alphabet, repeat=ln):
            g, w = both(f, ax, path)
            n += 1
            matched += w
            if g != w:
                bad.append((path, g, w))
    return n, matched, bad


@pytest.mark.parametrize("sym", SYMS)
def test_raypath_members_equal_the_reduction_predicate_prism(sym):
    # the reference's own benchmark and example filters: [3,5], [1,3,5], [4,6], [3,1,5,7,4]; plus basal-only and repeated faces
    targets = [[3, 5], [1, 3, 5], [4, 6], [3, 1, 5, 7, 4], [1, 2], [3, 3], [8, 2, 4]]
    for ax in axes()[:3] + axes()[-2:]:
        for rp in targets:
            if len(rp) > 4 and sym not in ("PBD", ""):
                continue   # the 5-face target over 8^5 sequences once per extreme is enough
            f = scenes.simple_filter(scenes.filter_term("raypath", raypath=rp), symmetry=sym)
            n, matched, bad = sweep(f, ax, PRISM_FACES, len(rp))
            assert matched >= 1, (sym, rp)
            assert not bad, (sym, rp, bad[:5])


@pytest.mark.parametrize("sym", ["", "P", "B", "PB", "BD", "PBD"])
def test_raypath_members_equal_the_reduction_predicate_pyramid(sym):
    targets = [[13, 25], [1, 15, 3], [23, 5], [13, 2, 27]]
    for ax in (axes()[0], axes()[2], axes()[-1]):
        for rp in targets:
            f = scenes.simple_filter(scenes.filter_term("raypath", raypath=rp), symmetry=sym)
            n, matched, bad = sweep(f, ax, PYRAMID_FACES, len(rp))
            assert matched >= 1
            assert not bad, (sym, rp, bad[:5])


@pytest.mark.parametrize("sym", SYMS)
def test_entry_exit_matrix_equals_the_reduction_predicate(sym):
    cases = [dict(entry=3, exit=5), dict(entry=3), dict(exit=1), dict(), dict(entry=13, exit=25), dict(entry=1, exit=2), dict(entry=4, exit=4)]
    for ax in axes()[:2] + axes()[-2:]:
        for kw in cases:
            for (mn, mx) in ((1, None), (2, 4), (3, 3)):
                f = scenes.simple_filter(scenes.filter_term("entry_exit", min_len=mn, max_len=mx, **kw), symmetry=sym)
                n, matched, bad = sweep(f, ax, PYRAMID_FACES, 2)
                assert not bad, (sym, kw, mn, mx, bad[:5])
                n, matched, bad = sweep(f, ax, PRISM_FACES, 4)
                assert not bad, (sym, kw, mn, mx, bad[:5])


def test_complex_filters_and_actions():
    T = scenes.filter_term
    ax = axes()[0]
    rng = np.random.default_rng(5)
    filters = [
        # the reference's benchmark filter (ms_multi_crystal_complex_filter.json): [3,5] OR [1,3,5], PBD
        scenes.complex_filter([[T("raypath", raypath=[3, 5])], [T("raypath", raypath=[1, 3, 5])]], symmetry="PBD"),
        # config_example.json filter 7: none OR (raypath AND crystal) OR direction filter_out semantics folded in a clause
        scenes.complex_filter([[T("none")], [T("raypath", raypath=[3, 1, 5, 7, 4]), T("crystal", crystal_id=3)], [T("direction", az=180, el=25, radii=0.5)]], symmetry="PBD"),
        scenes.complex_filter([[T("entry_exit", entry=3, exit=5, min_len=2, max_len=4), T("crystal", crystal_id=3)], [T("raypath", raypath=[1, 3, 2])]], symmetry="PB", action="filter_out"),
        scenes.complex_filter([[T("crystal", crystal_id=9)], [T("direction", az=0, el=-20, radii=30)]], symmetry="", action="filter_out"),
        scenes.complex_filter([], symmetry="P"),   # an empty complex filter matches nothing
        scenes.complex_filter([], symmetry="P", action="filter_out"),
        scenes.simple_filter(T("none")),
        scenes.simple_filter(T("none"), action="filter_out"),
        scenes.simple_filter(T("direction", az=180, el=25, radii=0.5), action="filter_out"),
        scenes.simple_filter(T("crystal", crystal_id=3)),
    ]
    for f in filters:
        bad = []
        seen = set()
        for _ in range(3000):
            ln = int(rng.integers(1, 7))
            path = tuple(int(x) for x in rng.choice(PRISM_FACES, size=ln))
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            if rng.random() < 0.3:   # inside the 0.5 degree / 30 degree cones now and then
                lon, lat = np.radians(180.0), np.radians(25.0)
                d = np.array([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)]) + rng.normal(size=3) * 2e-3
                d /= np.linalg.norm(d)
            cid = int(rng.choice([3, 9, 1]))
            g, w = both(f, ax, path, d=tuple(float(x) for x in d), cid=cid)
            seen.add(w)
            if g != w:
                bad.append((path, cid, g, w))
        assert not bad, bad[:5]
    # the benchmark filter once more, exhaustively over every path of up to 5 prism faces
    n, matched, bad = sweep(filters[0], ax, PRISM_FACES, 5)
    assert matched > 0 and not bad, bad[:5]


def test_long_paths_up_to_the_register():
    # 16 faces fill the 128-bit register; 9..16 use both halves
    rng = np.random.default_rng(9)
    ax = axes()[0]
    for ln in (8, 9, 12, 16):
        for sym in ("", "P", "PBD"):
            rp = [int(x) for x in rng.choice(PRISM_FACES, size=ln)]
            f = scenes.simple_filter(scenes.filter_term("raypath", raypath=rp), symmetry=sym)
            # every image of rp under the group + random perturbations of it
            hits = 0
            for rot in range(6):
                for flip in (0, 1):
                    q = [x if x < 3 else (x - 3 + rot) % 6 + 3 for x in rp]
                    if flip:
                        q = [3 - x if x < 3 else x for x in q]
                    g, w = both(f, ax, q)
                    assert g == w, (ln, sym, q)
                    hits += w
                    q2 = list(q)
                    q2[int(rng.integers(0, ln))] = int(rng.choice(PRISM_FACES))
                    g, w = both(f, ax, q2)
                    assert g == w, (ln, sym, q2)
            assert hits >= 1
