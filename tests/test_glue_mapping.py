"""integration/hip_backend_glue.hpp EXECUTED, not only type-checked: its `hip_glue::ToHalo` mapping of the reference's config
structs — SceneConfig, CrystalParam (prism / pyramid, sync groups, wedge angles), AxisDistribution, FilterConfig (every term kind, complex
filters), RenderConfig (every lens), the raypath-colour tables — runs in tests/cpp/glue_mapping_main.cpp on scenes built programmatically like
/root/reference/test/cpu_test_helpers.hpp:26-68 does, and the Halo* structs it fills are compared BYTE FOR BYTE with what
ice_halo_sim_amd/config.py makes of the equivalent JSON documents (the documents below).  A field the glue swaps (latitude for zenith, a
sync_group slot, exit for entry), drops or mis-scales fails here; so does a colour bit the Python restatement of BuildColorGateTable assigns
differently from the reference's own builder, which the glue calls.

Build container only (`ref` marker): needs /root/reference and g++.  Nothing it builds is kept (tools/glue_mapping_build.sh)."""
import ctypes as C
import json
import os
import subprocess

import pytest

from ice_halo_sim_amd import abi, config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.ref, pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs the reference checkout (build container)")]

COL = {"zenith": {"type": "gauss", "mean": 90, "std": 0.3}}
LENSES = ["linear", "fisheye_equal_area", "fisheye_equidistant", "fisheye_stereographic", "dual_fisheye_equal_area", "dual_fisheye_equidistant",
          "dual_fisheye_stereographic", "rectangular", "fisheye_orthographic", "dual_fisheye_orthographic", "globe"]


def _light(alt, az, diameter):
    return {"type": "sun", "altitude": alt, "azimuth": az, "diameter": diameter, "spectrum": [{"wavelength": 550, "weight": 1.0}]}


def _docs():
    docs = {}
    docs["prism_all_lenses"] = {
        "crystal": [{"id": 3, "type": "prism", "shape": {"height": 1.3}, "axis": COL}],
        "scene": {"ray_num": 1000, "max_hits": 7, "light_source": _light(20, 0, 0.5), "scattering": [{"prob": 0.0, "entries": [{"crystal": 3, "proportion": 10}]}]},
        "render": [{"id": i + 1, "lens": {"type": LENSES[i], "fov": 40 + 5 * i}, "resolution": [640 + 16 * i, 360 + 8 * i], "lens_shift": [i, -2 * i],
                    "view": {"azimuth": 10 * i, "elevation": 5 + i, "roll": -3 * i}, "visible": ["upper", "lower", "full"][i % 3], "overlap": 0.25 if i == 4 else 0.0}
                   for i in range(11)]}
    g = {"type": "gauss", "mean": 1.0, "std": 0.15}
    full = {"type": "uniform", "mean": 0, "std": 360}
    docs["pyramid_miller_stochastic"] = {
        "crystal": [{"id": 5, "type": "pyramid", "shape": {"upper_h": 0.1, "prism_h": 1.2, "lower_h": 0.5, "upper_indices": [2, 0, 3], "lower_indices": [1, 0, 1], "face_distance": [g] * 6},
                     "axis": {"zenith": full, "azimuth": full, "roll": full}}],
        "scene": {"ray_num": 1000, "max_hits": 8, "light_source": _light(35, 120, 1.0), "scattering": [{"prob": 0.0, "entries": [{"crystal": 5, "proportion": 100}]}]},
        "render": [{"id": 1, "lens": {"type": "rectangular", "fov": 0}, "resolution": [2048, 1024], "visible": "full"}]}
    docs["sync_groups_and_axes"] = {
        "crystal": [
            {"id": 1, "type": "prism",
             "shape": {"height": {"type": "gauss", "mean": 1.2, "std": 0.1},
                       "face_distance": [{"type": "gauss", "mean": 1.0, "std": 0.1}, 1.0, {"type": "uniform", "mean": 1.0, "std": 0.2}, 1.0, 0.9, 1.0],
                       "sync_group": {"face_distance": [4, 4, 9, 9, 7, 0]}},      # 4 -> 1, 9 -> 2 by first appearance; 7 is a one-member group -> 0
             "axis": {"zenith": {"type": "laplacian", "mean": 30, "std": 2}, "azimuth": {"type": "zigzag", "mean": 10, "std": 20}, "roll": 15}},
            {"id": 2, "type": "pyramid",
             "shape": {"upper_h": {"type": "uniform", "mean": 0.4, "std": 0.2}, "prism_h": 0.7, "lower_h": 0.2, "upper_wedge_angle": 31.5, "lower_wedge_angle": 24.25,
                       "sync_group": {"upper_h": 2, "prism_h": 2, "lower_h": 2}},
             "axis": {"zenith": {"type": "gauss_legacy", "mean": 80, "std": 5}, "azimuth": {"type": "gauss", "mean": 45, "std": 3}, "roll": {"type": "uniform", "mean": 30, "std": 60}}}],
        "scene": {"ray_num": 1000, "max_hits": 5, "light_source": _light(10, 0, 0.5),
                  "scattering": [{"prob": 0.0, "entries": [{"crystal": 1, "proportion": 2}, {"crystal": 2, "proportion": 3}]}]},
        "render": [{"id": 2, "lens": {"type": "linear", "fov": 60}, "resolution": [800, 600], "view": {"elevation": 20}}]}
    colax, plateax = {"zenith": {"type": "gauss", "mean": 90, "std": 0.5}}, {"zenith": {"type": "gauss", "mean": 0, "std": 1.0}}
    docs["filter_terms_three_layers"] = {
        "crystal": [{"id": 1, "type": "prism", "shape": {"height": 1.5}, "axis": colax}, {"id": 2, "type": "prism", "shape": {"height": 0.3}, "axis": plateax},
                    {"id": 7, "type": "prism", "shape": {"height": 1.0}, "axis": colax}],
        "filter": [{"id": 1, "type": "raypath", "raypath": [3, 5], "symmetry": "P"},
                   {"id": 2, "type": "entry_exit", "entry": 1, "exit": 3, "min_len": 2, "max_len": 5, "symmetry": "PBD"},
                   {"id": 3, "type": "direction", "az": 180, "el": 20, "radii": 2, "action": "filter_out"},
                   {"id": 4, "type": "entry_exit", "exit": 8, "symmetry": "B"},
                   {"id": 5, "type": "crystal", "crystal_id": 2, "action": "filter_out"}],
        "scene": {"ray_num": 1000, "max_hits": 9, "light_source": _light(25, 0, 0.5),
                  "scattering": [{"prob": 0.6, "entries": [{"crystal": 1, "proportion": 1, "filter": 1}, {"crystal": 2, "proportion": 2.5, "filter": 2}, {"crystal": 7, "proportion": 0.5}]},
                                 {"prob": 0.25, "entries": [{"crystal": 1, "proportion": 1, "filter": 3}, {"crystal": 2, "proportion": 1, "filter": 4}]},
                                 {"prob": 0.0, "entries": [{"crystal": 1, "proportion": 1, "filter": 5}]}]},
        "render": [{"id": 1, "lens": {"type": "dual_fisheye_equal_area", "fov": 180}, "resolution": [1024, 512], "view": {"elevation": 90}, "visible": "full", "overlap": 0.1}]}
    docs["complex_filters"] = {
        "crystal": [{"id": 3, "type": "prism", "shape": {"height": 1.3}, "axis": COL}, {"id": 6, "type": "prism", "shape": {"height": 0.3}, "axis": {"zenith": {"type": "gauss", "mean": 0, "std": 0.8}}}],
        "filter": [{"id": 1, "type": "raypath", "raypath": [1, 3, 2]}, {"id": 2, "type": "entry_exit", "entry": 1}, {"id": 3, "type": "crystal", "crystal_id": 3},
                   {"id": 4, "type": "raypath", "raypath": [3, 1, 5, 7, 4]},
                   {"id": 10, "type": "complex", "composition": [1, [2, 3], 4], "symmetry": "PBD"},
                   {"id": 11, "type": "complex", "composition": [[2, 1], 3], "symmetry": "D", "action": "filter_out"}],
        "scene": {"ray_num": 1000, "max_hits": 8, "light_source": _light(20, 0, 0.5),
                  "scattering": [{"prob": 0.0, "entries": [{"crystal": 3, "proportion": 10, "filter": 10}, {"crystal": 6, "proportion": 4, "filter": 11}]}]},
        "render": [{"id": 4, "lens": {"type": "fisheye_equal_area", "fov": 120}, "resolution": [1920, 1080], "view": {"elevation": 30}}]}
    docs["raypath_color_two_layers"] = {
        "crystal": [{"id": 1, "type": "prism", "shape": {"height": 1.4}, "axis": {"zenith": {"type": "gauss", "mean": 90, "std": 0.4}}},
                    {"id": 2, "type": "prism", "shape": {"height": 0.25}, "axis": plateax}],
        "scene": {"ray_num": 1000, "max_hits": 7, "light_source": _light(20, 0, 0.5),
                  "scattering": [{"prob": 0.5, "entries": [{"crystal": 1, "proportion": 1}, {"crystal": 2, "proportion": 1}]}, {"prob": 0.0, "entries": [{"crystal": 1, "proportion": 1}]}]},
        "render": [{"id": 1, "lens": {"type": "fisheye_equal_area", "fov": 180}, "resolution": [512, 256], "view": {"elevation": 30}}],
        "raypath_color": [
            {"color": [1, 0, 0], "match": [{"layer": 0, "crystal": 1, "type": "raypath", "raypath": [3, 5], "symmetry": "P"},
                                           {"layer": 1, "crystal": 1, "type": "entry_exit", "entry": 3, "exit": 5, "min_len": 2, "max_len": 4, "symmetry": "PB"}]},
            {"color": [0, 1, 0], "combine": "all", "match": [{"layer": 0, "crystal": 1, "type": "raypath", "raypath": [3, 5], "symmetry": "P"}, {"layer": 0, "crystal": 2}]},
            {"color": [0, 0, 1], "match": [{"layer": 0, "crystal": 2, "type": "direction", "az": 0, "el": 22, "radii": 3}]}]}
    return docs


@pytest.fixture(scope="module")
def glue_output():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "glue_mapping_build.sh")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            out[d["name"]] = d
    return out


def _hex(struct):
    return bytes(struct).hex()


def _canon_terms(hexs, ctype):
    """Fields a term kind never reads are don't-cares: scenes.filter_term seeds `min_len` = 1 (EntryExitFilterParam's default) on every kind,
    the glue writes it for entry_exit terms only; the matchers read min_len / max_len inside the entry_exit case alone (halo_trace.inl
    filter_check, halo_host.cpp BuildFilter, oracle ho_filter_check).  They are zeroed on both sides for the other kinds before the compare."""
    if not hexs:
        return hexs
    obj = ctype.from_buffer_copy(bytes.fromhex(hexs))
    terms = [obj.terms[k] for k in range(len(obj.terms))] if ctype is abi.HaloFilter else [obj.terms[k].predicate for k in range(len(obj.terms))]
    for t in terms:
        if t.type != abi.FILTER_ENTRY_EXIT:
            t.min_len = t.max_len = 0
    return bytes(obj).hex()


def _diff(name, a, b, ctype):
    """the first fields of `ctype` whose bytes differ between two hex strings (to say WHAT differs, not only that something does)"""
    if a == b:
        return None
    ab, bb = bytes.fromhex(a), bytes.fromhex(b)
    if len(ab) != len(bb):
        return "%s: sizes differ (%d vs %d bytes)" % (name, len(ab), len(bb))
    first = next(i for i in range(len(ab)) if ab[i] != bb[i])
    where = []
    for fname, ftype in ctype._fields_:
        off = getattr(ctype, fname).offset
        if off <= first < off + C.sizeof(ftype):
            where.append(fname)
    return "%s: first differing byte %d of %d (field %s): glue %s / config.py %s" % (name, first, len(ab), where, ab[first:first + 8].hex(), bb[first:first + 8].hex())


@pytest.mark.parametrize("name", sorted(_docs()))
def test_glue_mapping_equals_the_json_readers(name, glue_output):
    doc = _docs()[name]
    job = config.load_config(doc)
    g = glue_output[name]
    assert g["representable"] is True and g["color_class_overflow"] == 0 and g["color_term_overflow"] == 0
    # the scene with the table indices taken out (the glue appends one table row per filtered entry, config.py one per filter id: the
    # rows an entry points AT are compared below, entry by entry)
    sc = abi.HaloScene.from_buffer_copy(bytes(job.scene))
    gs = abi.HaloScene.from_buffer_copy(bytes.fromhex(g["scene"]))
    per_entry = {}
    for l in range(sc.layer_count):
        for e in range(sc.layers[l].entry_count):
            he = sc.layers[l].entries[e]
            per_entry[(l, e)] = (_hex(job.filters[he.filter_id - 1]) if he.filter_id > 0 else "", _hex(job.color_sets[he.color_id - 1]) if he.color_id > 0 else "")
            assert (he.filter_id > 0) == (gs.layers[l].entries[e].filter_id > 0), (name, l, e, "filtered?")
            assert (he.color_id > 0) == (gs.layers[l].entries[e].color_id > 0), (name, l, e, "coloured?")
            he.filter_id = he.color_id = 0
            gs.layers[l].entries[e].filter_id = gs.layers[l].entries[e].color_id = 0
    assert (gs.layer_count, [gs.layers[l].entry_count for l in range(gs.layer_count)]) == (sc.layer_count, [sc.layers[l].entry_count for l in range(sc.layer_count)])
    for l in range(sc.layer_count):          # entry by entry first: a readable message
        for e in range(sc.layers[l].entry_count):
            d = _diff("%s layer %d entry %d" % (name, l, e), _hex(gs.layers[l].entries[e]), _hex(sc.layers[l].entries[e]), abi.HaloEntry)
            assert d is None, d
    d = _diff(name + " scene", _hex(gs), _hex(sc), abi.HaloScene)
    assert d is None, d
    assert len(g["entries"]) == len(per_entry)
    for ent in g["entries"]:
        want_f, want_c = per_entry[(ent["layer"], ent["entry"])]
        d = _diff("%s filter of layer %d entry %d" % (name, ent["layer"], ent["entry"]), _canon_terms(ent["filter"], abi.HaloFilter), _canon_terms(want_f, abi.HaloFilter), abi.HaloFilter) if want_f or ent["filter"] else None
        assert d is None, d
        d = _diff("%s colour set of layer %d entry %d" % (name, ent["layer"], ent["entry"]), _canon_terms(ent["color"], abi.HaloColorSet), _canon_terms(want_c, abi.HaloColorSet), abi.HaloColorSet) if want_c or ent["color"] else None
        assert d is None, d
    assert g["classes"] == [_hex(c) for c in job.color_classes], name
    assert g["renders"] == [_hex(job.renders[k]) for k in sorted(job.renders)], name


def test_colour_past_the_caps_is_counted_and_the_tables_keep_what_fits(glue_output):
    """VERDICT r5 missing #7: the reference's GPU backends drop raypath-colour classes / groups past their caps and COUNT them
    (GetLastColorDegradeCounts, trace_backend.hpp:626-632; cuda_trace_backend.cu:3296-3304).  The glue counts the same way (19 classes -> 3
    over the shared cap of 16; 17 predicates on one placement -> 1 over), keeps the first 16 of each byte for byte as in the scene that
    fits, and BeginSession either refuses (default: the simulator falls back to the legacy path, which has no caps) or, under
    LUMICE_HIP_COLOR_OVERFLOW=degrade, renders and reports the counts."""
    g, base = glue_output["raypath_color_past_the_caps"], glue_output["raypath_color_two_layers"]
    assert g["representable"] is True
    assert (g["color_class_overflow"], g["color_term_overflow"]) == (3, 1)
    assert len(g["classes"]) == abi.COLOR_MAX_CLASSES and g["classes"][:3] == base["classes"]
    sets = {(e["layer"], e["entry"]): e["color"] for e in g["entries"]}
    base_sets = {(e["layer"], e["entry"]): e["color"] for e in base["entries"]}
    assert sets[(0, 1)] == base_sets[(0, 1)] and sets[(1, 0)] == base_sets[(1, 0)]        # the placements the extra classes do not touch
    full = abi.HaloColorSet.from_buffer_copy(bytes.fromhex(sets[(0, 0)]))
    small = abi.HaloColorSet.from_buffer_copy(bytes.fromhex(base_sets[(0, 0)]))
    assert full.term_count == abi.COLOR_MAX_TERMS and small.term_count == 1
    assert bytes(full.terms[0]) == bytes(small.terms[0])                                  # the shared predicate keeps its bit
    assert sorted(full.terms[k].bit for k in range(full.term_count)) == sorted(set(full.terms[k].bit for k in range(full.term_count)))   # one bit per predicate
    src = open(os.path.join(ROOT, "integration", "hip_backend_glue.hpp")).read()
    assert "LUMICE_HIP_COLOR_OVERFLOW" in src and "last_color_degrade_.color_class_overflow = t.color_class_overflow" in src


def test_a_swapped_field_in_the_glue_is_caught(glue_output, tmp_path):
    """the check itself: a copy of the glue with view azimuth and elevation exchanged (and one with the prism height's sync slot moved) no
    longer produces config.py's bytes"""
    src = open(os.path.join(ROOT, "integration", "hip_backend_glue.hpp")).read()
    for old, new in (("o.view_az = r.view_.az_;\n  o.view_el = r.view_.el_;", "o.view_az = r.view_.el_;\n  o.view_el = r.view_.az_;"),
                     ("c.sync_group[3 + i] = p.sync_group_[kShapeScalarFace0 + i];", "c.sync_group[3 + i] = p.sync_group_[kShapeScalarFace0 + (i + 1) % 6];")):
        assert old in src
        tree = tmp_path / ("t%d" % (hash(old) & 0xffff))
        (tree / "integration").mkdir(parents=True)
        for d in ("tools", "tests/cpp", "include", "ice_halo_sim_amd/csrc"):
            (tree / d).parent.mkdir(parents=True, exist_ok=True)
            os.symlink(os.path.join(ROOT, d), tree / d)
        (tree / "integration" / "hip_backend_glue.hpp").write_text(src.replace(old, new))
        script = (tree / "run.sh")
        script.write_text(open(os.path.join(ROOT, "tools", "glue_mapping_build.sh")).read().replace('ROOT=$(cd "$(dirname "$0")/.." && pwd)', 'ROOT=%s' % tree))
        r = subprocess.run(["bash", str(script)], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-1500:]
        got = {json.loads(l)["name"]: json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")}
        assert any(got[n]["scene"] != glue_output[n]["scene"] or got[n]["renders"] != glue_output[n]["renders"] for n in got), old
