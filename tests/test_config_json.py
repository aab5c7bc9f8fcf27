"""JSON config surface (reference src/config/*.cpp parsers; doc/configuration.md): the documents below follow the
reference's published schema (keys `mean`/`std`, `zenith`, `upper_indices`, `scattering[].entries[]`, lens `f`)."""
import copy
import math
import json
import os

import pytest

from ice_halo_sim_amd import abi, config

DOC = {
    "crystal": [
        {"id": 1, "type": "prism", "shape": {"height": 1.2}},
        {"id": 3, "type": "prism", "shape": {"height": 1.3, "face_distance": [1, 1, 1, 1, 1, 1]},
         "axis": {"zenith": {"type": "gauss", "mean": 90, "std": 0.3}, "roll": {"type": "uniform", "mean": 0, "std": 360},
                  "azimuth": {"type": "uniform", "mean": 0, "std": 360}}},
        {"id": 4, "type": "prism", "shape": {"height": {"type": "uniform", "mean": 0.5, "std": 0.4},
                                             "face_distance": [{"type": "gauss", "mean": 1, "std": 0.2}] * 6,
                                             "sync_group": {"height": 7, "face_distance": [7, 2, 2, 0, 9, 0]}},
         "axis": {"zenith": {"type": "gauss", "mean": 0, "std": 1.2}}},
        {"id": 5, "type": "pyramid", "shape": {"upper_h": 0.1, "lower_h": 0.5, "prism_h": 1.2, "upper_indices": [2, 0, 3]},
         "axis": {"zenith": 0}},
    ],
    "filter": [{"id": 1, "type": "none"}, {"id": 2, "type": "raypath", "raypath": [3, 5], "symmetry": "P"}],
    "scene": {
        "light_source": {"type": "sun", "altitude": 20.0, "azimuth": 0, "diameter": 0.5,
                         "spectrum": [{"wavelength": 450 + 40 * i, "weight": 1.0} for i in range(9)]},
        "ray_num": 1000, "max_hits": 7,
        "scattering": [{"prob": 0.0, "entries": [{"crystal": 3, "proportion": 10, "filter": 1}, {"crystal": 5}]}],
    },
    "render": [
        {"id": 1, "lens": {"type": "linear", "f": 14}, "resolution": [1920, 1080], "lens_shift": [0, 200],
         "view": {"azimuth": -10, "elevation": 20, "roll": 0}},
        {"id": 4, "lens": {"type": "fisheye_equal_area", "fov": 120}, "resolution": [1920, 1080], "view": {"elevation": 30}},
    ],
}


def test_document_maps_onto_the_abi_structs():
    job = config.load_config(copy.deepcopy(DOC))
    sc = job.scene
    assert (sc.max_hits, sc.layer_count, sc.layers[0].entry_count) == (7, 1, 2)
    assert (sc.sun_altitude, sc.sun_azimuth, sc.sun_diameter) == (20.0, 0.0, 0.5)
    e = sc.layers[0].entries[0]
    assert e.crystal_config_id == 3 and e.proportion == 10.0 and e.crystal.kind == abi.CRYSTAL_PRISM
    assert e.crystal.height[0].center == pytest.approx(1.3)
    # zenith gauss(90, 0.3) → internal latitude gauss(0, 0.3) (math.cpp:679-714)
    assert (e.axis.latitude.type, e.axis.latitude.center, e.axis.latitude.spread) == (abi.DIST_GAUSS, 0.0, pytest.approx(0.3))
    assert (e.axis.azimuth.type, e.axis.azimuth.spread) == (abi.DIST_UNIFORM, 360.0)
    p = sc.layers[0].entries[1]
    assert p.crystal.kind == abi.CRYSTAL_PYRAMID and p.proportion == 100.0  # default proportion (config_manager.cpp:110)
    assert [p.crystal.height[i].center for i in range(3)] == [pytest.approx(0.1), pytest.approx(1.2), pytest.approx(0.5)]
    assert p.crystal.wedge_upper_deg == pytest.approx(math.degrees(math.atan(0.866025403784 * 3 / 2 / 1.629)), rel=1e-6)
    assert p.crystal.wedge_lower_deg == 28.0
    assert (p.axis.latitude.type, p.axis.latitude.center) == (abi.DIST_NONE, 90.0)
    assert p.axis.azimuth.type == abi.DIST_UNIFORM  # `axis` present → azimuth/roll default to uniform 360
    assert len(job.wavelengths) == 9 and job.wavelengths[2].wavelength == 530.0 and job.wavelengths[0].illuminant == -1
    assert job.per_wavelength_ray_num() == 112  # ceil(1000 / 9)
    r1, r4 = job.renders[1], job.renders[4]
    assert r1.lens_type == abi.LENS_LINEAR and r1.fov == pytest.approx(math.degrees(math.atan2(12.0, 14.0) * 2))
    assert (r1.lens_shift[0], r1.lens_shift[1], r1.view_az, r1.view_el) == (0, 200, -10.0, 20.0)
    assert (r4.lens_type, r4.fov, r4.visible, r4.view_el) == (abi.LENS_FISHEYE_EQUAL_AREA, 120.0, abi.VISIBLE_UPPER, 30.0)


def test_absent_axis_and_sync_groups():
    doc = copy.deepcopy(DOC)
    doc["scene"]["scattering"][0]["entries"] = [{"crystal": 1}, {"crystal": 4}]
    job = config.load_config(doc)
    a = job.scene.layers[0].entries[0].axis  # no `axis`: everything fixed, zenith 0 (math.cpp AxisDistribution())
    assert (a.latitude.type, a.latitude.center, a.azimuth.type, a.roll.type) == (abi.DIST_NONE, 90.0, abi.DIST_NONE, abi.DIST_NONE)
    c = job.scene.layers[0].entries[1].crystal
    # groups {7: height+d0, 2: d1+d2, 9: d4 alone} → canonical {1: height+d0, 2: d1+d2}, singleton dropped
    assert list(c.sync_group) == [1, 0, 0, 1, 2, 2, 0, 0, 0]
    # members take the leader's distribution (crystal_config.cpp:102-133): d0 follows height's uniform
    assert (c.face_dist[0].type, c.face_dist[0].center) == (abi.DIST_UNIFORM, pytest.approx(0.5))
    assert c.face_dist[1].type == abi.DIST_GAUSS


def test_partial_distribution_objects_keep_the_slot_defaults():
    """An object overwrites only the keys it carries; the rest keep the destination's seeded value (from_json(Distribution&)
    math.cpp:593-630 with the seeds of math.cpp:536-538 / :692-714 and crystal_config.hpp:63, crystal_config.cpp:310-313)."""
    c, a = config.parse_crystal({"id": 1, "type": "prism", "shape": {"height": {"type": "gauss", "std": 0.1},
                                                                      "face_distance": [{"type": "uniform", "std": 0.2}, 1.1]},
                                 "axis": {"zenith": {"type": "gauss", "std": 5}, "roll": {"type": "uniform"},
                                          "azimuth": {"type": "uniform", "mean": 10}}})
    # zenith slot is seeded {none, 90, 0}: missing mean = zenith 90 -> latitude 0
    assert (a.latitude.type, a.latitude.center, a.latitude.spread) == (abi.DIST_GAUSS, 0.0, 5.0)
    # azimuth / roll are seeded {uniform, 0, 360}
    assert (a.roll.type, a.roll.center, a.roll.spread) == (abi.DIST_UNIFORM, 0.0, 360.0)
    assert (a.azimuth.type, a.azimuth.center, a.azimuth.spread) == (abi.DIST_UNIFORM, 10.0, 360.0)
    # prism height is seeded {none, 1.0, 0}; face distances {none, 1.0, 0}
    assert (c.height[0].type, c.height[0].center, c.height[0].spread) == (abi.DIST_GAUSS, 1.0, pytest.approx(0.1))
    assert (c.face_dist[0].type, c.face_dist[0].center, c.face_dist[0].spread) == (abi.DIST_UNIFORM, 1.0, pytest.approx(0.2))
    assert (c.face_dist[1].type, c.face_dist[1].center) == (abi.DIST_NONE, pytest.approx(1.1))
    # pyramid heights are seeded {none, 0, 0} (crystal_config.hpp:69-71)
    c2, _ = config.parse_crystal({"id": 2, "type": "pyramid", "shape": {"prism_h": {"type": "gauss", "std": 0.1}, "upper_h": 0.2}})
    assert (c2.height[1].type, c2.height[1].center, c2.height[1].spread) == (abi.DIST_GAUSS, 0.0, pytest.approx(0.1))
    # a full object is unchanged by the defaults
    _, a3 = config.parse_crystal({"id": 3, "type": "prism", "shape": {}, "axis": {"zenith": {"type": "gauss", "mean": 20, "std": 5}}})
    assert (a3.latitude.center, a3.latitude.spread, a3.roll.type, a3.roll.spread) == (70.0, 5.0, abi.DIST_UNIFORM, 360.0)


def test_illuminant_and_rejections():
    doc = copy.deepcopy(DOC)
    doc["scene"]["light_source"]["spectrum"] = "D65"
    job = config.load_config(doc)
    assert len(job.wavelengths) == 1 and job.wavelengths[0].illuminant == abi.ILLUM["D65"]
    bad = copy.deepcopy(DOC)
    bad["crystal"][1]["axis"]["zenith"] = {"mean": 90, "std": 1}
    with pytest.raises(config.ConfigError, match="type"):
        config.load_config(bad)
    bad = copy.deepcopy(DOC)
    del bad["scene"]["scattering"][0]["prob"]
    with pytest.raises(config.ConfigError, match="prob"):
        config.load_config(bad)
    bad = copy.deepcopy(DOC)
    bad["scene"]["max_hits"] = 65
    with pytest.raises(config.ConfigError, match="max_hits"):
        config.load_config(bad)
    rc = copy.deepcopy(DOC)
    rc["raypath_color"] = [{"color": [1, 0, 0], "match": [{"layer": 0, "crystal": 99}]}]
    with pytest.raises(config.ConfigError, match="no scattering setting with crystal_id 99"):
        config.load_config(rc)


def test_filters_map_onto_the_filter_table():
    doc = copy.deepcopy(DOC)
    doc["filter"] = [
        {"id": 1, "type": "none"},
        {"id": 2, "type": "raypath", "raypath": [3, 1, 5, 7, 4], "symmetry": "PBD"},
        {"id": 4, "type": "entry_exit", "entry": 3, "exit": 5, "action": "filter_in"},
        {"id": 5, "type": "direction", "az": 180, "el": 25, "radii": 0.5, "action": "filter_out"},
        {"id": 6, "type": "crystal", "crystal_id": 3},
        {"id": 7, "type": "complex", "composition": [1, [2, 6], 5]},
    ]
    doc["scene"]["scattering"][0]["entries"] = [{"crystal": 3, "filter": 7}, {"crystal": 5, "filter": 2}, {"crystal": 1}]
    job = config.load_config(doc)
    ents = job.scene.layers[0].entries
    assert [ents[i].filter_id for i in range(3)] == [1, 2, 0] and len(job.filters) == 2
    f7, f2 = job.filters
    assert f7.is_complex == 1 and f7.or_count == 3 and list(f7.and_counts)[:3] == [1, 2, 1]
    assert [f7.terms[k].type for k in range(4)] == [abi.FILTER_NONE, abi.FILTER_RAYPATH, abi.FILTER_CRYSTAL, abi.FILTER_DIRECTION]
    assert f7.terms[3].radii == 0.5 and f7.terms[2].crystal_id == 3
    assert f2.is_complex == 0 and f2.symmetry == 7 and list(f2.terms[0].raypath)[:5] == [3, 1, 5, 7, 4] and f2.terms[0].raypath_len == 5


@pytest.mark.skipif(not os.path.isdir("/root/reference/examples"), reason="reference examples only exist in the build container")
def test_reference_example_files_parse():
    for name in ("bench_config.json", "bench_config_stoch.json", "lens_orthographic.json"):
        path = os.path.join("/root/reference/examples", name)
        if os.path.exists(path):
            job = config.load_config(path)
            assert job.scene.layer_count >= 1 and job.renders
    if os.path.exists("/root/reference/examples/config_example.json"):
        job = config.load_config("/root/reference/examples/config_example.json")
        assert job.ray_num == 450_000_000 and len(job.wavelengths) == 9 and sorted(job.renders) == [1, 2, 3, 4]


with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_e2e_configs.json")) as _f:
    E2E_DOCS = json.load(_f)       # tests/golden/make_e2e_config_bundle.py: the reference's test/e2e/configs/*.json
E2E_CONFIGS = sorted(E2E_DOCS)


@pytest.mark.parametrize("name", E2E_CONFIGS)
def test_reference_e2e_config_documents_parse(name):
    """Every config document of the reference's end-to-end tests that is kept as a fixture maps onto the ABI structs: symmetry
    spellings ("none", "PBD"), 12-clause OR filters, raypath_color tables, several renderers, pyramid Miller indices ..."""
    job = config.load_config(E2E_DOCS[name])
    assert job.scene.layer_count >= 1 and job.renders and job.wavelengths
    for l in range(job.scene.layer_count):
        for e in range(job.scene.layers[l].entry_count):
            ent = job.scene.layers[l].entries[e]
            assert 0 <= ent.filter_id <= len(job.filters) and 0 <= ent.color_id <= len(job.color_sets)
    if name == "parity_big_or_with_color":
        big = [f for f in job.filters if f.is_complex]
        assert len(big) == 1 and big[0].or_count == 12 and len(job.color_classes) == 3
    if name in ("raypath_symmetry_4_6", "ms_filter_leak_impossible"):
        assert job.filters[0].symmetry == 0          # "symmetry": "none"


@pytest.mark.gpu
def test_cli_benchmark_line(tmp_path):
    import json
    import subprocess
    import sys
    doc = copy.deepcopy(DOC)
    doc["scene"]["ray_num"] = 9_000_000
    cfg = tmp_path / "cfg.json"
    cfg.write_text(json.dumps(doc))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "ice_halo_sim_amd.cli", "-f", str(cfg), "--render", "4", "--benchmark",
                        "--out-rgb", str(tmp_path / "o.ppm")], capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("[BENCHMARK] ")][0]
    rec = json.loads(line[len("[BENCHMARK] "):])
    assert rec["rays"] == 9_000_000 and rec["rays_per_sec"] > 1e7 and rec["rate_basis"] in ("steady", "active_short")
    assert (tmp_path / "o.ppm").stat().st_size > 1920 * 1080 * 3


def test_raypath_color_maps_onto_color_sets_and_classes():
    """raypath_color (doc shape of test/e2e/configs/raypath_color_multi_layer.json): bits are assigned per unique
    (layer, crystal, predicate, symmetry) in class/match order; class bits OR their refs; placements must be unambiguous."""
    import json
    doc = copy.deepcopy(DOC)
    doc.pop("filter")
    doc["crystal"] = [{"id": i, "type": "prism", "shape": {"height": 1.0 + 0.1 * i}} for i in (1, 2, 3, 4, 5)]
    doc["scene"]["scattering"] = [{"prob": 0.9, "entries": [{"crystal": 1, "proportion": 30}, {"crystal": 4, "proportion": 10}, {"crystal": 5, "proportion": 10}]},
                                  {"prob": 0.0, "entries": [{"crystal": 2, "proportion": 10}, {"crystal": 3, "proportion": 10}]}]
    doc["raypath_color"] = {"mode": "dominant", "classes": [
        {"color": [1.0, 0.0, 1.0], "combine": "all", "match": [{"layer": 0, "crystal": 1}, {"layer": 1, "crystal": 2}]},
        {"color": [1.0, 0.55, 0.0], "match": [{"layer": 1, "crystal": 3}]},
        {"color": [1.0, 0.0, 0.0], "match": [{"layer": 0, "crystal": 4}]},
        {"color": [0.0, 1.0, 0.0], "combine": "any", "match": [{"layer": 0, "crystal": 5, "type": "entry_exit", "min_len": 2, "max_len": 2},
                                                                 {"layer": 0, "crystal": 5, "type": "entry_exit", "min_len": 3},
                                                                 {"layer": 0, "crystal": 5, "type": "entry_exit", "min_len": 2, "max_len": 2}]},
        {"color": [0.0, 0.0, 1.0], "match": [{"layer": 0, "crystal": 1, "type": "raypath", "raypath": [3, 5], "symmetry": "PBD"},
                                             {"layer": 0, "crystal": 1}]}]}
    job = config.load_config(doc)
    assert job.color_mode == "dominant" and len(job.color_classes) == 5 and len(job.color_sets) == 5
    bits = [c.bits for c in job.color_classes]
    assert bits == [0b11, 0b100, 0b1000, 0b110000, 0b1000001]           # duplicate refs reuse their bit; new predicate gets bit 6
    assert [c.combine_all for c in job.color_classes] == [1, 0, 0, 0, 0]
    ids = [[e.color_id for e in job.scene.layers[l].entries[:job.scene.layers[l].entry_count]] for l in (0, 1)]
    assert ids == [[1, 2, 3], [4, 5]]
    s1 = job.color_sets[0]                                               # crystal 1 on layer 0: the `none` predicate and the symmetric raypath
    assert s1.term_count == 2 and [s1.terms[k].bit for k in range(2)] == [0, 6]
    assert s1.terms[0].predicate.type == abi.FILTER_NONE and s1.terms[1].predicate.type == abi.FILTER_RAYPATH and s1.terms[1].symmetry == 7
    s5 = job.color_sets[2]
    assert s5.term_count == 2 and s5.terms[0].predicate.max_len == 2 and s5.terms[1].predicate.min_len == 3
    # ambiguity and range errors follow the reference's messages
    bad = json.loads(json.dumps(doc))
    bad["scene"]["scattering"][0]["entries"].append({"crystal": 1, "proportion": 1})
    with pytest.raises(config.ConfigError, match="matches 2 scattering settings"):
        config.load_config(bad)
    bad = json.loads(json.dumps(doc))
    bad["raypath_color"]["classes"][0]["match"][0]["layer"] = 7
    with pytest.raises(config.ConfigError, match="layer index out of range"):
        config.load_config(bad)
    bad = json.loads(json.dumps(doc))
    bad["raypath_color"]["classes"][1]["combine"] = "xor"
    with pytest.raises(config.ConfigError, match="unknown combine"):
        config.load_config(bad)


@pytest.mark.gpu
def test_cli_output_directory_contract_of_the_reference(tmp_path):
    """The reference's e2e harness drives its binary as `-f <config> -o <dir> [--format png]` and looks for img_<id>.<fmt> per render entry
    and img_<id>_components.<fmt> with raypath_color (main.cpp:244-315; test_smoke.py, test_raypath_color_painter_default.py::
    test_bare_array_renders_default_painter: both files exist, non-empty, at the configured resolution)."""
    import json
    import subprocess
    import sys
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(name, *extra):
        cfg = tmp_path / (name + ".json")
        cfg.write_text(json.dumps(E2E_DOCS[name]))
        out = tmp_path / ("out_" + name)
        out.mkdir()                                               # like the reference, the CLI does not create the output directory
        r = subprocess.run([sys.executable, "-m", "ice_halo_sim_amd.cli", "-f", str(cfg), "-o", str(out)] + list(extra), capture_output=True, text=True, cwd=root)
        assert r.returncode == 0, r.stdout + r.stderr
        return out, r.stdout

    out, log = run("multi_lens")                                  # three render entries, default format jpg
    for rid in (1, 2, 3):
        f = out / ("img_%02d.jpg" % rid)
        assert f.exists() and f.stat().st_size > 0 and ("Saved: %s" % f) in log
        assert Image.open(f).size == tuple(E2E_DOCS["multi_lens"]["render"][rid - 1]["resolution"])
    assert not list(out.glob("*_components.*"))                   # no raypath_color section: the mono path alone
    out, _ = run("painter_default_overlap", "--format", "png", "--backend", "cpu")
    res = tuple(E2E_DOCS["painter_default_overlap"]["render"][0]["resolution"])
    for name in ("img_01.png", "img_01_components.png"):
        assert (out / name).exists() and (out / name).stat().st_size > 0 and Image.open(out / name).size == res


# ---- what the reference's CLI tests say about rejected and warned-about documents (test/e2e-correctness/test_cli.py:145-345) -------------
def test_config_errors_carry_the_references_migration_guidance():
    """Derived from a real reference document (halo_22.json), each differing from the working one in exactly one deleted key, like the
    reference's tests: a missing `prob` names the layer, the field and the write-up that keeps the old behaviour; an axis slot without
    `type` names the crystal, the slot and both legal spellings; an `axis` without `zenith` is rejected."""
    from ice_halo_sim_amd import config
    base = E2E_DOCS["halo_22"]
    assert "prob" in base["scene"]["scattering"][0] and "type" in base["crystal"][0]["axis"]["zenith"] and "type" in base["crystal"][0]["axis"]["azimuth"]
    config.load_config(copy.deepcopy(base))
    doc = copy.deepcopy(base)
    del doc["scene"]["scattering"][0]["prob"]
    with pytest.raises(config.ConfigError) as e:
        config.load_config(doc)
    for part in ("scattering[0]", "prob", '"prob": 0'):          # TestScatteringProbRequired.test_error_message_is_actionable
        assert part in str(e.value)
    for slot in ("zenith", "azimuth"):                            # TestAxisSlotTypeRequired.test_cli_rejects_axis_slot_without_type
        doc = copy.deepcopy(base)
        del doc["crystal"][0]["axis"][slot]["type"]
        with pytest.raises(config.ConfigError) as e:
            config.load_config(doc)
        for part in ("crystal[id=1]", "axis." + slot, '"%s": 20' % slot, '"type": "gauss"'):   # ...test_error_message_is_actionable
            assert part in str(e.value), (part, str(e.value))
    doc = copy.deepcopy(base)
    del doc["crystal"][0]["axis"]["zenith"]
    with pytest.raises(config.ConfigError) as e:                  # test_cli_rejects_axis_without_zenith / test_missing_zenith_message_is_actionable
        config.load_config(doc)
    assert "crystal[id=1]" in str(e.value) and "zenith" in str(e.value)
    doc = copy.deepcopy(base)
    doc["crystal"][0]["shape"]["height"] = {"mean": 1.2, "std": 0.1}            # the backstop in from_json(Distribution&): shape scalars too
    with pytest.raises(config.ConfigError) as e:
        config.load_config(doc)
    assert '"type"' in str(e.value)


def test_last_layer_prob_warning_like_the_reference_cli():
    """TestLastLayerProbWarning: parity_single_ms_bd_filter.json has last-layer prob 0.5 (warned), halo_22.json 0.0 (silent)."""
    from ice_halo_sim_amd import config
    warned = config.load_config(E2E_DOCS["parity_single_ms_bd_filter"]).warnings
    assert len(warned) == 1 and warned[0].startswith("Last scattering layer has prob=0.5000")
    assert config.load_config(E2E_DOCS["halo_22"]).warnings == []


def test_cli_fails_gracefully_like_the_reference(tmp_path):
    """test/regression-sentinel/test_errors.py: a nonexistent config, invalid JSON, documents missing `scene` / `render` / `crystal` (the
    reference's own error fixtures, restated — each a few lines), a nonexistent output directory, an unknown option: non-zero exit, a
    message, no traceback.  None of these reaches the GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scene = {"light_source": {"type": "sun", "altitude": 20.0, "spectrum": "E"}, "ray_num": 1000, "max_hits": 3,
             "scattering": [{"prob": 0.0, "entries": [{"crystal": 1, "proportion": 1}]}]}
    crystal = [{"id": 1, "type": "prism", "shape": {"height": 1.2}}]
    render = [{"id": 1, "lens": {"type": "linear", "fov": 80}, "resolution": [64, 64]}]
    docs = {"missing_crystal": {"filter": [], "scene": scene, "render": render},
            "missing_render": {"crystal": crystal, "filter": [], "scene": scene},
            "missing_scene": {"crystal": crystal, "filter": [], "render": render}}
    cases = [(["-f", "nonexistent_file_that_does_not_exist.json"], "nonexistent config"), (["-z"], "unknown option")]
    bad = tmp_path / "invalid_json.json"
    bad.write_text("{bad json syntax here\n")
    cases.append((["-f", str(bad)], "invalid JSON"))
    for name, doc in docs.items():
        f = tmp_path / (name + ".json")
        f.write_text(json.dumps(doc))
        cases.append((["-f", str(f), "-o", str(tmp_path)], name))
    ok = tmp_path / "halo_22.json"
    ok.write_text(json.dumps(E2E_DOCS["halo_22"]))
    cases.append((["-f", str(ok), "-o", "/nonexistent/path/that/does/not/exist"], "nonexistent output dir"))
    cases.append((["-f", str(ok), "-o", str(tmp_path), "--format", "bmp"], "invalid format"))        # test_cli.py:113-143
    cases.append((["-f", str(ok), "-o", str(tmp_path), "--quality", "0"], "quality out of range"))
    cases.append((["-f", str(ok), "-o", str(tmp_path), "--quality", "abc"], "quality not a number"))
    cases.append((["-f", str(ok), "-o", str(tmp_path), "--quality"], "quality without a value"))
    for args, what in cases:
        r = subprocess.run([sys.executable, "-m", "ice_halo_sim_amd.cli"] + args, capture_output=True, text=True, cwd=root)
        assert r.returncode > 0, what
        assert "Traceback" not in r.stderr, (what, r.stderr)
        assert (r.stderr + r.stdout).strip(), what
    assert not list(tmp_path.glob("img_*"))
    r = subprocess.run([sys.executable, "-m", "ice_halo_sim_amd.cli", "-h"], capture_output=True, text=True, cwd=root)   # test_help_flag
    assert r.returncode == 0 and "usage:" in r.stdout.lower()
    r = subprocess.run([sys.executable, "-m", "ice_halo_sim_amd.cli"], capture_output=True, text=True, cwd=root)         # test_no_args
    assert r.returncode != 0
