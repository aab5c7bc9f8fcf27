"""Reader for Lumice's JSON configuration surface, restricted to what the trace path consumes.

Restates the reference parsers (paths under /root/reference/src):
  config/config_manager.cpp:96-231  ParseSceneConfig / from_json(ConfigManager)   scene, scattering, render lookup
  config/crystal_config.cpp:282-433 prism / pyramid shape (+ sync_group, Miller indices → wedge angle)
  core/math.cpp:593-726             Distribution and axis (zenith → latitude, azimuth/roll default uniform 360)
  config/light_config.cpp:44-79     sun + spectrum (wavelength list | illuminant name)
  config/render_config.cpp:55-118   lens (fov | focal length f), config_manager.cpp:36-80 render entry
  server/ray_num_semantics.hpp      per-wavelength ray count = ceil(ray_num / N_wavelengths)
Host-side plumbing only: it fills the C-ABI structs; nothing is computed here.
"""
import json
import math

from . import abi, scenes

LENS_NAMES = {
    "linear": abi.LENS_LINEAR, "fisheye_equal_area": abi.LENS_FISHEYE_EQUAL_AREA,
    "fisheye_equidistant": abi.LENS_FISHEYE_EQUIDISTANT, "fisheye_stereographic": abi.LENS_FISHEYE_STEREOGRAPHIC,
    "dual_fisheye_equal_area": abi.LENS_DUAL_FISHEYE_EQUAL_AREA, "dual_fisheye_equidistant": abi.LENS_DUAL_FISHEYE_EQUIDISTANT,
    "dual_fisheye_stereographic": abi.LENS_DUAL_FISHEYE_STEREOGRAPHIC, "rectangular": abi.LENS_RECTANGULAR,
    "fisheye_orthographic": abi.LENS_FISHEYE_ORTHOGRAPHIC, "dual_fisheye_orthographic": abi.LENS_DUAL_FISHEYE_ORTHOGRAPHIC,
    "globe": abi.LENS_GLOBE,
}
VISIBLE_NAMES = {"upper": abi.VISIBLE_UPPER, "lower": abi.VISIBLE_LOWER, "full": abi.VISIBLE_FULL}
MAX_HITS = 64  # kMaxHits, core/def.hpp:27 (enforced at parse time, config_manager.cpp:144-146)


class ConfigError(ValueError):
    pass


class UnsupportedConfig(ConfigError):
    """The configuration is valid Lumice JSON but asks for something this backend does not implement; the reference
    would answer `IsCompatible() == false` / BackendUnavailableError and fall back to its legacy CPU path."""


def _dist(obj, default=0.0, default_spread=0.0):
    """from_json(Distribution&) (math.cpp:593-630): `default` / `default_spread` are the destination slot's seeded values,
    which an object keeps for the keys it does not carry."""
    if isinstance(obj, bool):
        raise ConfigError("distribution value is neither a number nor an object: %r" % (obj,))
    if isinstance(obj, (int, float)):
        return abi.dist(float(obj))
    if isinstance(obj, dict):
        if "type" not in obj:
            raise ConfigError('distribution object is missing required key "type". Write either a bare number (e.g. 20) or an object naming the '
                              'distribution (e.g. {"type": "gauss", "mean": 20, "std": 5}).')   # math.cpp:611-615
        try:
            return abi.dist(obj, default, default_spread)
        except KeyError:
            raise ConfigError("unknown distribution type: %r" % (obj["type"],))
    raise ConfigError("distribution value is neither a number nor an object: %r" % (obj,))


def _canonical_sync_groups(groups, applicable):
    """CanonicalizeSyncGroups (crystal_config.cpp:48-98): drop groups on absent slots and one-member groups, renumber
    1..N by first appearance."""
    g = [groups[i] if applicable[i] else 0 for i in range(len(groups))]
    g = [v if (v and g.count(v) >= 2) else 0 for v in g]
    remap, out = {}, []
    for v in g:
        if v and v not in remap:
            remap[v] = len(remap) + 1
        out.append(remap.get(v, 0))
    return out


_atanf = None


def _miller_to_alpha(i1, i4):
    """MillerToAlpha, crystal_config.cpp:328-340 (i1 == 0 → default 28 degrees) — in FLOAT like the reference: its constants are float
    literals, the products and quotients round to float one by one and the arctangent is libm's atanf.  (Evaluated in double and rounded at
    the end, the (2,0,3) wedge came out one ulp above the reference's — found by tests/test_glue_mapping.py.)"""
    global _atanf
    if i1 == 0:
        return 28.0
    import ctypes
    import numpy as np
    f = np.float32
    x = f(f(f(f(0.866025403784) * f(i4)) / f(i1)) / f(1.629))
    if _atanf is None:
        try:
            _atanf = ctypes.CDLL("libm.so.6").atanf
            _atanf.restype, _atanf.argtypes = ctypes.c_float, [ctypes.c_float]
        except OSError:
            _atanf = lambda v: float(np.arctan(f(v)))
    return float(f(f(_atanf(float(x))) * f(57.2957795131)))


def parse_crystal(j):
    """CrystalConfig from_json (crystal_config.cpp:414-431). Returns (HaloCrystal, HaloAxis)."""
    shape = j.get("shape", {})
    kind = j["type"]
    fd = [abi.dist(1.0)] * 6
    if "face_distance" in shape:
        vals = list(shape["face_distance"])[:6]
        fd = [_dist(v, 1.0) for v in vals] + [abi.dist(1.0)] * (6 - len(vals))  # slots seeded {none, 1.0} (crystal_config.cpp:310-313)
    sg = shape.get("sync_group", {})
    face_groups = list(sg.get("face_distance", [0] * 6))[:6]
    face_groups += [0] * (6 - len(face_groups))
    c = abi.HaloCrystal()
    if kind == "prism":
        c.kind = abi.CRYSTAL_PRISM
        heights = [_dist(shape["height"], 1.0) if "height" in shape else abi.dist(1.0), abi.dist(0.0), abi.dist(0.0)]  # h_ seeded {none, 1.0, 0} (crystal_config.hpp:63)
        groups = _canonical_sync_groups([int(sg.get("height", 0)), 0, 0] + [int(v) for v in face_groups],
                                        [True, False, False] + [True] * 6)
        c.wedge_upper_deg = c.wedge_lower_deg = 28.0
    elif kind == "pyramid":
        c.kind = abi.CRYSTAL_PYRAMID
        if "prism_h" not in shape:
            raise ConfigError('pyramid shape is missing required key "prism_h"')
        heights = [_dist(shape.get("upper_h", 0.0)), _dist(shape["prism_h"]), _dist(shape.get("lower_h", 0.0))]
        groups = _canonical_sync_groups([int(sg.get("upper_h", 0)), int(sg.get("prism_h", 0)), int(sg.get("lower_h", 0))] +
                                        [int(v) for v in face_groups], [True] * 9)
        wedge = {}
        for side in ("upper", "lower"):
            wedge[side] = 28.0  # PyramidCrystalParam defaults, crystal_config.hpp:74-75
            if side + "_wedge_angle" in shape:
                wedge[side] = float(shape[side + "_wedge_angle"])
            elif isinstance(shape.get(side + "_indices"), list) and len(shape[side + "_indices"]) == 3:
                idx = shape[side + "_indices"]
                wedge[side] = _miller_to_alpha(int(idx[0]), int(idx[2]))
        c.wedge_upper_deg, c.wedge_lower_deg = wedge["upper"], wedge["lower"]
    else:
        raise ConfigError('unknown crystal type: %r. Write either "prism" or "pyramid".' % (kind,))
    # NormalizeSyncGroups (crystal_config.cpp:102-133): members take their group leader's distribution
    slots = heights + fd
    for i in range(9):
        if groups[i]:
            leader = next(k for k in range(9) if groups[k] == groups[i])
            slots[i] = slots[leader]
    for i in range(3):
        c.height[i] = slots[i]
    for i in range(6):
        c.face_dist[i] = slots[3 + i]
    for i in range(9):
        c.sync_group[i] = groups[i]
    ax = j.get("axis")
    if ax is None:
        axis = scenes.axis()
    else:
        hint = lambda key: ('Write axis.%s either as a bare number for a fixed angle (e.g. "%s": 20) or as an object naming the distribution '
                            '(e.g. "%s": {"type": "gauss", "mean": 20, "std": 5}).' % (key, key, key))   # FormatAxisSlotHint, math.cpp:651-655
        if "zenith" not in ax:   # from_json(AxisDistribution&), math.cpp:693-704
            raise ConfigError('axis is present but has no "zenith", which is required whenever `axis` is written at all (omit `axis` entirely to get '
                              'the default orientation instead). ' + hint("zenith"))
        for key in ("zenith", "azimuth", "roll"):
            if isinstance(ax.get(key), dict) and "type" not in ax[key]:   # ParseAxisSlot, math.cpp:668-675
                raise ConfigError('axis.%s is a distribution object with no "type". ' % key + hint(key))
        axis = scenes.axis(zenith=ax["zenith"], azimuth=ax.get("azimuth"), roll=ax.get("roll"))
    return c, axis


def parse_lens(j):
    """LensParam from_json (render_config.cpp:55-118): `fov` in degrees or focal length `f` on 35 mm film."""
    if j["type"] not in LENS_NAMES:
        raise ConfigError("unknown lens type: %r" % (j["type"],))
    t = LENS_NAMES[j["type"]]
    if "fov" in j:
        fov = float(j["fov"])
    elif "f" in j:
        f, d = float(j["f"]), 12.0
        if t in (abi.LENS_LINEAR, abi.LENS_GLOBE):
            fov = math.degrees(math.atan2(d, f) * 2)
        elif t in (abi.LENS_FISHEYE_EQUAL_AREA, abi.LENS_DUAL_FISHEYE_EQUAL_AREA):
            if d / (2 * f) > 1.0:
                raise ConfigError("focal length too short for equal area fisheye (f >= 6mm required)")
            fov = math.degrees(math.asin(d / (2 * f)) * 4)
        elif t in (abi.LENS_FISHEYE_EQUIDISTANT, abi.LENS_DUAL_FISHEYE_EQUIDISTANT):
            fov = math.degrees(d / f)
        elif t in (abi.LENS_FISHEYE_STEREOGRAPHIC, abi.LENS_DUAL_FISHEYE_STEREOGRAPHIC):
            fov = math.degrees(math.atan(d / (2 * f)) * 4)
        elif t == abi.LENS_RECTANGULAR:
            fov = 0.0
        else:
            if d / f > 1.0:
                raise ConfigError("focal length too short for orthographic fisheye")
            fov = math.degrees(math.asin(d / f) * 2)
    else:
        raise ConfigError("missing key [fov] or [f]")
    max_fov = {abi.LENS_LINEAR: 179.0, abi.LENS_FISHEYE_STEREOGRAPHIC: 359.0, abi.LENS_FISHEYE_ORTHOGRAPHIC: 180.0,
               abi.LENS_DUAL_FISHEYE_ORTHOGRAPHIC: 180.0, abi.LENS_GLOBE: 90.0}.get(t, 360.0)
    if t != abi.LENS_RECTANGULAR and not (0 < fov <= max_fov):
        raise ConfigError("fov must be in (0, %d] degrees for this lens type" % int(max_fov))
    return t, fov


def parse_render(j):
    """ParseRenderConfig (config_manager.cpp:36-80); defaults RenderConfig render_config.hpp:71-103."""
    t, fov = (abi.LENS_LINEAR, 90.0)
    if "lens" in j:
        t, fov = parse_lens(j["lens"])
    res = j["resolution"]
    view = j.get("view", {})
    vis = j.get("visible", "upper")
    if vis not in VISIBLE_NAMES:
        raise ConfigError("unknown visible range: %r" % (vis,))
    return scenes.render(t, int(res[0]), int(res[1]), fov=fov, az=float(view.get("azimuth", 0.0)),
                         el=float(view.get("elevation", 0.0)), ro=float(view.get("roll", 0.0)), visible=VISIBLE_NAMES[vis],
                         overlap=max(0.0, float(j.get("overlap", 0.0))), lens_shift=tuple(j.get("lens_shift", (0, 0))))


def _parse_simple_filter(j):
    """SimpleFilterParam from_json (filter_config.cpp:62-118)."""
    t = j.get("type", "none")
    if t == "none":
        return scenes.filter_term("none")
    if t == "raypath":
        return scenes.filter_term("raypath", raypath=[int(v) for v in j["raypath"]])
    if t == "entry_exit":
        mn = int(j["min_len"]) if j.get("min_len") is not None else 1
        mx = int(j["max_len"]) if j.get("max_len") is not None else None
        if mn < 1:
            raise ConfigError("entry_exit filter: min_len must be >= 1, got %d" % mn)
        if mx is not None and (mx < mn or mx > MAX_HITS):
            raise ConfigError("entry_exit filter: max_len (%d) must be in [min_len, %d]" % (mx, MAX_HITS))
        return scenes.filter_term("entry_exit", entry=j.get("entry"), exit=j.get("exit"), min_len=mn, max_len=mx)
    if t == "direction":
        return scenes.filter_term("direction", az=float(j["az"]), el=float(j["el"]), radii=float(j["radii"]))
    if t == "crystal":
        return scenes.filter_term("crystal", crystal_id=int(j["crystal_id"]))
    raise ConfigError("SimpleFilterParam: unknown type %r" % (t,))


def parse_filters(jfilters):
    """from_json(ConfigManager) filter passes (config_manager.cpp:184-215): simple filters first, then complex ones, whose
    `composition` is an OR-list of filter ids or AND-lists of ids.  Returns {id: HaloFilter}."""
    simple, out = {}, {}
    for jf in jfilters:
        if jf.get("type") == "complex":
            continue
        term = _parse_simple_filter(jf)
        simple[int(jf["id"])] = term
        out[int(jf["id"])] = scenes.simple_filter(term, jf.get("symmetry", ""), jf.get("action", "filter_in"))
    for jf in jfilters:
        if jf.get("type") != "complex":
            continue
        clauses = []
        for c in jf["composition"]:
            ids = c if isinstance(c, list) else [c]
            for fid in ids:
                if int(fid) not in simple:
                    raise ConfigError("complex filter %s refers to unknown simple filter id %s" % (jf.get("id"), fid))
            clauses.append([simple[int(fid)] for fid in ids])
        if len(clauses) > abi.FILTER_MAX_OR or sum(len(c) for c in clauses) > abi.FILTER_MAX_TERMS:
            raise UnsupportedConfig("complex filter %s exceeds the backend's clause caps" % jf.get("id"))
        out[int(jf["id"])] = scenes.complex_filter(clauses, jf.get("symmetry", ""), jf.get("action", "filter_in"))
    return out


class TraceJob:
    """Everything the trace path needs from one Lumice config document."""

    def __init__(self):
        self.scene = None
        self.renders = {}          # id -> HaloRender
        self.render_meta = {}      # id -> {"intensity_factor", "ray_color", "background", "opacity"}
        self.wavelengths = []      # list of HaloWl (one per discrete wavelength, or a single illuminant entry)
        self.ray_num = 0           # total root rays requested (None = "infinite")
        self.geom_clock = None
        self.filters = []          # HaloFilter table; HaloEntry.filter_id is a 1-based index into it
        self.color_sets, self.color_classes, self.color_meta, self.color_mode = [], [], [], "painter"   # raypath_color
        self.warnings = []         # what the reference CLI logs as warnings about a document it still runs

    def per_wavelength_ray_num(self):
        """ceil(ray_num / N_wl) — server/ray_num_semantics.hpp:13-17."""
        n = max(1, len(self.wavelengths))
        return None if self.ray_num is None else -(-int(self.ray_num) // n)


def load_config(source):
    """source: path, JSON text or dict → TraceJob.  Mirrors from_json(ConfigManager) (config_manager.cpp:170-231)."""
    if isinstance(source, dict):
        doc = source
    else:
        text = source
        if not text.lstrip().startswith("{"):
            try:
                with open(source) as f:
                    text = f.read()
            except OSError as e:
                raise ConfigError("cannot read config file %s: %s" % (source, e.strerror or e))
        try:
            doc = json.loads(text)
        except ValueError as e:
            raise ConfigError("config is not valid JSON: %s" % e)
    if not isinstance(doc, dict):
        raise ConfigError("config root must be a JSON object")
    for key in ("crystal", "scene", "render"):   # from_json(ConfigManager) reads all three with .at() (config_manager.cpp:170-231): a document without one fails
        if key not in doc:
            raise ConfigError('config is missing required section "%s"' % key)
    try:
        return _load_document(doc)
    except (KeyError, TypeError, IndexError) as e:   # a required key the sections' own parsers read with .at()
        raise ConfigError("config is missing or mistypes a required field: %s: %s" % (type(e).__name__, e))


def _load_document(doc):
    crystals = {}
    for jc in doc["crystal"]:
        try:
            crystals[int(jc["id"])] = parse_crystal(jc)
        except ConfigError as e:
            raise ConfigError("crystal[id=%s]: %s" % (jc.get("id"), e))
    filters = parse_filters(doc.get("filter", []))
    job = TraceJob()
    filter_slot = {}
    for jr in doc.get("render", []):
        r = parse_render(jr)
        job.renders[int(jr["id"])] = r
        # the appearance fields PostSnapshot reads (config_manager.cpp:62-73, render_config.hpp:84-92); opacity and the grids belong
        # to the GUI's overlay pass and are carried, not drawn
        job.render_meta[int(jr["id"])] = {"intensity_factor": float(jr.get("intensity_factor", 1.0)),
                                          "ray_color": [float(v) for v in jr.get("ray_color", (-1.0, -1.0, -1.0))],
                                          "background": [float(v) for v in jr.get("background", (0.0, 0.0, 0.0))],
                                          "opacity": float(jr.get("opacity", 1.0))}
    js = doc["scene"]
    rn = js["ray_num"]
    job.ray_num = None if rn == "infinite" else int(rn)
    max_hits = int(js["max_hits"])
    if max_hits == 0 or max_hits > MAX_HITS:
        raise ConfigError("max_hits must be in [1, %d]" % MAX_HITS)
    if "geom_clock" in js:
        job.geom_clock = int(js["geom_clock"])
    ls = js["light_source"]
    if ls.get("type") != "sun":
        raise ConfigError("unknown light source type: %r" % (ls.get("type"),))
    spec = ls["spectrum"]
    if isinstance(spec, str):
        if spec not in abi.ILLUM:
            raise ConfigError("unknown illuminant: %r" % (spec,))
        job.wavelengths = [scenes.wl_illuminant(spec, 64)]
    elif isinstance(spec, list):
        job.wavelengths = [scenes.wl_discrete(float(w["wavelength"]), float(w["weight"])) for w in spec]
    else:
        raise ConfigError("light_source.spectrum is neither a string nor an array")
    layers = []
    for li, jl in enumerate(js["scattering"]):
        if "prob" not in jl:
            raise ConfigError('scene.scattering[%d] is missing required field "prob" (multi-scattering probability). The historical default '
                              'was 0.0; add "prob": 0 explicitly to keep that behavior.' % li)   # config_manager.cpp:108-112: the text IS the migration guidance
        entries = []
        for je in jl["entries"]:
            cid = int(je["crystal"])
            if cid not in crystals:
                raise ConfigError("scattering entry refers to unknown crystal id %d" % cid)
            slot = 0
            if "filter" in je:
                fid = int(je["filter"])
                if fid not in filters:
                    raise ConfigError("scattering entry refers to unknown filter id %d" % fid)
                if fid not in filter_slot:
                    job.filters.append(filters[fid])
                    filter_slot[fid] = len(job.filters)
                slot = filter_slot[fid]
            crystal, axis = crystals[cid]
            entries.append(scenes.entry(crystal, axis, float(je.get("proportion", 100.0)), cid, slot))
        layers.append((float(jl["prob"]), entries))
    if len(layers) > abi.MAX_LAYERS or any(len(e) > abi.MAX_ENTRIES for _, e in layers):
        raise UnsupportedConfig("more scattering layers / entries than the backend's caps")
    job.warnings = []
    if layers and layers[-1][0] > 0.0:   # WarnOnLastLayerProb, main.cpp:60-90
        job.warnings.append("Last scattering layer has prob=%.4f > 0: that fraction of filter-pass rays will be discarded (no next layer to receive "
                            "them). Set the last layer's prob to 0 unless this is intentional." % layers[-1][0])
    job.scene = scenes.scene(layers, max_hits=max_hits, sun_altitude=float(ls["altitude"]),
                             sun_azimuth=float(ls.get("azimuth", 0.0)), sun_diameter=float(ls.get("diameter", 0.0)))
    if "raypath_color" in doc and doc["raypath_color"]:
        parse_raypath_color(doc["raypath_color"], layers, job)
    return job


def parse_raypath_color(jrc, layers, job):
    """raypath_color → the backend's colour sets + classes.  Restates from_json(RaypathColorConfig)
    (raypath_color_config.cpp:84-101), BuildColorGateTable (color_gate_table.cpp:57-103: one bit per unique
    (layer, crystal_id, predicate, symmetry) in class/match order, placement must be unambiguous) and BuildColorClassTable
    (color_class_table.cpp:34-87: class bits = OR of its refs' bits, combine any|all).  Class colours / visibility / the
    composite mode are display-side and kept on the job for a compositor."""
    jclasses = jrc["classes"] if isinstance(jrc, dict) else jrc
    job.color_mode = jrc.get("mode", "painter") if isinstance(jrc, dict) else "painter"
    gate = []          # [(layer, crystal_id, key, term, symmetry, bit)]
    job.color_meta = []
    classes = []
    for jc in jclasses:
        combine = jc.get("combine", "any")
        if combine not in ("any", "all"):
            raise ConfigError('raypath_color: unknown combine "%s" (expected "any" or "all")' % combine)
        bits = []
        for ref in jc["match"]:
            layer, cid = int(ref["layer"]), int(ref["crystal"])
            if layer >= len(layers):
                raise ConfigError("raypath_color: layer index out of range (layer=%d, crystal_id=%d)" % (layer, cid))
            n_match = sum(1 for e in layers[layer][1] if e.crystal_config_id == cid)
            if n_match == 0:
                raise ConfigError("raypath_color: no scattering setting with crystal_id %d on layer %d" % (cid, layer))
            if n_match >= 2:
                raise ConfigError("raypath_color: crystal_id %d matches %d scattering settings on layer %d" % (cid, n_match, layer))
            pred = {k: v for k, v in ref.items() if k not in ("layer", "crystal", "symmetry")}
            sym = "".join(sorted(ref.get("symmetry", "")))
            key = (layer, cid, json.dumps(pred, sort_keys=True), sym)
            found = next((g for g in gate if g[:2] + (g[2],) + (g[4],) == key[:3] + (sym,)), None)
            if found is None:
                bit = len(gate) if len(gate) < 64 else None      # ComponentTable::kMaxBits; overflow = kNoBit
                found = (layer, cid, key[2], _parse_simple_filter(pred), sym, bit)
                gate.append(found)
            if found[5] is not None:
                bits.append(found[5])
        classes.append(scenes.color_class(bits, combine))
        job.color_meta.append({"color": [float(v) for v in jc["color"]], "visible": bool(jc.get("visible", True)),
                               "solo": bool(jc.get("solo", False))})
    if len(classes) > abi.COLOR_MAX_CLASSES:
        raise UnsupportedConfig("more raypath_color classes than the backend's cap")
    job.color_classes = classes
    job.color_sets = []
    for li, (_, entries) in enumerate(layers):
        for e in entries:
            terms = [(g[3], g[4], g[5]) for g in gate if g[0] == li and g[1] == e.crystal_config_id and g[5] is not None]
            if not terms:
                continue
            if len(terms) > abi.COLOR_MAX_TERMS:
                raise UnsupportedConfig("more raypath_color predicates on one placement than the backend's cap")
            job.color_sets.append(scenes.color_set(terms))
            e.color_id = len(job.color_sets)
    # entries were copied into the scene struct before this pass: write the ids through
    for li, (_, entries) in enumerate(layers):
        for ei, e in enumerate(entries):
            job.scene.layers[li].entries[ei].color_id = e.color_id
