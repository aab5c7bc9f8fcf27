"""ctypes mirror of include/halo_trace.h (the C-ABI boundary).  Plumbing only: no compute here.

Struct layouts follow the header field-for-field; `tests/test_abi_layout.py` checks sizes against the
compiled library (`halo_abi_sizeof`).
"""
import ctypes as C

HALO_OK, HALO_UNAVAILABLE, HALO_FATAL = 0, 1, 2
MAX_LAYERS, MAX_ENTRIES, MAX_HITS, MAX_FACES, MAX_FACE_VTX, MAX_TRIS = 4, 16, 64, 20, 12, 64
HALO_MAX_LAYERS, HALO_MAX_ENTRIES, HALO_MAX_HITS = MAX_LAYERS, MAX_ENTRIES, MAX_HITS
LUT_NODES, WL_POOL_MAX, PATH_CAP = 257, 255, 64

DIST_NONE, DIST_UNIFORM, DIST_GAUSS, DIST_ZIGZAG, DIST_LAPLACIAN, DIST_GAUSS_LEGACY = range(6)
CRYSTAL_PRISM, CRYSTAL_PYRAMID = 0, 1
(LENS_LINEAR, LENS_FISHEYE_EQUAL_AREA, LENS_FISHEYE_EQUIDISTANT, LENS_FISHEYE_STEREOGRAPHIC,
 LENS_DUAL_FISHEYE_EQUAL_AREA, LENS_DUAL_FISHEYE_EQUIDISTANT, LENS_DUAL_FISHEYE_STEREOGRAPHIC,
 LENS_RECTANGULAR, LENS_FISHEYE_ORTHOGRAPHIC, LENS_DUAL_FISHEYE_ORTHOGRAPHIC, LENS_GLOBE) = range(11)
VISIBLE_UPPER, VISIBLE_LOWER, VISIBLE_FULL = 0, 1, 2
ILLUM = {"D50": 0, "D55": 1, "D65": 2, "D75": 3, "A": 4, "E": 5}


class HaloDist(C.Structure):
    _fields_ = [("type", C.c_int32), ("center", C.c_float), ("spread", C.c_float)]


class HaloAxis(C.Structure):
    _fields_ = [("azimuth", HaloDist), ("latitude", HaloDist), ("roll", HaloDist)]


class HaloCrystal(C.Structure):
    _fields_ = [("kind", C.c_int32), ("height", HaloDist * 3), ("face_dist", HaloDist * 6),
                ("sync_group", C.c_int32 * 9), ("wedge_upper_deg", C.c_float), ("wedge_lower_deg", C.c_float)]


class HaloEntry(C.Structure):
    _fields_ = [("crystal", HaloCrystal), ("axis", HaloAxis), ("proportion", C.c_float),
                ("crystal_config_id", C.c_int32), ("filter_id", C.c_int32), ("color_id", C.c_int32)]


FILTER_MAX_OR, FILTER_MAX_TERMS = 64, 64
FILTER_NONE, FILTER_RAYPATH, FILTER_ENTRY_EXIT, FILTER_DIRECTION, FILTER_CRYSTAL = range(5)
SYM_P, SYM_B, SYM_D = 1, 2, 4


class HaloFilterTerm(C.Structure):
    _fields_ = [("type", C.c_int32), ("raypath_len", C.c_int32), ("raypath", C.c_uint8 * MAX_HITS), ("has_entry", C.c_int32),
                ("entry", C.c_int32), ("has_exit", C.c_int32), ("exit_face", C.c_int32), ("min_len", C.c_uint32),
                ("max_len", C.c_uint32), ("az", C.c_float), ("el", C.c_float), ("radii", C.c_float), ("crystal_id", C.c_int32)]


class HaloFilter(C.Structure):
    _fields_ = [("action", C.c_int32), ("symmetry", C.c_int32), ("is_complex", C.c_int32), ("or_count", C.c_int32),
                ("and_counts", C.c_int32 * FILTER_MAX_OR), ("terms", HaloFilterTerm * FILTER_MAX_TERMS)]


COLOR_MAX_TERMS, COLOR_MAX_CLASSES = 16, 16


class HaloColorTerm(C.Structure):
    _fields_ = [("predicate", HaloFilterTerm), ("symmetry", C.c_int32), ("bit", C.c_int32)]


class HaloColorSet(C.Structure):
    _fields_ = [("term_count", C.c_int32), ("reserved", C.c_int32), ("terms", HaloColorTerm * COLOR_MAX_TERMS)]


class HaloColorClass(C.Structure):
    _fields_ = [("bits", C.c_uint64), ("combine_all", C.c_int32), ("reserved", C.c_int32)]


class HaloLayer(C.Structure):
    _fields_ = [("prob", C.c_float), ("entry_count", C.c_int32), ("entries", HaloEntry * MAX_ENTRIES)]


class HaloScene(C.Structure):
    _fields_ = [("sun_altitude", C.c_float), ("sun_azimuth", C.c_float), ("sun_diameter", C.c_float),
                ("max_hits", C.c_int32), ("layer_count", C.c_int32), ("layers", HaloLayer * MAX_LAYERS)]


class HaloRender(C.Structure):
    _fields_ = [("lens_type", C.c_int32), ("fov", C.c_float), ("width", C.c_int32), ("height", C.c_int32),
                ("lens_shift", C.c_int32 * 2), ("view_az", C.c_float), ("view_el", C.c_float),
                ("view_ro", C.c_float), ("visible", C.c_int32), ("overlap", C.c_float)]


class HaloWl(C.Structure):
    _fields_ = [("wavelength", C.c_float), ("weight", C.c_float), ("illuminant", C.c_int32),
                ("pool_size", C.c_int32)]


class HaloHostRays(C.Structure):
    _fields_ = [("d", C.POINTER(C.c_float)), ("p", C.POINTER(C.c_float)), ("w", C.POINTER(C.c_float)),
                ("tf", C.POINTER(C.c_uint32)), ("crystal", C.c_void_p)]   # crystal: HaloGeomTables* or NULL


class HaloLayerStats(C.Structure):
    _fields_ = [("root_count", C.c_uint64), ("exit_count", C.c_uint64), ("continuation_count", C.c_uint64),
                ("exit_w_sum", C.c_double), ("kernel_ms", C.c_double), ("pixel_hits", C.c_uint64), ("launches", C.c_uint64)]


class HaloExitRecord(C.Structure):
    _fields_ = [("dir", C.c_float * 3), ("weight", C.c_float), ("root", C.c_uint32), ("seq", C.c_uint16),
                ("layer", C.c_uint8), ("path_len", C.c_uint8), ("path", C.c_uint8 * PATH_CAP),
                ("pixel", C.c_int32), ("crystal_id", C.c_uint16), ("wl_idx", C.c_uint16), ("color_mask", C.c_uint64)]


class HaloRouteInfo(C.Structure):
    _fields_ = [("launches", C.c_uint32), ("mode_mask", C.c_uint32), ("geom_mask", C.c_uint32), ("accum_mask", C.c_uint32),
                ("source_mask", C.c_uint32), ("plane_cnt", C.c_uint32), ("plane_copies", C.c_uint32), ("shuffle_chunk", C.c_uint32),
                ("spec_mask", C.c_uint32), ("generic_launches", C.c_uint32)]


MODE_PLAIN, MODE_FILTER, MODE_CAPTURE, MODE_GENERIC, MODE_COLOR = 1, 2, 4, 8, 16   # HaloRouteInfo.mode_mask bits
SPEC_LAST, SPEC_LENS, SPEC_VIS, SPEC_NOGATE = 1, 2, 4, 8                           # HaloRouteInfo.spec_mask bits


ACCUM_XYZ, ACCUM_SCALAR, ACCUM_BIN1, ACCUM_BIN2, ACCUM_LOG, ACCUM_LOG_XYZ, ACCUM_NONE = 1, 2, 4, 8, 16, 32, 64   # HaloRouteInfo.accum_mask bits


class HaloGeomTables(C.Structure):
    _fields_ = [("face_cnt", C.c_int32), ("face_n", C.c_float * (MAX_FACES * 3)), ("face_d", C.c_float * MAX_FACES),
                ("face_number", C.c_int32 * MAX_FACES), ("tri_cnt", C.c_int32), ("tri_v", C.c_float * (MAX_TRIS * 9)),
                ("tri_n", C.c_float * (MAX_TRIS * 3)), ("tri_area", C.c_float * MAX_TRIS),
                ("tri_face", C.c_int32 * MAX_TRIS)]


class HaloDisplay(C.Structure):
    _fields_ = [("intensity_factor", C.c_float), ("ray_color", C.c_float * 3), ("background", C.c_float * 3)]


COMPOSITE_DOMINANT, COMPOSITE_ADDITIVE, COMPOSITE_PAINTER = 0, 1, 2


class HaloCompositeClass(C.Structure):
    """Display half of a ColorClass (reference config/color_class_table.hpp:21-35)."""
    _fields_ = [("color", C.c_float * 3), ("z_order", C.c_int32), ("visible", C.c_int32), ("solo", C.c_int32)]


class HaloComposite(C.Structure):
    _fields_ = [("mode", C.c_int32), ("display_exposure_scale", C.c_float), ("intensity_factor", C.c_float), ("class_count", C.c_int32),
                ("classes", HaloCompositeClass * COLOR_MAX_CLASSES)]


def composite(classes, mode="painter", display_exposure_scale=1.0, intensity_factor=1.0):
    """HaloComposite from [{"color": (r, g, b), "visible": True, "solo": False, "z_order": i}, ...] (z_order defaults to the list
    position, like BuildColorClassTable) and a mode name or HALO_COMPOSITE_* value."""
    names = {"dominant": COMPOSITE_DOMINANT, "additive": COMPOSITE_ADDITIVE, "painter": COMPOSITE_PAINTER}
    spec = HaloComposite()
    spec.mode = mode if isinstance(mode, int) else names.get(mode, COMPOSITE_PAINTER)   # ParseCompositeMode: unknown -> painter
    spec.display_exposure_scale = float(display_exposure_scale)
    spec.intensity_factor = float(intensity_factor)
    spec.class_count = len(classes)
    for i, c in enumerate(classes):
        k = spec.classes[i]
        k.color = (C.c_float * 3)(*[float(v) for v in c.get("color", (1.0, 1.0, 1.0))])
        k.z_order = int(c.get("z_order", i))
        k.visible = 1 if c.get("visible", True) else 0
        k.solo = 1 if c.get("solo", False) else 0
    return spec


class ProjParams(C.Structure):
    """lm_proj::ProjParams — reference src/core/shared/projection_shared.h:106-118 (76 bytes)."""
    _fields_ = [("proj_type", C.c_int32), ("img_w", C.c_int32), ("img_h", C.c_int32), ("visible_range", C.c_int32),
                ("lens_shift_x", C.c_int32), ("lens_shift_y", C.c_int32), ("scale", C.c_float), ("az0", C.c_float),
                ("r_scale", C.c_float), ("max_abs_dz", C.c_float), ("rot", C.c_float * 9)]


def dist(spec=None, default=0.0, default_spread=0.0):
    """HaloDist from a JSON-style value: number | {"type","mean","std"} (reference doc/configuration.md).
    An object overwrites only the keys it carries: a missing "mean" / "std" keeps the DESTINATION slot's seeded value
    (`default` / `default_spread`), exactly like from_json(Distribution&) (reference src/core/math.cpp:593-630) — e.g. axis
    roll is seeded uniform / 0 / 360 (math.cpp:692-714), so {"type": "uniform"} there means a full turn, not spread 0."""
    d = HaloDist()
    if spec is None:
        d.type, d.center, d.spread = DIST_NONE, float(default), 0.0
    elif isinstance(spec, (int, float)):
        d.type, d.center, d.spread = DIST_NONE, float(spec), 0.0
    elif isinstance(spec, HaloDist):
        return spec
    else:
        names = {"none": DIST_NONE, "uniform": DIST_UNIFORM, "gauss": DIST_GAUSS, "zigzag": DIST_ZIGZAG,
                 "laplacian": DIST_LAPLACIAN, "gauss_legacy": DIST_GAUSS_LEGACY}
        d.type = names[spec["type"]]
        d.center = float(spec.get("mean", default))
        d.spread = float(spec.get("std", default_spread))
    return d
