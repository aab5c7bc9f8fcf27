"""Programmatic scene builders (the reference builds its test scenes the same way:
test/cpu_test_helpers.hpp:26-68 MakeCpuScene / MakeRectangularRender) and the BASELINE.json configs."""
from . import abi


def prism_crystal(height=1.0, face_distance=None, sync_group=None):
    c = abi.HaloCrystal()
    c.kind = abi.CRYSTAL_PRISM
    c.height[0] = abi.dist(height)
    fd = face_distance if face_distance is not None else [1.0] * 6
    for i in range(6):
        c.face_dist[i] = abi.dist(fd[i])
    for i in range(9):
        c.sync_group[i] = 0 if sync_group is None else int(sync_group[i])
    return c


ICE_CRYSTAL_C = 1.629  # kIceCrystalC (reference src/core/crystal.hpp)


def miller_wedge_deg(i1, i4):
    """Crystal::CreatePyramid(Miller) → wedge angle (reference crystal.cpp:413-426): atan((sqrt3/2) * i4 / i1 / c)."""
    import numpy as np
    if i1 == 0:
        return 0.0
    return float(np.float32(np.arctan(np.float32(np.float32(1.73205080757) / 2) * i4 / i1 / np.float32(ICE_CRYSTAL_C))) * np.float32(180.0 / 3.14159265359))


def pyramid_crystal(upper_h=0.2, prism_h=1.0, lower_h=0.2, upper_wedge=None, lower_wedge=None, face_distance=None,
                    upper_miller=(1, 1), lower_miller=(1, 1), sync_group=None):
    """PyramidCrystalParam: heights are fractions of the way from the shoulder to the natural apex; wedge angles in
    degrees, or Miller indices (i1, i4) (doc/configuration.md; examples/config_example.json crystal id 5)."""
    c = abi.HaloCrystal()
    c.kind = abi.CRYSTAL_PYRAMID
    c.height[0], c.height[1], c.height[2] = abi.dist(upper_h), abi.dist(prism_h), abi.dist(lower_h)
    fd = face_distance if face_distance is not None else [1.0] * 6
    for i in range(6):
        c.face_dist[i] = abi.dist(fd[i])
    for i in range(9):
        c.sync_group[i] = 0 if sync_group is None else int(sync_group[i])
    c.wedge_upper_deg = float(upper_wedge) if upper_wedge is not None else miller_wedge_deg(*upper_miller)
    c.wedge_lower_deg = float(lower_wedge) if lower_wedge is not None else miller_wedge_deg(*lower_miller)
    return c


def axis(zenith=None, azimuth=None, roll=None):
    """JSON-convention axis → internal AxisDistribution (reference src/core/math.cpp:679-726):
    latitude = 90 - zenith (center only; spread kept); when the `axis` object is present, azimuth and roll
    default to uniform over 360; when absent everything is fixed with zenith 0."""
    a = abi.HaloAxis()
    present = not (zenith is None and azimuth is None and roll is None)
    # seeded destination values an object with missing keys keeps (math.cpp:536-538, 692-714): zenith slot = the
    # AxisDistribution default latitude {none, 90, 0} (so a zenith object without "mean" has zenith centre 90 BEFORE the
    # 90 - x flip, i.e. latitude 0); azimuth / roll = {uniform, 0, 360}
    z = abi.dist(zenith, 0.0) if not isinstance(zenith, dict) else abi.dist(zenith, 90.0, 0.0)
    a.latitude.type, a.latitude.center, a.latitude.spread = z.type, 90.0 - z.center, z.spread
    full = {"type": "uniform", "mean": 0.0, "std": 360.0}
    a.azimuth = abi.dist(azimuth if azimuth is not None else (full if present else None), 0.0, 360.0)
    a.roll = abi.dist(roll if roll is not None else (full if present else None), 0.0, 360.0)
    return a


def entry(crystal, axis_dist, proportion=1.0, crystal_config_id=1, filter_id=0, color_id=0):
    e = abi.HaloEntry()
    e.color_id = int(color_id)
    e.crystal = crystal
    e.axis = axis_dist
    e.proportion = float(proportion)
    e.crystal_config_id = int(crystal_config_id)
    e.filter_id = int(filter_id)
    return e


def filter_term(kind="none", raypath=None, entry=None, exit=None, min_len=1, max_len=None, az=0.0, el=0.0, radii=0.0, crystal_id=0):
    """SimpleFilterParam (reference src/config/filter_config.cpp:62-118)."""
    t = abi.HaloFilterTerm()
    t.type = {"none": abi.FILTER_NONE, "raypath": abi.FILTER_RAYPATH, "entry_exit": abi.FILTER_ENTRY_EXIT,
              "direction": abi.FILTER_DIRECTION, "crystal": abi.FILTER_CRYSTAL}[kind]
    if raypath is not None:
        t.raypath_len = len(raypath)
        for i, f in enumerate(raypath):
            t.raypath[i] = int(f)
    t.has_entry, t.entry = (0, 0) if entry is None else (1, int(entry))
    t.has_exit, t.exit_face = (0, 0) if exit is None else (1, int(exit))
    t.min_len, t.max_len = int(min_len), 0 if max_len is None else int(max_len)
    t.az, t.el, t.radii, t.crystal_id = float(az), float(el), float(radii), int(crystal_id)
    return t



def symmetry_bits(symmetry):
    """FilterSymmetryFromString (config/filter_config.cpp:161-175): the letters P, B, D set their bit, every other character is
    ignored — "none", "" and "PBD" are all legal spellings in the reference's configs."""
    bits = 0
    for c in symmetry or "":
        bits |= {"P": abi.SYM_P, "B": abi.SYM_B, "D": abi.SYM_D}.get(c, 0)
    return bits

def _sym(symmetry):
    return symmetry_bits(symmetry)


def color_set(terms):
    """terms: list of (HaloFilterTerm predicate, symmetry string, bit) — one crystal entry's raypath-colour predicates
    (reference ColorGatePlacement, src/config/color_gate_table.hpp:62-81)."""
    cs = abi.HaloColorSet()
    cs.term_count = len(terms)
    for k, (pred, symmetry, bit) in enumerate(terms):
        cs.terms[k].predicate = pred
        cs.terms[k].symmetry = _sym(symmetry)
        cs.terms[k].bit = int(bit)
    return cs


def color_class(bits, combine="any"):
    """bits: iterable of bit numbers; combine 'any' | 'all' (ColorGateParams::color_class_bits / _combine)."""
    c = abi.HaloColorClass()
    c.bits = 0
    for b in bits:
        c.bits |= 1 << int(b)
    c.combine_all = 1 if combine == "all" else 0
    return c


def simple_filter(term, symmetry="", action="filter_in"):
    f = abi.HaloFilter()
    f.action = 0 if action == "filter_in" else 1
    f.symmetry = symmetry_bits(symmetry)
    f.is_complex = 0
    f.terms[0] = term
    return f


def complex_filter(or_clauses, symmetry="", action="filter_in"):
    """or_clauses: list of AND-clauses, each a list of HaloFilterTerm (ComplexFilterParam, filter_config.hpp:44-46)."""
    f = abi.HaloFilter()
    f.action = 0 if action == "filter_in" else 1
    f.symmetry = symmetry_bits(symmetry)
    f.is_complex = 1
    f.or_count = len(or_clauses)
    k = 0
    for i, clause in enumerate(or_clauses):
        f.and_counts[i] = len(clause)
        for t in clause:
            f.terms[k] = t
            k += 1
    return f


def scene(layers, max_hits=7, sun_altitude=20.0, sun_azimuth=0.0, sun_diameter=0.5):
    """layers: list of (prob, [entries])."""
    s = abi.HaloScene()
    s.sun_altitude, s.sun_azimuth, s.sun_diameter = float(sun_altitude), float(sun_azimuth), float(sun_diameter)
    s.max_hits = int(max_hits)
    s.layer_count = len(layers)
    for li, (prob, entries) in enumerate(layers):
        s.layers[li].prob = float(prob)
        s.layers[li].entry_count = len(entries)
        for ei, e in enumerate(entries):
            s.layers[li].entries[ei] = e
    return s


def render(lens=abi.LENS_FISHEYE_EQUAL_AREA, width=1920, height=1080, fov=180.0, az=0.0, el=30.0, ro=0.0,
           visible=abi.VISIBLE_UPPER, overlap=0.0, lens_shift=(0, 0)):
    r = abi.HaloRender()
    r.lens_type, r.fov, r.width, r.height = int(lens), float(fov), int(width), int(height)
    r.lens_shift[0], r.lens_shift[1] = int(lens_shift[0]), int(lens_shift[1])
    r.view_az, r.view_el, r.view_ro = float(az), float(el), float(ro)
    r.visible, r.overlap = int(visible), float(overlap)
    return r


def wl_discrete(wavelength, weight=1.0):
    w = abi.HaloWl()
    w.wavelength, w.weight, w.illuminant, w.pool_size = float(wavelength), float(weight), -1, 0
    return w


def wl_illuminant(name="D65", pool_size=64):
    w = abi.HaloWl()
    w.wavelength, w.weight, w.illuminant, w.pool_size = 0.0, 0.0, abi.ILLUM[name], int(pool_size)
    return w


# --- BASELINE.json configs -----------------------------------------------------------------
CONFIG_WAVELENGTHS_9 = [450.0 + 40.0 * i for i in range(9)]  # examples/config_example.json:217-227


def column_crystal_entry():
    """examples/config_example.json crystal id 3: prism h=1.3, zenith gauss(90, 0.3), azimuth/roll uniform 360."""
    return entry(prism_crystal(1.3, [1.0] * 6),
                 axis(zenith={"type": "gauss", "mean": 90, "std": 0.3}, azimuth={"type": "uniform", "mean": 0, "std": 360},
                      roll={"type": "uniform", "mean": 0, "std": 360}), proportion=10.0, crystal_config_id=3)


def config2_scene():
    """configs[1]: single-scatter hex column, max_hits 7."""
    return scene([(0.0, [column_crystal_entry()])], max_hits=7)


def config2_render(width=1920, height=1080):
    return render(abi.LENS_FISHEYE_EQUAL_AREA, width, height, fov=180.0, el=30.0, visible=abi.VISIBLE_UPPER)


def config3_scene():
    """configs[2]: plate (h=0.3, zenith gauss(0, 0.8)) prob 1.0 over a random column (full-sphere axis)."""
    plate = entry(prism_crystal(0.3), axis(zenith={"type": "gauss", "mean": 0, "std": 0.8}), 1.0, 6)
    col = entry(prism_crystal(1.3, [1.0] * 6),
                axis(zenith={"type": "uniform", "mean": 90, "std": 360}, azimuth={"type": "uniform", "mean": 0, "std": 360}), 1.0, 3)
    return scene([(1.0, [plate]), (0.0, [col])], max_hits=7)


def stochastic_prism_entry():
    """examples/bench_config_stoch.json: prism h=1, six face distances gauss(1, 0.15), full-sphere axis."""
    g = {"type": "gauss", "mean": 1.0, "std": 0.15}
    full = {"type": "uniform", "mean": 0.0, "std": 360.0}
    return entry(prism_crystal(1.0, [g] * 6), axis(zenith=full, azimuth=full, roll=full), 100.0, 1)
