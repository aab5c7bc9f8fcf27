// halo_host.cpp — host-side table producers (see halo_host.hpp).  Each routine names the reference
// routine it replaces (paths under /root/reference).  Compiled with -ffp-contract=off so that the float
// tables handed to the device are the ones the reference's host code would have produced.
#include "halo_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

#include "cie_tables.inc"
#include "halo_geom.h"

namespace halo {
namespace host {

namespace {

using Mat3 = std::array<float, 9>;

// Rotation::FillMat (geo3d.cpp:100-113): Rodrigues matrix for a unit axis.
Mat3 AxisAngle(const float ax[3], float theta) {
  const float c = std::cos(theta), s = std::sin(theta), cc = 1.0f - c;
  return {ax[0] * ax[0] * cc + c,         ax[0] * ax[1] * cc - ax[2] * s, ax[0] * ax[2] * cc + ax[1] * s,
          ax[0] * ax[1] * cc + ax[2] * s, ax[1] * ax[1] * cc + c,         ax[1] * ax[2] * cc - ax[0] * s,
          ax[0] * ax[2] * cc - ax[1] * s, ax[1] * ax[2] * cc + ax[0] * s, ax[2] * ax[2] * cc + c};
}

// Rotation::Chain (geo3d.cpp:32-46): m <- r * m, accumulating k = 0,1,2 in order.
Mat3 LeftMul(const Mat3& r, const Mat3& m) {
  Mat3 o{};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float acc = 0.0f;
      for (int k = 0; k < 3; k++) acc += r[i * 3 + k] * m[k * 3 + j];
      o[i * 3 + j] = acc;
    }
  return o;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// crystal geometry: the builders live in halo_geom.h, shared verbatim with the device generator
// (halo_shapegen.hip); here they are wrapped into the table layout of the C ABI.
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr double kPiD = 3.14159265358979323846;
}

uint32_t PcgHash(uint32_t x) { return geom::PcgHash32(x); }

void FromShapeDev(const ShapeDev& s, HaloGeomTables& g) {
  std::memset(&g, 0, sizeof(g));
  g.face_cnt = s.face_cnt;
  g.tri_cnt = s.tri_cnt;
  for (int f = 0; f < s.face_cnt; f++) {
    g.face_n[f * 3 + 0] = s.face[f][0];
    g.face_n[f * 3 + 1] = s.face[f][1];
    g.face_n[f * 3 + 2] = s.face[f][2];
    g.face_d[f] = s.face[f][3];
    g.face_number[f] = s.face_number[f];
  }
  for (int t = 0; t < s.tri_cnt; t++) {
    std::memcpy(g.tri_v + t * 9, s.tri_v[t], 36);
    g.tri_n[t * 3 + 0] = s.tri_na[t][0];
    g.tri_n[t * 3 + 1] = s.tri_na[t][1];
    g.tri_n[t * 3 + 2] = s.tri_na[t][2];
    g.tri_area[t] = s.tri_na[t][3];
    g.tri_face[t] = s.tri_face[t];
  }
}

bool BuildPrism(float h, const float dist[6], HaloGeomTables& out) {
  ShapeDev s;
  const bool ok = geom::BuildPrismShape(h, dist, s);
  FromShapeDev(s, out);
  return ok;
}

geom::CrystalRecipe MakeRecipe(const HaloCrystal& c) {
  geom::CrystalRecipe r{};
  r.c = c;
  auto cot = [](float wedge) {  // wedge legality + cone slope (FillHexCrystalCoef geo3d.cpp:346-512)
    if (!(wedge >= 0.1f && wedge <= 89.9f)) return -1.0;
    return static_cast<double>(kSqrt3 / 4.0f) / std::tan(static_cast<double>(wedge) * static_cast<double>(kDegToRad));
  };
  r.cot_u = cot(c.wedge_upper_deg);
  r.cot_l = cot(c.wedge_lower_deg);
  geom::FillDrawPlan(r);
  return r;
}

bool BuildPyramid(float wedge_u, float wedge_l, float h1, float h2, float h3, const float dist[6], HaloGeomTables& out) {
  HaloCrystal c{};
  c.wedge_upper_deg = wedge_u;
  c.wedge_lower_deg = wedge_l;
  const geom::CrystalRecipe r = MakeRecipe(c);
  ShapeDev s;
  const bool ok = geom::BuildPyramidShape(r.cot_u, r.cot_l, h1, h2, h3, dist, s);
  FromShapeDev(s, out);
  return ok;
}

void ToShapeDev(const HaloGeomTables& g, ShapeDev& s) {
  std::memset(&s, 0, sizeof(s));
  s.face_cnt = g.face_cnt;
  s.tri_cnt = g.tri_cnt;
  for (int f = 0; f < g.face_cnt; f++) {
    s.face[f][0] = g.face_n[f * 3 + 0];
    s.face[f][1] = g.face_n[f * 3 + 1];
    s.face[f][2] = g.face_n[f * 3 + 2];
    s.face[f][3] = g.face_d[f];
    s.face_number[f] = static_cast<uint8_t>(g.face_number[f]);
  }
  for (int t = 0; t < g.tri_cnt; t++) {
    std::memcpy(s.tri_v[t], g.tri_v + t * 9, 36);
    s.tri_na[t][0] = g.tri_n[t * 3 + 0];
    s.tri_na[t][1] = g.tri_n[t * 3 + 1];
    s.tri_na[t][2] = g.tri_n[t * 3 + 2];
    s.tri_na[t][3] = g.tri_area[t];
    s.tri_face[t] = static_cast<uint8_t>(g.tri_face[t]);
  }
  geom::ShapeCursor cur;
  cur.fid = g.face_cnt;
  cur.tri = g.tri_cnt;
  geom::FinalizeSlabs(s, cur);
}

// ---------------------------------------------------------------------------------------------------
// emit-gate filters
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr int kFnPeriod = 6;  // Crystal::fn_period_ for the hexagonal families (crystal.cpp:366,397)

std::vector<uint8_t> PShift(std::vector<uint8_t> rp) {  // Crystal::PCanonicalShift crystal.cpp:514-532
  int first = -1;
  for (uint8_t& x : rp) {
    if (x < 3) continue;
    const int pyr = x / 10;
    int pri = x % 10;
    if (first < 0) first = pri;
    pri = (pri + kFnPeriod - first) % kFnPeriod + 3;
    x = static_cast<uint8_t>(pyr * 10 + pri);
  }
  return rp;
}
}  // namespace

std::vector<uint8_t> ReduceRaypath(const std::vector<uint8_t>& rp, uint8_t symmetry, int sigma_a, bool d_applicable) {
  if (symmetry == 0) return rp;
  std::vector<uint8_t> cur = rp;
  if (symmetry & HALO_SYM_P) cur = PShift(cur);
  if ((symmetry & HALO_SYM_D) && d_applicable) {
    std::vector<uint8_t> refl = cur;
    for (uint8_t& x : refl) {
      if (x < 3) continue;
      const int pyr = x / 10;
      int pri = x % 10 - 3;
      pri = (sigma_a - pri + kFnPeriod) % kFnPeriod;
      x = static_cast<uint8_t>(pyr * 10 + pri + 3);
    }
    if (symmetry & HALO_SYM_P) refl = PShift(refl);
    if (refl < cur) cur = refl;
  }
  if (symmetry & HALO_SYM_B) {
    std::vector<uint8_t> refl = cur;
    bool changed = false;
    for (uint8_t& x : refl) {
      if (x <= 2) {
        x = static_cast<uint8_t>(3 - x);
        changed = true;
      } else if (x >= 13 && x <= 18) {
        x = static_cast<uint8_t>(x + 10);
        changed = true;
      } else if (x >= 23 && x <= 28) {
        x = static_cast<uint8_t>(x - 10);
        changed = true;
      }
    }
    if (changed && refl < cur) cur = refl;
  }
  return cur;
}

int ComputeSigmaA(float roll) {
  if (std::fabs(roll) > 1e6f) return 0;
  const int n = (static_cast<int>(std::round(roll / 30.0f)) % 6 + 6) % 6;
  return (6 - n) % 6;
}

bool IsDApplicable(const HaloAxis& a) {
  auto near = [](float x, float y) { return std::fabs(x - y) < kFloatEps; };
  const bool az_sym = a.azimuth.type == HALO_DIST_UNIFORM && near(a.azimuth.spread, 360.0f);  // IsAzRotationallySymmetric
  const float rem = std::fmod(std::fmod(a.roll.center, 30.0f) + 30.0f, 30.0f);
  return az_sym && (near(rem, 0.0f) || near(rem, 30.0f));
}

FilterDev BuildFilter(const HaloFilter& f, const HaloAxis& axis) {
  FilterDev d{};
  d.is_complex = f.is_complex ? 1 : 0;
  d.action = f.action ? 1 : 0;
  d.symmetry = static_cast<uint8_t>(f.symmetry & 7);
  const bool dap = IsDApplicable(axis);
  d.d_applicable = dap ? 1 : 0;
  d.sigma_a = dap ? ComputeSigmaA(axis.roll.center) : 0;
  auto fill = [&](const HaloFilterTerm& t, FilterTermDev& o) {
    o.type = static_cast<uint8_t>(t.type);
    std::vector<uint8_t> rp;
    if (t.type == HALO_FILTER_RAYPATH) {
      for (int i = 0; i < t.raypath_len && i < kFilterPathCap; i++) rp.push_back(t.raypath[i]);
    } else if (t.type == HALO_FILTER_ENTRY_EXIT) {
      o.has_entry = t.has_entry ? 1 : 0;
      o.has_exit = t.has_exit ? 1 : 0;
      o.min_len = t.min_len;
      o.max_len = t.max_len;
      if (t.has_entry) rp.push_back(static_cast<uint8_t>(t.entry));
      if (t.has_exit) rp.push_back(static_cast<uint8_t>(t.exit_face));
    } else if (t.type == HALO_FILTER_DIRECTION) {  // FillDirection device_filter_desc.cpp:57-65
      const float lon = t.az * kDegToRad, lat = t.el * kDegToRad;
      o.dir[0] = std::cos(lat) * std::cos(lon);
      o.dir[1] = std::cos(lat) * std::sin(lon);
      o.dir[2] = std::sin(lat);
      o.radii_c = std::cos(t.radii * kDegToRad);
    } else if (t.type == HALO_FILTER_CRYSTAL) {
      o.crystal_id = static_cast<uint32_t>(t.crystal_id);
    }
    if (!rp.empty()) {
      const std::vector<uint8_t> canon = ReduceRaypath(rp, d.symmetry, d.sigma_a, dap);
      o.canonical_len = static_cast<uint8_t>(canon.size());
      std::copy(canon.begin(), canon.end(), o.canonical);
      if (canon.size() <= 16) {  // packed form for the register path of the device filter
        uint64_t w[2] = {0, 0};
        for (size_t i = 0; i < canon.size(); i++) w[i / 8] |= static_cast<uint64_t>(canon[i]) << (8 * (7 - i % 8));
        o.canon_hi = w[0];
        o.canon_lo = w[1];
      }
    }
  };
  if (!f.is_complex) {
    fill(f.terms[0], d.terms[0]);
  } else {
    d.or_count = static_cast<uint32_t>(std::min(f.or_count, HALO_FILTER_MAX_OR));
    int k = 0;
    for (uint32_t o = 0; o < d.or_count; o++) {
      d.and_counts[o] = static_cast<uint8_t>(f.and_counts[o]);
      for (int a = 0; a < f.and_counts[o] && k < HALO_FILTER_MAX_TERMS; a++, k++) fill(f.terms[k], d.terms[k]);
    }
  }
  return d;
}

std::vector<std::array<uint64_t, 2>> RaypathMembers(const std::vector<uint8_t>& canon, uint8_t symmetry, int sigma_a, bool d_applicable) {
  std::vector<std::array<uint64_t, 2>> out;
  if (canon.empty() || canon.size() > 16) return out;   // longer than the path register: no path of a fast kernel can match
  auto pack = [](const std::vector<uint8_t>& q) {
    std::array<uint64_t, 2> w = {0ull, 0ull};   // {hi, lo}: element 0 is the oldest face, the last element sits in the low byte of lo
    for (size_t i = 0; i < q.size(); i++) {
      const size_t pos = q.size() - 1 - i;      // byte position from the low end
      w[pos < 8 ? 1 : 0] |= static_cast<uint64_t>(q[i]) << (8 * (pos % 8));
    }
    return w;
  };
  // reduce(p) is one of the images of p under {rotations} x {D} x {B}, so every p with reduce(p) == canon is an image of canon
  for (int rot = 0; rot < kFnPeriod; rot++)
    for (int dd = 0; dd < 2; dd++)
      for (int bb = 0; bb < 2; bb++) {
        std::vector<uint8_t> q = canon;
        for (uint8_t& x : q) {
          if (x >= 3) {
            const int pyr = x / 10;
            int pri = x % 10 - 3;
            pri = ((pri + rot) % kFnPeriod + kFnPeriod) % kFnPeriod;
            if (dd) pri = ((sigma_a - pri) % kFnPeriod + kFnPeriod) % kFnPeriod;
            x = static_cast<uint8_t>(pyr * 10 + pri + 3);
          }
          if (bb) {
            if (x <= 2) x = static_cast<uint8_t>(3 - x);
            else if (x >= 13 && x <= 18) x = static_cast<uint8_t>(x + 10);
            else if (x >= 23 && x <= 28) x = static_cast<uint8_t>(x - 10);
          }
        }
        if (ReduceRaypath(q, symmetry, sigma_a, d_applicable) != canon) continue;
        const auto w = pack(q);
        if (std::find(out.begin(), out.end(), w) == out.end()) out.push_back(w);
      }
  return out;
}

namespace {
struct FastTermView {   // the packed fields of a FastTerm (halo_device.h)
  uint32_t type, last, bit, len, min_len, max_len, orbit_n, orbit_off, ee_off;
};
FastTermView ViewTerm(const FastTerm& t) {
  return {t.w0 & 0xFFu, (t.w0 >> 8) & 0xFFu, (t.w0 >> 16) & 0xFFu, t.w0 >> 24, t.w1 & 0xFFu, (t.w1 >> 8) & 0xFFu, t.w1 >> 16, t.w2 & 0xFFFFu, t.w2 >> 16};
}
void PackTerm(const FastTermView& v, FastTerm& t) {
  t.w0 = (v.type & 0xFFu) | ((v.last & 0xFFu) << 8) | ((v.bit & 0xFFu) << 16) | ((v.len & 0xFFu) << 24);
  t.w1 = (v.min_len & 0xFFu) | ((v.max_len & 0xFFu) << 8) | (v.orbit_n << 16);
  t.w2 = (v.orbit_off & 0xFFFFu) | (v.ee_off << 16);
}
}  // namespace

bool BuildFastTables(const HaloFilter* filter, const HaloColorSet* colors, const HaloAxis& axis, uint32_t crystal_id, FastTables& out) {
  const bool dap = IsDApplicable(axis);
  const int sigma_a = dap ? ComputeSigmaA(axis.roll.center) : 0;
  uint32_t orbit_used = 0, ee_used = 0;
  bool fits = true;
  auto fill = [&](const HaloFilterTerm& t, uint8_t symmetry, uint32_t last, uint32_t bit, FastTerm& out_term) {
    out_term = FastTerm{};
    FastTermView o{};
    o.type = static_cast<uint32_t>(t.type);
    o.last = last;
    o.bit = bit;
    o.ee_off = 0xFFFFu;
    struct Packer {   // writes the packed words on every way out of the lambda
      FastTermView& v;
      FastTerm& t;
      ~Packer() { PackTerm(v, t); }
    } packer{o, out_term};
    if (t.type == HALO_FILTER_RAYPATH) {
      std::vector<uint8_t> rp;
      for (int i = 0; i < t.raypath_len && i < kFilterPathCap; i++) rp.push_back(t.raypath[i]);
      const std::vector<uint8_t> canon = ReduceRaypath(rp, symmetry, sigma_a, dap);
      o.len = static_cast<uint32_t>(std::min<size_t>(canon.size(), 255));
      const auto members = RaypathMembers(canon, symmetry, sigma_a, dap);
      const size_t padded = (members.size() + 7u) & ~size_t(7);   // the kernel compares eight members per scalar load; ~0 is no path
      if (orbit_used + padded > static_cast<size_t>(kFastOrbitCap)) {
        fits = false;
        return;
      }
      o.orbit_off = orbit_used;
      o.orbit_n = static_cast<uint32_t>(padded);
      for (size_t k = 0; k < padded; k++) {
        out.orbit_hi[orbit_used] = k < members.size() ? members[k][0] : ~0ull;
        out.orbit_lo[orbit_used] = k < members.size() ? members[k][1] : ~0ull;
        orbit_used++;
      }
    } else if (t.type == HALO_FILTER_ENTRY_EXIT) {  // DeviceFilterMatchSimple filter_shared.h:180-224 as a matrix over (entry face, exit face)
      o.min_len = std::min<uint32_t>(t.min_len, 255u);                       // (paths here have at most 16 faces)
      o.max_len = t.max_len == 0u ? 0u : std::min<uint32_t>(t.max_len, 255u);
      if (!t.has_entry && !t.has_exit) return;   // no face constraint (ee_off 0xFFFF): the length bounds are the whole term
      if (ee_used >= static_cast<uint32_t>(kFastEeCap)) {
        fits = false;
        return;
      }
      o.ee_off = ee_used;
      std::vector<uint8_t> want;
      if (t.has_entry) want.push_back(static_cast<uint8_t>(t.entry));
      if (t.has_exit) want.push_back(static_cast<uint8_t>(t.exit_face));
      const std::vector<uint8_t> canon = ReduceRaypath(want, symmetry, sigma_a, dap);
      for (uint32_t e = 0; e < 32; e++) {
        uint32_t row = 0;
        for (uint32_t x = 0; x < 32; x++) {
          std::vector<uint8_t> q;
          if (t.has_entry) q.push_back(static_cast<uint8_t>(e));
          if (t.has_exit) q.push_back(static_cast<uint8_t>(x));
          if (ReduceRaypath(q, symmetry, sigma_a, dap) == canon) row |= 1u << x;
        }
        out.ee[ee_used][e] = row;
      }
      ee_used++;
    } else if (t.type == HALO_FILTER_DIRECTION) {  // FillDirection device_filter_desc.cpp:57-65
      const float lon = t.az * kDegToRad, lat = t.el * kDegToRad;
      out_term.dir[0] = std::cos(lat) * std::cos(lon);
      out_term.dir[1] = std::cos(lat) * std::sin(lon);
      out_term.dir[2] = std::sin(lat);
      out_term.radii_c = std::cos(t.radii * kDegToRad);
    } else if (t.type == HALO_FILTER_CRYSTAL) {
      out_term.crystal_id = static_cast<uint32_t>(t.crystal_id);
    }
  };
  out.has_filter = out.action = out.term_cnt = out.color_terms = 0;
  out.dir0[0] = out.dir0[1] = out.dir0[2] = out.radii0 = 0.0f;
  out.len_mode = 0x5555555555555555ull;   // no filter: every length passes
  if (filter != nullptr) {
    const uint8_t sym = static_cast<uint8_t>(filter->symmetry & 7);
    out.has_filter = 1;
    out.action = filter->action ? 1u : 0u;
    uint32_t k = 0;
    if (!filter->is_complex) {
      fill(filter->terms[0], sym, 1u, 0u, out.fterm[k++]);
    } else {  // OR over AND-clauses as one flat list; an AND-clause without terms is true (filter_shared.h:263-291) = one pass-all term
      const int oc = std::min(filter->or_count, HALO_FILTER_MAX_OR);
      int src = 0;
      for (int o = 0; o < oc; o++) {
        const int n = filter->and_counts[o];
        if (n <= 0) {
          HaloFilterTerm none{};
          none.type = HALO_FILTER_NONE;
          fill(none, sym, 1u, 0u, out.fterm[k++]);
        }
        for (int a = 0; a < n && src < HALO_FILTER_MAX_TERMS; a++, src++) fill(filter->terms[src], sym, (a == n - 1) ? 1u : 0u, 0u, out.fterm[k++]);
      }
    }
    out.term_cnt = k;
    out.len_mode = 0;
    // fold what is known before the launch, per path length: 0 false, 1 true, 2 depends on the exit
    for (uint32_t L = 0; L <= 16; L++) {
      int m = 0, all = 1;   // OR accumulator (false), AND accumulator (true)
      for (uint32_t i = 0; i < k; i++) {
        const FastTermView t = ViewTerm(out.fterm[i]);
        int v = 2;
        if (t.type == HALO_FILTER_NONE) v = 1;
        else if (t.type == HALO_FILTER_RAYPATH) v = (L != t.len || t.orbit_n == 0) ? 0 : 2;
        else if (t.type == HALO_FILTER_ENTRY_EXIT) {
          if (L == 0 || L < t.min_len || (t.max_len != 0 && L > t.max_len)) v = 0;
          else v = t.ee_off == 0xFFFFu ? 1 : 2;
        } else if (t.type == HALO_FILTER_CRYSTAL) v = crystal_id == out.fterm[i].crystal_id ? 1 : 0;
        else if (t.type != HALO_FILTER_DIRECTION) v = 0;   // unknown type: never matches
        all = (all == 0 || v == 0) ? 0 : ((all == 1 && v == 1) ? 1 : 2);
        if (t.last) {
          m = (m == 1 || all == 1) ? 1 : ((m == 0 && all == 0) ? 0 : 2);
          all = 1;
        }
      }
      if (m != 2 && out.action) m = 1 - m;
      if (m == 2 && k == 1 && ViewTerm(out.fterm[0]).type == HALO_FILTER_DIRECTION) m = 3;   // one direction term: evaluated from the header
      out.len_mode |= static_cast<uint64_t>(m) << (2 * L);
    }
    if (k == 1 && ViewTerm(out.fterm[0]).type == HALO_FILTER_DIRECTION) {
      for (int a = 0; a < 3; a++) out.dir0[a] = out.fterm[0].dir[a];
      out.radii0 = out.fterm[0].radii_c;
    }
  }
  if (colors != nullptr) {
    out.color_terms = static_cast<uint32_t>(std::min(colors->term_count, HALO_COLOR_MAX_TERMS));
    for (uint32_t k = 0; k < out.color_terms; k++) {
      fill(colors->terms[k].predicate, static_cast<uint8_t>(colors->terms[k].symmetry & 7), 0u, static_cast<uint32_t>(colors->terms[k].bit & 0xFF), out.cterm[k]);
    }
  }
  return fits;
}

bool FastFilterCheck(const FastTables& F, const uint8_t* path, uint32_t L, const float dir[3], uint32_t crystal_id) {
  uint64_t hi = 0, lo = 0;   // the kernels' path register: newest face in the low byte
  for (uint32_t i = 0; i < L; i++) {
    hi = (hi << 8) | (lo >> 56);
    lo = (lo << 8) | path[i];
  }
  const uint32_t lm = static_cast<uint32_t>(F.len_mode >> (2 * L)) & 3u;
  if (lm == 3u) {
    const bool m = F.dir0[0] * dir[0] + F.dir0[1] * dir[1] + F.dir0[2] * dir[2] > F.radii0;
    return F.action == 0u ? m : !m;
  }
  if (lm != 2u) return lm == 1u;
  auto term = [&](const FastTerm& raw) {
    const FastTermView t = ViewTerm(raw);
    if (t.type == HALO_FILTER_NONE) return true;
    if (t.type == HALO_FILTER_RAYPATH) {
      if (L != t.len) return false;
      for (uint32_t k = 0; k < t.orbit_n; k++)
        if (lo == F.orbit_lo[t.orbit_off + k] && (L <= 8u || hi == F.orbit_hi[t.orbit_off + k])) return true;
      return false;
    }
    if (t.type == HALO_FILTER_ENTRY_EXIT) {
      if (L == 0u || L < t.min_len) return false;
      if (t.max_len != 0u && L > t.max_len) return false;
      if (t.ee_off == 0xFFFFu) return true;
      const uint32_t sh = 8u * (L - 1u);
      const uint32_t first = static_cast<uint32_t>((sh < 64u ? lo >> sh : hi >> (sh - 64u)) & 0xFFull), last = static_cast<uint32_t>(lo & 0xFFull);
      return first < 32u && last < 32u && ((F.ee[t.ee_off][first & 31u] >> last) & 1u) != 0u;
    }
    if (t.type == HALO_FILTER_DIRECTION) return raw.dir[0] * dir[0] + raw.dir[1] * dir[1] + raw.dir[2] * dir[2] > raw.radii_c;
    if (t.type == HALO_FILTER_CRYSTAL) return crystal_id == raw.crystal_id;
    return false;
  };
  bool m = false, all = true;
  for (uint32_t k = 0; k < F.term_cnt; k++) {
    all = term(F.fterm[k]) && all;
    if (ViewTerm(F.fterm[k]).last) {
      m = m || all;
      all = true;
    }
  }
  return F.action == 0u ? m : !m;
}

uint64_t FastColorMask(const FastTables& F, uint64_t carried, const uint8_t* path, uint32_t L, const float dir[3], uint32_t crystal_id) {
  // the kernels' fast_color_bits (halo_trace.inl) on the host: every colour predicate that matches ORs its bit into the carried mask
  uint64_t hi = 0, lo = 0;
  for (uint32_t i = 0; i < L; i++) {
    hi = (hi << 8) | (lo >> 56);
    lo = (lo << 8) | path[i];
  }
  uint64_t mask = carried;
  for (uint32_t k = 0; k < F.color_terms; k++) {
    const FastTerm& raw = F.cterm[k];
    const FastTermView t = ViewTerm(raw);
    bool m = false;
    if (t.type == HALO_FILTER_NONE) m = true;
    else if (t.type == HALO_FILTER_RAYPATH) {
      if (L == t.len)
        for (uint32_t j = 0; j < t.orbit_n && !m; j++) m = lo == F.orbit_lo[t.orbit_off + j] && (L <= 8u || hi == F.orbit_hi[t.orbit_off + j]);
    } else if (t.type == HALO_FILTER_ENTRY_EXIT) {
      if (!(L == 0u || L < t.min_len || (t.max_len != 0u && L > t.max_len))) {
        if (t.ee_off == 0xFFFFu) m = true;
        else {
          const uint32_t sh = 8u * (L - 1u);
          const uint32_t first = static_cast<uint32_t>((sh < 64u ? lo >> sh : hi >> (sh - 64u)) & 0xFFull), last = static_cast<uint32_t>(lo & 0xFFull);
          m = first < 32u && last < 32u && ((F.ee[t.ee_off][first & 31u] >> last) & 1u) != 0u;
        }
      }
    } else if (t.type == HALO_FILTER_DIRECTION) m = raw.dir[0] * dir[0] + raw.dir[1] * dir[1] + raw.dir[2] * dir[2] > raw.radii_c;
    else if (t.type == HALO_FILTER_CRYSTAL) m = crystal_id == raw.crystal_id;
    if (m && t.bit < 64u) mask |= 1ull << t.bit;
  }
  return mask;
}

bool BuildEntryFast(const ShapeDev& s, EntryFastDev& out) {
  std::memset(&out, 0, sizeof(out));
  if (s.face_cnt != kEntryFastFaces || s.slab_cnt != 4 || s.single_cnt != 0 || s.tri_cnt < 8 || s.tri_cnt > kEntryFastTris) return false;
  static const int want_plus[4] = {0, 2, 3, 4}, want_minus[4] = {1, 5, 6, 7};
  for (int k = 0; k < 4; k++) {
    int32_t ip, im;
    std::memcpy(&ip, &s.slab[k][5], 4);
    std::memcpy(&im, &s.slab[k][6], 4);
    if (ip != want_plus[k] || im != want_minus[k]) return false;
  }
  int next = 0;
  float face_area[kEntryFastFaces];
  for (int f = 0; f < kEntryFastFaces; f++) {   // triangles grouped face by face, in face order, 1..4 per face
    int cnt = 0;
    while (next + cnt < s.tri_cnt && s.tri_face[next + cnt] == f) cnt++;
    if (cnt < 1 || cnt > 4) return false;
    float a = 0.0f;
    for (int k = 0; k < cnt; k++) {
      out.tri_area[f][k] = s.tri_na[next + k][3];
      a += s.tri_na[next + k][3];   // the partial sums the device's per-face view (FaceIndex) forms, in the same order
    }
    face_area[f] = a;
    out.tri0n[f] = static_cast<uint32_t>(next) | (static_cast<uint32_t>(cnt) << 8);
    next += cnt;
  }
  if (next != s.tri_cnt) return false;
  for (int k = 0; k < 4; k++) {
    out.slab_area[k][0] = face_area[want_plus[k]];
    out.slab_area[k][1] = face_area[want_minus[k]];
  }
  for (int t = 0; t < s.tri_cnt; t++)
    for (int c = 0; c < 9; c++) out.tri_v[t][c] = s.tri_v[t][c];
  // regular hexagonal prism?  slab normals = the builder's exact table values, one plane constant for both basal faces, one
  // for all six sides (the kernel's literals: halo_trace.inl kHexC / kHexS)
  {
    const float c60 = 0.5f, s60 = static_cast<float>(geom::Sin6(1));
    const float want[4][3] = {{0.0f, 0.0f, 1.0f}, {1.0f, 0.0f, 0.0f}, {c60, s60, 0.0f}, {-c60, s60, 0.0f}};
    bool reg = true;
    for (int k = 0; k < 4 && reg; k++) reg = s.slab[k][0] == want[k][0] && s.slab[k][1] == want[k][1] && s.slab[k][2] == want[k][2] && s.slab[k][3] == s.slab[k][4];
    for (int k = 2; k < 4 && reg; k++) reg = s.slab[k][3] == s.slab[1][3];
    for (int f = 0; f < kEntryFastFaces && reg; f++) reg = s.face_number[f] == f + 1;   // the path recorder of the filter kernels counts on it
    out.hex_regular = reg ? 1u : 0u;
    out.hex_d_basal = s.slab[0][3];
    out.hex_d_side = s.slab[1][3];
  }
  return true;
}

bool IsDeterministic(const HaloCrystal& c) {
  const int nh = (c.kind == HALO_CRYSTAL_PRISM) ? 1 : 3;
  for (int i = 0; i < nh; i++)
    if (c.height[i].type != HALO_DIST_NONE) return false;
  for (int i = 0; i < 6; i++)
    if (c.face_dist[i].type != HALO_DIST_NONE) return false;
  return true;
}

bool MakeShapeDev(uint32_t seed, const HaloCrystal& c, uint64_t shape_index, ShapeDev& out) {
  return geom::MakeShapeDev(seed, MakeRecipe(c), shape_index, out);
}

bool MakeShape(uint32_t seed, const HaloCrystal& c, uint64_t shape_index, HaloGeomTables& out) {
  ShapeDev s;
  const bool ok = MakeShapeDev(seed, c, shape_index, s);
  FromShapeDev(s, out);
  return ok;
}

// ---------------------------------------------------------------------------------------------------
// latitude LUT (lat_lut.cpp:24-204)
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr int kFine = 4096;
constexpr int kQuad = 1 << 16;

void FoldLatitude(float phi, float& phi_out, bool& flip) {  // lm_pcg::normalize_latitude pcg_shared.h:311-322
  const float pi = 3.14159265358979323846f, half_pi = 1.5707963267948966f;
  float theta = half_pi - phi;
  theta = std::fmod(theta, 2.0f * pi);
  if (theta < 0.0f) theta += 2.0f * pi;
  flip = theta > pi;
  if (flip) theta = 2.0f * pi - theta;
  phi_out = half_pi - theta;
}
}  // namespace

LatLut BuildLatLut(const HaloDist& lat) {
  LatLut lut;
  const double mean = static_cast<double>(lat.center) * (kPiD / 180.0);
  const double scale = static_cast<double>(lat.spread) * (kPiD / 180.0);
  const double dtheta = kPiD / kFine;
  std::vector<double> mass(kFine, 0.0), fmass(kFine, 0.0);
  auto deposit = [&](double latitude, double weight) {
    float folded = 0.0f;
    bool flip = false;
    FoldLatitude(static_cast<float>(latitude), folded, flip);
    const double colat = kPiD / 2.0 - static_cast<double>(folded);
    const double w = weight * std::sin(colat);
    if (w <= 0.0) return;
    const int bin = std::min(std::max(static_cast<int>(colat / dtheta), 0), kFine - 1);
    mass[bin] += w;
    if (flip) fmass[bin] += w;
  };
  if (lat.type == HALO_DIST_GAUSS) {
    const double lo = mean - 12.0 * scale, hi = mean + 12.0 * scale, dL = (hi - lo) / kQuad;
    const double inv2s2 = scale > 0.0 ? 1.0 / (2.0 * scale * scale) : 0.0;
    for (int i = 0; i < kQuad; ++i) {
      const double L = lo + (i + 0.5) * dL, d = L - mean;
      deposit(L, std::exp(-d * d * inv2s2) * dL);
    }
  } else {
    const double dU = 1.0 / kQuad;
    for (int i = 0; i < kQuad; ++i) {
      const double u = (i + 0.5) * dU;
      double L = mean;
      if (lat.type == HALO_DIST_UNIFORM) L = (u - 0.5) * scale + mean;
      else if (lat.type == HALO_DIST_ZIGZAG) L = std::fabs(scale * std::sin(u * 2.0 * kPiD) + mean);
      else if (lat.type == HALO_DIST_LAPLACIAN) {
        const double sgn = (u < 0.5) ? -1.0 : 1.0;
        L = mean - scale * sgn * std::log(std::max(1.0 - 2.0 * std::fabs(u - 0.5), 1e-30));
      }
      deposit(L, dU);
    }
  }
  std::vector<double> cm(kFine + 1, 0.0), cf(kFine + 1, 0.0);
  for (int i = 0; i < kFine; ++i) {
    cm[i + 1] = cm[i] + mass[i];
    cf[i + 1] = cf[i] + fmass[i];
  }
  const double total = cm[kFine];
  auto delta_at = [&](double colat) {  // DegenerateLut lat_lut.cpp:63-72
    const float c = static_cast<float>(std::min(std::max(colat, 0.0), kPiD));
    for (int i = 0; i < kLutNodes; ++i) {
      lut.theta[i] = c;
      lut.cdf[i] = static_cast<float>(i) / static_cast<float>(kLutNodes - 1);
      lut.flip[i] = 0.0f;
    }
  };
  if (!(total > 0.0)) {
    float folded = 0.0f;
    bool flip = false;
    FoldLatitude(static_cast<float>(mean), folded, flip);
    delta_at(kPiD / 2.0 - static_cast<double>(folded));
    return lut;
  }
  double t_lo = 0.0, t_hi = kPiD;
  for (int i = 0; i <= kFine; ++i)
    if (cm[i] / total >= 1e-7) {
      t_lo = i * dtheta;
      break;
    }
  for (int i = kFine; i >= 0; --i)
    if (cm[i] / total <= 1.0 - 1e-7) {
      t_hi = i * dtheta;
      break;
    }
  if (!(t_hi > t_lo)) {
    delta_at(0.5 * (t_lo + t_hi));
    return lut;
  }
  auto interp = [&](const std::vector<double>& cum, double theta) {
    const double x = theta / dtheta;
    const int i = static_cast<int>(x);
    if (i < 0) return cum.front();
    if (i >= kFine) return cum.back();
    const double f = x - i;
    return cum[i] * (1.0 - f) + cum[i + 1] * f;
  };
  const double span = t_hi - t_lo;
  for (int n = 0; n < kLutNodes; ++n) {
    const double t = t_lo + span * n / (kLutNodes - 1);
    lut.theta[n] = static_cast<float>(t);
    lut.cdf[n] = static_cast<float>(interp(cm, t) / total);
  }
  for (int n = 1; n < kLutNodes; ++n)
    if (lut.cdf[n] <= lut.cdf[n - 1]) lut.cdf[n] = std::nextafter(lut.cdf[n - 1], std::numeric_limits<float>::infinity());
  for (int n = 0; n + 1 < kLutNodes; ++n) {
    const double t0 = lut.theta[n], t1 = lut.theta[n + 1];
    const double m = interp(cm, t1) - interp(cm, t0);
    const double fm = interp(cf, t1) - interp(cf, t0);
    lut.flip[n] = (m > 0.0) ? static_cast<float>(std::min(std::max(fm / m, 0.0), 1.0)) : 0.0f;
  }
  lut.flip[kLutNodes - 1] = lut.flip[kLutNodes - 2];
  return lut;
}

uint32_t SelectLatPath(const HaloAxis& a) {
  auto near = [](float x, float y) { return std::fabs(x - y) < kFloatEps; };
  const bool full = a.azimuth.type == HALO_DIST_UNIFORM && near(a.azimuth.center, 0.0f) && near(a.azimuth.spread, 360.0f) &&
                    a.latitude.type == HALO_DIST_UNIFORM && near(a.latitude.center, 90.0f) && near(a.latitude.spread, 360.0f);
  if (full) return kLatFullSphere;
  if (a.latitude.type == HALO_DIST_NONE) return kLatNoRandom;
  if (a.latitude.type == HALO_DIST_GAUSS_LEGACY) return kLatGaussLegacy;
  return kLatLut;
}

// ---------------------------------------------------------------------------------------------------
// projection POD (scatter_accum.hpp:18-27, lens_proj_build.hpp:22-137, projection.cpp:192-204)
// ---------------------------------------------------------------------------------------------------
ProjDev BuildProj(const HaloRender& cfg) {
  ProjDev p{};
  p.proj_type = cfg.lens_type;
  p.img_w = cfg.width;
  p.img_h = cfg.height;
  p.visible_range = cfg.visible;
  p.lens_shift_x = cfg.lens_shift[0];
  p.lens_shift_y = cfg.lens_shift[1];
  p.scale = 1.0f;
  p.az0 = 0.0f;
  p.r_scale = 1.0f;
  p.max_abs_dz = 0.0f;
  const float ez[3] = {0, 0, 1}, ey[3] = {0, 1, 0};
  Mat3 cam = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  cam = LeftMul(AxisAngle(ez, (-90.0f + cfg.view_ro) * kDegToRad), cam);
  cam = LeftMul(AxisAngle(ey, (90.0f - cfg.view_el) * kDegToRad), cam);
  cam = LeftMul(AxisAngle(ez, cfg.view_az * kDegToRad), cam);
  std::copy(cam.begin(), cam.end(), p.rot);
  const float short_pix = static_cast<float>(std::min(cfg.width, cfg.height));
  const float fov = cfg.fov * kDegToRad;
  switch (cfg.lens_type) {
    case HALO_LENS_LINEAR:
    case HALO_LENS_GLOBE: p.scale = short_pix / 2.0f / std::tan(fov / 2.0f); break;
    case HALO_LENS_FISHEYE_EQUAL_AREA: p.scale = short_pix / 2.0f / std::sqrt(2.0f) / std::sin(fov / 4.0f); break;
    case HALO_LENS_FISHEYE_EQUIDISTANT: p.scale = short_pix * kPiHalf / fov; break;
    case HALO_LENS_FISHEYE_STEREOGRAPHIC: p.scale = short_pix / 2.0f / std::tan(fov / 4.0f); break;
    case HALO_LENS_FISHEYE_ORTHOGRAPHIC: p.scale = short_pix / 2.0f / std::sin(fov / 2.0f); break;
    case HALO_LENS_RECTANGULAR: {
      p.scale = static_cast<float>(std::min(cfg.width / 2, cfg.height)) / kPi;
      const float zx = cam[2], zy = cam[5];  // cam * (0,0,1)
      p.az0 = std::atan2(zy, zx);
      break;
    }
    default: break;
  }
  if (cfg.overlap > 0) {
    if (cfg.lens_type == HALO_LENS_DUAL_FISHEYE_EQUAL_AREA) {
      p.max_abs_dz = cfg.overlap;
      p.r_scale = 1.0f / std::sqrt(1.0f + cfg.overlap);
    } else if (cfg.lens_type == HALO_LENS_DUAL_FISHEYE_EQUIDISTANT) {
      p.max_abs_dz = cfg.overlap;
      p.r_scale = kPiHalf / (kPiHalf + std::asin(cfg.overlap));
    } else if (cfg.lens_type == HALO_LENS_DUAL_FISHEYE_STEREOGRAPHIC) {
      p.max_abs_dz = cfg.overlap;
      p.r_scale = 1.0f / std::tan((kPiHalf + std::asin(cfg.overlap)) / 2.0f);
    }
  }
  return p;
}

// ---------------------------------------------------------------------------------------------------
// spectrum
// ---------------------------------------------------------------------------------------------------
double IceRefractiveIndex(double wl) {  // optics.cpp:180-197 with kCoefAvr optics.hpp:30
  const float B1 = 0.701777f, B2 = 1.091144f, C1 = 0.884400f, C2 = 0.796950f;
  if (wl < 350.0f || wl > 900.0f) return 1.0f;
  wl /= 1e3;
  double n = 1.0;
  n += B1 / (1 - C1 * 1e-2f / wl / wl);
  n += B2 / (1 - C2 * 1e2f / wl / wl);
  return std::sqrt(n);
}

namespace {
float Daylight(float cct, float wl) {  // illuminant.cpp:13-87
  if (wl < HALO_DAY_MIN_NM || wl > HALO_DAY_MAX_NM) return 0.0f;
  const float ti = 1.0f / cct, ti2 = ti * ti, ti3 = ti2 * ti;
  const float xd = (cct <= 7000.0f) ? 0.244063f + 0.09911e3f * ti + 2.9678e6f * ti2 - 4.6070e9f * ti3
                                    : 0.237040f + 0.24748e3f * ti + 1.9018e6f * ti2 - 2.0064e9f * ti3;
  const float yd = -3.000f * xd * xd + 2.870f * xd - 0.275f;
  const float den = 0.0241f + 0.2562f * xd - 0.7341f * yd;
  const float m1 = (-1.3515f - 1.7703f * xd + 5.9114f * yd) / den;
  const float m2 = (0.0300f - 31.4424f * xd + 30.0717f * yd) / den;
  constexpr int np = static_cast<int>(sizeof(HALO_DAYLIGHT) / sizeof(HALO_DAYLIGHT[0]));
  const float fi = (wl - HALO_DAY_MIN_NM) / static_cast<float>(HALO_DAY_STEP_NM);
  int i0 = static_cast<int>(fi);
  float frac = fi - static_cast<float>(i0);
  if (i0 >= np - 1) {
    i0 = np - 1;
    frac = 0.0f;
  }
  const int i1 = i0 + (i0 < np - 1 ? 1 : 0);
  auto mix = [&](int ch) { return HALO_DAYLIGHT[i0][ch] + frac * (HALO_DAYLIGHT[i1][ch] - HALO_DAYLIGHT[i0][ch]); };
  return mix(0) + m1 * mix(1) + m2 * mix(2);
}
}  // namespace

float IlluminantSpd(int type, float wl) {  // illuminant.cpp:113-134
  switch (type) {
    case HALO_ILLUM_D50: return Daylight(5003.0f, wl);
    case HALO_ILLUM_D55: return Daylight(5503.0f, wl);
    case HALO_ILLUM_D65: return Daylight(6504.0f, wl);
    case HALO_ILLUM_D75: return Daylight(7504.0f, wl);
    case HALO_ILLUM_A: {
      if (wl < HALO_DAY_MIN_NM || wl > HALO_DAY_MAX_NM || wl <= 0.0f) return 0.0f;
      const float ratio = 560.0f / wl;
      const float r5 = ratio * ratio * ratio * ratio * ratio;
      const float e_ref = std::exp(1.4388e7f / (2856.0f * 560.0f));
      const float e_lam = std::exp(1.4388e7f / (2856.0f * wl));
      return 100.0f * r5 * (e_ref - 1.0f) / (e_lam - 1.0f);
    }
    case HALO_ILLUM_E: return (wl < HALO_DAY_MIN_NM || wl > HALO_DAY_MAX_NM) ? 0.0f : 1.0f;
    default: return 0.0f;
  }
}

std::vector<WlEntryDev> BuildWlPool(const HaloWl& wl) {  // ComputeWlPool wl_pool.hpp:67-91
  auto entry = [](float lambda, float weight) {
    WlEntryDev e{};
    e.n_idx = static_cast<float>(IceRefractiveIndex(lambda));
    e.spd_weight = weight;
    const int key = static_cast<int>(lambda + 0.5f);
    if (key >= HALO_CMF_MIN_NM && key <= HALO_CMF_MAX_NM) {
      e.cmf_x = HALO_CMF[key - HALO_CMF_MIN_NM][0];
      e.cmf_y = HALO_CMF[key - HALO_CMF_MIN_NM][1];
      e.cmf_z = HALO_CMF[key - HALO_CMF_MIN_NM][2];
    }
    return e;
  };
  std::vector<WlEntryDev> pool;
  if (wl.illuminant >= 0) {
    uint32_t M = wl.pool_size > 0 ? static_cast<uint32_t>(wl.pool_size) : 64u;
    M = std::min<uint32_t>(M, HALO_WL_POOL_MAX);
    for (uint32_t m = 0; m < M; ++m) {
      const float lambda = 380.0f + (static_cast<float>(m) + 0.5f) * 400.0f / static_cast<float>(M);
      pool.push_back(entry(lambda, IlluminantSpd(wl.illuminant, lambda)));
    }
  } else {
    pool.push_back(entry(wl.wavelength, wl.weight));
  }
  return pool;
}

// ---------------------------------------------------------------------------------------------------
// PartitionCrystalRayNum (simulator.cpp:519-582): floor + carry, then largest-remainder correction
// ---------------------------------------------------------------------------------------------------
std::vector<uint64_t> Partition(const float* prop, int n, uint64_t ray_num, double* carry) {
  std::vector<uint64_t> out(static_cast<size_t>(n), 0);
  if (n == 0 || ray_num == 0) return out;
  float total = 0.0f;
  for (int i = 0; i < n; i++) total += std::max(0.0f, prop[i]);
  if (total <= 0.0f) return out;
  uint64_t assigned = 0;
  for (int i = 0; i < n; i++) {
    const double ideal = carry[i] + (static_cast<double>(std::max(0.0f, prop[i])) / total) * ray_num;
    const uint64_t alloc = static_cast<uint64_t>(std::max(0.0, ideal));
    carry[i] = ideal - static_cast<double>(alloc);
    out[i] = alloc;
    assigned += alloc;
  }
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  if (assigned < ray_num) {
    const uint64_t deficit = std::min<uint64_t>(ray_num - assigned, n);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return carry[a] > carry[b]; });
    for (uint64_t k = 0; k < deficit; k++) {
      out[order[k]]++;
      carry[order[k]] -= 1.0;
    }
  } else if (assigned > ray_num) {
    uint64_t surplus = assigned - ray_num;
    std::vector<int> reclaim;
    for (int i : order)
      if (out[i] > 0) reclaim.push_back(i);
    std::stable_sort(reclaim.begin(), reclaim.end(), [&](int a, int b) { return carry[a] < carry[b]; });
    for (size_t k = 0; k < reclaim.size() && surplus > 0; k++, surplus--) {
      out[reclaim[k]]--;
      carry[reclaim[k]] += 1.0;
    }
  }
  return out;
}

}  // namespace host
}  // namespace halo
