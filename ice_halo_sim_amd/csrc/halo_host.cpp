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
// RNG
// ---------------------------------------------------------------------------------------------------
uint32_t PcgHash(uint32_t x) {  // pcg_shared.h:193-197
  x = x * 747796405u + 2891336453u;
  x = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
  return (x >> 22u) ^ x;
}
float Pcg::Uniform() {
  uint32_t h = PcgHash(seed ^ PcgHash(key + slot));
  slot++;
  return static_cast<float>(h >> 8) * (1.0f / 16777216.0f);
}
float Pcg::Gaussian() {
  const float two_pi = 2.0f * 3.14159265358979323846f;
  float u1 = std::fmax(Uniform(), 1e-7f);
  float u2 = Uniform();
  return std::sqrt(-2.0f * std::log(u1)) * std::cos(two_pi * u2);
}
float Pcg::Get(const HaloDist& d) {  // RandomNumberGenerator::Get math.cpp:418-444 over the PCG stream
  const float two_pi = 2.0f * 3.14159265358979323846f;
  switch (d.type) {
    case HALO_DIST_UNIFORM: return (Uniform() - 0.5f) * d.spread + d.center;
    case HALO_DIST_GAUSS:
    case HALO_DIST_GAUSS_LEGACY: return Gaussian() * d.spread + d.center;
    case HALO_DIST_ZIGZAG: return std::fabs(d.spread * std::sin(Uniform() * two_pi) + d.center);
    case HALO_DIST_LAPLACIAN: {
      float u = Uniform();
      float sgn = (u < 0.5f) ? -1.0f : 1.0f;
      float arg = std::fmax(1.0f - 2.0f * std::fabs(u - 0.5f), 1e-30f);
      return d.center - d.spread * sgn * std::log(arg);
    }
    default: return d.center;
  }
}

// ---------------------------------------------------------------------------------------------------
// prism geometry
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr double kPiD = 3.14159265358979323846;
// exact 60-degree direction tables (geo3d_closedform.hpp:48-52)
constexpr double kCos6[6] = {1.0, 0.5, -0.5, -1.0, -0.5, 0.5};
constexpr double kS = 0.86602540378443864676;
constexpr double kSin6[6] = {0.0, kS, kS, 0.0, -kS, -kS};

struct Pt {
  double x, y;
};

// intersection of half-plane boundaries i and j (Cramer, geo3d_closedform.cpp:27-35)
bool Meet(int i, int j, const double r[6], Pt& out) {
  const double det = kCos6[i] * kSin6[j] - kSin6[i] * kCos6[j];
  if (det == 0.0) return false;
  out.x = (r[i] * kSin6[j] - r[j] * kSin6[i]) / det;
  out.y = (kCos6[i] * r[j] - kCos6[j] * r[i]) / det;
  return true;
}

struct HexSection {
  std::vector<Pt> ring;          // CCW corners, one per adjacent pair of present sides
  std::array<bool, 6> present{}; // side bounds the polygon
  bool bounded = false;
};

// 2-D intersection of the six half-planes cos(i*60)x + sin(i*60)y <= r[i] (SolveHexCrossSection,
// geo3d_closedform.cpp:124-302): enumerate non-parallel pairs, keep feasible corners, dedupe within
// tol = 5*eps*max|r|, a side is present iff >= 2 corners sit on it, then walk present sides in order.
HexSection SolveHex(const double r[6]) {
  HexSection hs;
  double scale = 0.0;
  for (int i = 0; i < 6; i++) scale = std::max(scale, std::fabs(r[i]));
  const double tol = 5.0 * static_cast<double>(kFloatEps) * scale;
  std::vector<Pt> cand;
  for (int i = 0; i < 6; i++)
    for (int j = i + 1; j < 6; j++) {
      if (j == i + 3) continue;
      Pt q{};
      Meet(i, j, r, q);
      bool ok = true;
      for (int m = 0; m < 6 && ok; m++)
        if (m != i && m != j && kCos6[m] * q.x + kSin6[m] * q.y > r[m] + tol) ok = false;
      if (!ok) continue;
      bool dup = false;
      for (const Pt& c : cand)
        if (std::sqrt((c.x - q.x) * (c.x - q.x) + (c.y - q.y) * (c.y - q.y)) <= tol) {
          dup = true;
          break;
        }
      if (!dup && cand.size() < 12) cand.push_back(q);
    }
  std::vector<int> sides;
  for (int i = 0; i < 6; i++) {
    int on = 0;
    for (const Pt& c : cand)
      if (std::fabs(kCos6[i] * c.x + kSin6[i] * c.y - r[i]) <= tol) on++;
    hs.present[i] = on >= 2;
    if (hs.present[i]) sides.push_back(i);
  }
  const int n = static_cast<int>(sides.size());
  bool opposite_adjacent = false;
  for (int k = 0; k < n; k++)
    if (std::abs(sides[k] - sides[(k + 1) % n]) == 3) opposite_adjacent = true;
  hs.bounded = n >= 3 && !opposite_adjacent;
  if (!hs.bounded) return hs;
  for (int k = 0; k < n; k++) {
    Pt q{};
    Meet(sides[k], sides[(k + 1) % n], r, q);
    hs.ring.push_back(q);
  }
  return hs;
}

struct FaceLoop {              // one face slot of CrystalGeom (crystal.hpp:78)
  bool present = false;
  float plane[4] = {0, 0, 0, 0};
  float normal[3] = {0, 0, 0};
  int number = 0;
  std::vector<std::array<float, 3>> loop;  // CCW corners seen from outside
};

// Compact the present faces and fan-triangulate them (Crystal::PopulateFromCfGeom crystal.cpp:304-347 +
// detail::BuildEntrySubTris simulator.cpp:90-129).
void Tabulate(const std::vector<FaceLoop>& faces, HaloGeomTables& out) {
  std::memset(&out, 0, sizeof(out));
  int fid = 0, t = 0;
  for (const FaceLoop& f : faces) {
    if (!f.present) continue;
    std::memcpy(out.face_n + fid * 3, f.normal, sizeof(f.normal));
    const float len = std::sqrt(f.plane[0] * f.plane[0] + f.plane[1] * f.plane[1] + f.plane[2] * f.plane[2]);
    out.face_d[fid] = (len > kFloatEps) ? f.plane[3] / len : 0.0f;
    out.face_number[fid] = f.number;
    const int nv = static_cast<int>(f.loop.size());
    for (int k = 1; k + 1 < nv && nv >= 3 && t < kMaxTris; k++) {
      float* v = out.tri_v + t * 9;
      std::memcpy(v, f.loop[0].data(), 12);
      std::memcpy(v + 3, f.loop[k].data(), 12);
      std::memcpy(v + 6, f.loop[k + 1].data(), 12);
      const float a[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]};
      const float b[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
      float nrm[3] = {-b[1] * a[2] + a[1] * b[2], b[0] * a[2] - a[0] * b[2], -b[0] * a[1] + a[0] * b[1]};  // Cross3 math.cpp:36
      const float mag = std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
      out.tri_area[t] = mag / 2.0f;
      for (float& c : nrm) c = (mag > 0.0f) ? c / mag : 0.0f;
      std::memcpy(out.tri_n + t * 3, nrm, sizeof(nrm));
      out.tri_face[t] = fid;
      t++;
    }
    fid++;
  }
  out.face_cnt = fid;
  out.tri_cnt = t;
}

}  // namespace

// ComputeClosedFormPrism (geo3d_closedform.cpp:1318-1407) + AdaptClosedFormPrismToCrystalGeom (crystal.cpp:109-186)
bool BuildPrism(float h, const float dist[6], HaloGeomTables& out) {
  std::memset(&out, 0, sizeof(out));
  if (!(h > kFloatEps)) return false;
  const double k_r = kSqrt3 / 4.0, k_d = kSqrt3 / 8.0;
  double r[6];
  for (int i = 0; i < 6; i++) r[i] = k_r * static_cast<double>(dist[i]);
  const HexSection hs = SolveHex(r);
  const int n = static_cast<int>(hs.ring.size());
  if (n < 3) return false;  // IsValidClosedFormPrism crystal.cpp:77-79
  std::vector<std::array<float, 2>> c(n);
  for (int k = 0; k < n; k++) c[k] = {static_cast<float>(hs.ring[k].x), static_cast<float>(hs.ring[k].y)};
  const float zt = 0.5f * h, zb = -0.5f * h;

  std::vector<FaceLoop> faces(8);
  for (int s = 0; s < 8; s++) faces[s].number = s + 1;
  // basal faces
  faces[0].present = faces[1].present = hs.bounded;
  faces[0].normal[2] = 1.0f;
  faces[0].plane[2] = 1.0f;
  faces[0].plane[3] = -zt;
  faces[1].normal[2] = -1.0f;
  faces[1].plane[2] = -1.0f;
  faces[1].plane[3] = -zt;
  for (int k = 0; k < n; k++) {
    faces[0].loop.push_back({c[k][0], c[k][1], zt});
    faces[1].loop.push_back({c[n - 1 - k][0], c[n - 1 - k][1], zb});
  }
  // side faces: rectangle between ring corners k-1 and k for the k-th present side
  int k = 0;
  for (int i = 0; i < 6; i++) {
    FaceLoop& f = faces[2 + i];
    f.normal[0] = static_cast<float>(kCos6[i]);
    f.normal[1] = static_cast<float>(kSin6[i]);
    f.plane[0] = 0.5f * static_cast<float>(kCos6[i]);
    f.plane[1] = 0.5f * static_cast<float>(kSin6[i]);
    f.plane[3] = -static_cast<float>(k_d * static_cast<double>(dist[i]));
    f.present = hs.present[i];
    if (!f.present) continue;
    const auto& a = c[(k - 1 + n) % n];
    const auto& b = c[k];
    f.loop = {{a[0], a[1], zb}, {b[0], b[1], zb}, {b[0], b[1], zt}, {a[0], a[1], zt}};
    k++;
  }
  Tabulate(faces, out);
  return out.face_cnt > 0;
}

// ---------------------------------------------------------------------------------------------------
// pyramid family (Crystal::CreatePyramid crystal.cpp:379-426).  Plane set, cone slope, wedge legality and the
// basal cut follow the reference (FillHexCrystalCoef geo3d.cpp:346-512; ComputeClosedFormPyramid
// geo3d_closedform.cpp:1404-1420); the solid is assembled as a half-space intersection: every feasible
// concurrence of three planes is a vertex, a face is the CCW-sorted set of vertices on its plane.
// ---------------------------------------------------------------------------------------------------
namespace {

struct Plane {
  double a = 0, b = 0, c = 0, d = 0;
  double Eval(const double x[3]) const { return a * x[0] + b * x[1] + c * x[2] + d; }
};

bool Concurrence(const Plane& p, const Plane& q, const Plane& r, double out[3]) {
  const double det = p.a * (q.b * r.c - q.c * r.b) - p.b * (q.a * r.c - q.c * r.a) + p.c * (q.a * r.b - q.b * r.a);
  if (std::fabs(det) < 1e-9) return false;
  const double dx = -p.d, dy = -q.d, dz = -r.d;
  out[0] = (dx * (q.b * r.c - q.c * r.b) - p.b * (dy * r.c - q.c * dz) + p.c * (dy * r.b - q.b * dz)) / det;
  out[1] = (p.a * (dy * r.c - q.c * dz) - dx * (q.a * r.c - q.c * r.a) + p.c * (q.a * dz - dy * r.a)) / det;
  out[2] = (p.a * (q.b * dz - dy * r.b) - p.b * (q.a * dz - dy * r.a) + dx * (q.a * r.b - q.b * r.a)) / det;
  return true;
}

// extreme z over the feasible vertices of one cone's six planes = its natural apex
bool ConeApexZ(const Plane* cone, double tol, int sign, double& z) {
  bool found = false;
  double x[3];
  for (int i = 0; i < 6; i++)
    for (int j = i + 1; j < 6; j++)
      for (int k = j + 1; k < 6; k++) {
        if (!Concurrence(cone[i], cone[j], cone[k], x)) continue;
        bool ok = true;
        for (int m = 0; m < 6 && ok; m++) ok = cone[m].Eval(x) <= tol;
        if (!ok) continue;
        if (!found || sign * x[2] > sign * z) z = x[2];
        found = true;
      }
  return found;
}

}  // namespace

bool BuildPyramid(float wedge_u, float wedge_l, float h1, float h2, float h3, const float dist[6], HaloGeomTables& out) {
  std::memset(&out, 0, sizeof(out));
  static const int kNumber[20] = {1, 2, 3, 4, 5, 6, 7, 8, 13, 14, 15, 16, 17, 18, 23, 24, 25, 26, 27, 28};
  const bool upper = h1 > kFloatEps && wedge_u >= 0.1f && wedge_u <= 89.9f;
  const bool lower = h3 > kFloatEps && wedge_l >= 0.1f && wedge_l <= 89.9f;
  if (!upper && !lower && h2 < kFloatEps) return false;
  const double k8 = static_cast<double>(kSqrt3) / 8.0, half = 0.5 * static_cast<double>(h2);
  const double a1 = upper ? static_cast<double>(kSqrt3 / 4.0f) / std::tan(static_cast<double>(wedge_u) * static_cast<double>(kDegToRad)) : -1.0;
  const double a2 = lower ? static_cast<double>(kSqrt3 / 4.0f) / std::tan(static_cast<double>(wedge_l) * static_cast<double>(kDegToRad)) : -1.0;
  Plane raw[20], unit[20];
  bool active[20] = {};
  for (int i = 0; i < 6; i++) {
    raw[2 + i] = {0.5 * kCos6[i], 0.5 * kSin6[i], 0.0, -k8 * static_cast<double>(dist[i])};
    active[2 + i] = true;
    if (upper) {
      raw[8 + i] = {0.5 * a1 * kCos6[i], 0.5 * a1 * kSin6[i], k8, -k8 * (half + a1 * static_cast<double>(dist[i]))};
      active[8 + i] = true;
    }
    if (lower) {
      raw[14 + i] = {0.5 * a2 * kCos6[i], 0.5 * a2 * kSin6[i], -k8, -k8 * (half + a2 * static_cast<double>(dist[i]))};
      active[14 + i] = true;
    }
  }
  double scale = std::fabs(half);
  for (int s = 2; s < 20; s++) {
    if (!active[s]) continue;
    const double len = std::sqrt(raw[s].a * raw[s].a + raw[s].b * raw[s].b + raw[s].c * raw[s].c);
    unit[s] = {raw[s].a / len, raw[s].b / len, raw[s].c / len, raw[s].d / len};
    scale = std::fmax(scale, std::fabs(unit[s].d));
  }
  const double tol = 5.0 * static_cast<double>(kFloatEps) * std::fmax(scale, 1e-3);
  double z_top = half, z_bot = -half, apex = 0.0;
  if (upper) {
    if (!ConeApexZ(unit + 8, tol, +1, apex)) return false;
    z_top = half + static_cast<double>(h1) * (apex - half);
  }
  if (lower) {
    if (!ConeApexZ(unit + 14, tol, -1, apex)) return false;
    z_bot = -half + static_cast<double>(h3) * (apex + half);
  }
  raw[0] = unit[0] = {0.0, 0.0, 1.0, -z_top};
  raw[1] = unit[1] = {0.0, 0.0, -1.0, z_bot};
  active[0] = active[1] = true;

  std::vector<std::array<double, 3>> verts;
  for (int i = 0; i < 20; i++) {
    if (!active[i]) continue;
    for (int j = i + 1; j < 20; j++) {
      if (!active[j]) continue;
      for (int k = j + 1; k < 20; k++) {
        if (!active[k]) continue;
        double x[3];
        if (!Concurrence(unit[i], unit[j], unit[k], x)) continue;
        bool ok = true;
        for (int m = 0; m < 20 && ok; m++)
          if (active[m]) ok = unit[m].Eval(x) <= tol;
        if (!ok) continue;
        bool dup = false;
        for (const auto& v : verts) {
          const double dx = v[0] - x[0], dy = v[1] - x[1], dz = v[2] - x[2];
          if (std::sqrt(dx * dx + dy * dy + dz * dz) <= 2.0 * tol) {
            dup = true;
            break;
          }
        }
        if (!dup && verts.size() < 96) verts.push_back({x[0], x[1], x[2]});
      }
    }
  }
  std::vector<FaceLoop> faces(20);
  int present = 0;
  for (int s = 0; s < 20; s++) {
    FaceLoop& f = faces[s];
    f.number = kNumber[s];
    if (!active[s]) continue;
    f.plane[0] = static_cast<float>(raw[s].a);
    f.plane[1] = static_cast<float>(raw[s].b);
    f.plane[2] = static_cast<float>(raw[s].c);
    f.plane[3] = static_cast<float>(raw[s].d);
    f.normal[0] = static_cast<float>(unit[s].a);
    f.normal[1] = static_cast<float>(unit[s].b);
    f.normal[2] = static_cast<float>(unit[s].c);
    std::vector<int> on;
    for (size_t v = 0; v < verts.size(); v++)
      if (std::fabs(unit[s].Eval(verts[v].data())) <= 2.0 * tol && on.size() < HALO_MAX_FACE_VTX) on.push_back(static_cast<int>(v));
    if (on.size() < 3) continue;
    double c[3] = {0, 0, 0};
    for (int v : on)
      for (int a = 0; a < 3; a++) c[a] += verts[v][a] / static_cast<double>(on.size());
    const double n[3] = {unit[s].a, unit[s].b, unit[s].c};
    double e1[3] = {verts[on[0]][0] - c[0], verts[on[0]][1] - c[1], verts[on[0]][2] - c[2]};
    const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    if (l1 <= tol) continue;
    for (double& e : e1) e /= l1;
    const double e2[3] = {n[1] * e1[2] - n[2] * e1[1], n[2] * e1[0] - n[0] * e1[2], n[0] * e1[1] - n[1] * e1[0]};
    std::vector<std::pair<double, int>> order;
    for (size_t k = 0; k < on.size(); k++) {
      const double r[3] = {verts[on[k]][0] - c[0], verts[on[k]][1] - c[1], verts[on[k]][2] - c[2]};
      double ang = (k == 0) ? 0.0 : std::atan2(r[0] * e2[0] + r[1] * e2[1] + r[2] * e2[2], r[0] * e1[0] + r[1] * e1[1] + r[2] * e1[2]);
      if (ang < 0.0) ang += 2.0 * kPiD;
      order.emplace_back(ang, on[k]);
    }
    std::stable_sort(order.begin(), order.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
    f.present = true;
    for (const auto& o : order)
      f.loop.push_back({static_cast<float>(verts[o.second][0]), static_cast<float>(verts[o.second][1]), static_cast<float>(verts[o.second][2])});
    present++;
  }
  if (present < 4) return false;  // IsValidClosedFormPyramid crystal.cpp:93-101
  Tabulate(faces, out);
  return out.face_cnt > 0;
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

bool IsDeterministic(const HaloCrystal& c) {
  const int nh = (c.kind == HALO_CRYSTAL_PRISM) ? 1 : 3;
  for (int i = 0; i < nh; i++)
    if (c.height[i].type != HALO_DIST_NONE) return false;
  for (int i = 0; i < 6; i++)
    if (c.face_dist[i].type != HALO_DIST_NONE) return false;
  return true;
}

bool MakeShape(uint32_t seed, const HaloCrystal& c, uint64_t shape_index, HaloGeomTables& out) {
  const uint32_t lo = static_cast<uint32_t>(shape_index & 0xFFFFFFFFull);
  const uint32_t hi = static_cast<uint32_t>(shape_index >> 32);
  Pcg rng{(hi == 0u) ? (seed ^ kNonceShapeHost) : ((seed ^ kNonceShapeHost) ^ PcgHash(hi)), lo * 1000003u, 0u};
  // SyncGroupSampler (simulator.cpp:361-393): first member of a group draws, later members reuse the raw value
  int grp[9];
  float val[9];
  int cached = 0;
  auto draw = [&](int group, const HaloDist& d) {
    if (group == 0) return rng.Get(d);
    for (int i = 0; i < cached; i++)
      if (grp[i] == group) return val[i];
    const float v = rng.Get(d);
    grp[cached] = group;
    val[cached] = v;
    cached++;
    return v;
  };
  float dist[6];
  if (c.kind == HALO_CRYSTAL_PRISM) {
    const float h = std::fabs(draw(c.sync_group[0], c.height[0]));  // heights fold, distances stay signed
    for (int i = 0; i < 6; i++) dist[i] = draw(c.sync_group[3 + i], c.face_dist[i]);
    return BuildPrism(h, dist, out);
  }
  const float h1 = std::fabs(draw(c.sync_group[0], c.height[0]));
  const float h2 = std::fabs(draw(c.sync_group[1], c.height[1]));
  const float h3 = std::fabs(draw(c.sync_group[2], c.height[2]));
  for (int i = 0; i < 6; i++) dist[i] = draw(c.sync_group[3 + i], c.face_dist[i]);
  return BuildPyramid(c.wedge_upper_deg, c.wedge_lower_deg, h1, h2, h3, dist, out);
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
