// halo_geom.h — crystal geometry → kernel tables, written once for host AND device.
//
// The same functions build a crystal's ShapeDev on the host (deterministic crystals, parity tests, the C-ABI
// halo_host_*_geometry exports) and inside halo_shapegen_kernel (stochastic crystals: one thread per sampled shape,
// SURVEY §8 row f4).  Everything is fixed-size and uses only IEEE basic operations (+ - * / sqrt, fp64 solve, fp32
// tables), so with contraction off both sides produce bit-identical tables from the same scalars.  The only libm
// calls are in the scalar sampler (logf / cosf / sinf for Gauss / zigzag / Laplacian draws); the CCW ordering of pyramid
// face loops sorts on a pseudo-angle built from basic operations (same order as atan2, no libm).
//
// Reference: ComputeClosedFormPrism geo3d_closedform.cpp:1318-1407, SolveHexCrossSection :124-302,
// AdaptClosedFormPrismToCrystalGeom crystal.cpp:109-186, Crystal::PopulateFromCfGeom crystal.cpp:304-347,
// detail::BuildEntrySubTris simulator.cpp:90-129, FillHexCrystalCoef geo3d.cpp:346-512, SyncGroupSampler
// simulator.cpp:361-393.
#ifndef HALO_GEOM_H_
#define HALO_GEOM_H_

#include <math.h>
#include <stdint.h>

#include "halo_device.h"

#if defined(__HIPCC__)
#define HALO_GEOM_HD __host__ __device__ inline
#else
#define HALO_GEOM_HD inline
#endif

namespace halo {
namespace geom {

constexpr float kGeomFloatEps = 1e-5f;          // math::kFloatEps math.hpp:21
constexpr float kGeomSqrt3 = 1.73205080757f;    // math::kSqrt3
constexpr float kGeomDegToRad = 3.14159265359f / 180.0f;
constexpr double kGeomPiD = 3.14159265358979323846;
constexpr uint32_t kNonceShape = 0x6A09E667u;   // domain of the shape-scalar stream (ours: the reference draws from mt19937)

// exact 60-degree direction tables (geo3d_closedform.hpp:48-52)
// (selects, not tables: a local table indexed by a lane-dependent i is written to scratch and read back on the device)
HALO_GEOM_HD double Cos6(int i) {   // 1, 1/2, -1/2, -1, -1/2, 1/2
  const double m = (i == 0 || i == 3) ? 1.0 : 0.5;
  return (i >= 2 && i <= 4) ? -m : m;
}
HALO_GEOM_HD double Sin6(int i) {   // 0, s, s, 0, -s, -s
  const double s = 0.86602540378443864676;
  return (i == 0 || i == 3) ? 0.0 : (i < 3 ? s : -s);
}

// ---------------------------------------------------------------------------------------------------
// counter-based scalar stream (same hash as the device ray streams, pcg_shared.h:193-197)
// ---------------------------------------------------------------------------------------------------
HALO_GEOM_HD uint32_t PcgHash32(uint32_t x) {
  x = x * 747796405u + 2891336453u;
  x = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
  return (x >> 22u) ^ x;
}

struct ScalarStream {
  uint32_t seed, key, slot;
};
HALO_GEOM_HD float Uniform(ScalarStream& s) {
  const uint32_t h = PcgHash32(s.seed ^ PcgHash32(s.key + s.slot));
  s.slot++;
  return static_cast<float>(h >> 8) * (1.0f / 16777216.0f);
}
HALO_GEOM_HD float Gaussian(ScalarStream& s) {
  const float two_pi = 2.0f * 3.14159265358979323846f;
  const float u1 = fmaxf(Uniform(s), 1e-7f);
  const float u2 = Uniform(s);
  return sqrtf(-2.0f * logf(u1)) * cosf(two_pi * u2);
}
// RandomNumberGenerator::Get (math.cpp:418-444) over the PCG stream
HALO_GEOM_HD float Draw(ScalarStream& s, const HaloDist& d) {
  const float two_pi = 2.0f * 3.14159265358979323846f;
  switch (d.type) {
    case HALO_DIST_UNIFORM: return (Uniform(s) - 0.5f) * d.spread + d.center;
    case HALO_DIST_GAUSS:
    case HALO_DIST_GAUSS_LEGACY: return Gaussian(s) * d.spread + d.center;
    case HALO_DIST_ZIGZAG: return fabsf(d.spread * sinf(Uniform(s) * two_pi) + d.center);
    case HALO_DIST_LAPLACIAN: {
      const float u = Uniform(s);
      const float sgn = (u < 0.5f) ? -1.0f : 1.0f;
      const float arg = fmaxf(1.0f - 2.0f * fabsf(u - 0.5f), 1e-30f);
      return d.center - d.spread * sgn * logf(arg);
    }
    default: return d.center;
  }
}

// ---------------------------------------------------------------------------------------------------
// table emission: compact the present faces and fan-triangulate them (PopulateFromCfGeom + BuildEntrySubTris)
// ---------------------------------------------------------------------------------------------------
struct ShapeCursor {
  int fid = 0, tri = 0;
  // The builder's own copy of the emitted face rows (unit normal, plane constant), for FinalizeSlabs: on the device the record
  // lies in HBM, 1.4-4 KB from the next lane's, and every value read back from it is a dependent, uncoalesced load (the prism
  // generator sat idle 90 % of its time on those: 10.8 % VALU-active, 1.3 GB fetched by a kernel that has no input).
  float (*fn)[4] = nullptr;
  float d_last = 0.0f;   // plane constant of the face emitted last
};

// Host: zero the record.  Device: the generator's caller clears the whole pool with one memset instead of 3.7 KB of
// strided stores per thread.
template <class S>
HALO_GEOM_HD void ClearShape(S& s) {
#if defined(__HIP_DEVICE_COMPILE__)
  s.face_cnt = s.tri_cnt = s.slab_cnt = s.single_cnt = 0;   // rows beyond the counts are never read
#else
  uint32_t* w = reinterpret_cast<uint32_t*>(&s);
  for (uint32_t i = 0; i < sizeof(S) / 4u; i++) w[i] = 0u;
#endif
}

// one present face: plane = raw coefficients (a, b, c, d), normal = unit outward, loop = CCW corners seen from outside
template <class S>
HALO_GEOM_HD void EmitFace(S& s, ShapeCursor& cur, const float plane[4], const float normal[3], int number,
                           const float (*loop)[3], int nv) {
  const int fid = cur.fid;
  const float len = sqrtf(plane[0] * plane[0] + plane[1] * plane[1] + plane[2] * plane[2]);
  const float dn = (len > kGeomFloatEps) ? plane[3] / len : 0.0f;
  s.face[fid][0] = normal[0];
  s.face[fid][1] = normal[1];
  s.face[fid][2] = normal[2];
  s.face[fid][3] = dn;
  cur.d_last = dn;
  if (cur.fn != nullptr) {
    cur.fn[fid][0] = normal[0];
    cur.fn[fid][1] = normal[1];
    cur.fn[fid][2] = normal[2];
    cur.fn[fid][3] = dn;
  }
  s.face_number[fid] = static_cast<uint8_t>(number);
  for (int k = 1; k + 1 < nv && nv >= 3 && cur.tri < static_cast<int>(sizeof(s.tri_na) / 16u); k++) {
    const int t = cur.tri;
    float v[9];   // computed on the values, stored once: nothing is read back from the record
    for (int a = 0; a < 3; a++) {
      v[a] = loop[0][a];
      v[3 + a] = loop[k][a];
      v[6 + a] = loop[k + 1][a];
    }
    for (int a = 0; a < 9; a++) s.tri_v[t][a] = v[a];
    const float a[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]};
    const float b[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
    float nrm[3] = {-b[1] * a[2] + a[1] * b[2], b[0] * a[2] - a[0] * b[2], -b[0] * a[1] + a[0] * b[1]};  // Cross3 math.cpp:36
    const float mag = sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    s.tri_na[t][3] = mag / 2.0f;
    for (int c = 0; c < 3; c++) s.tri_na[t][c] = (mag > 0.0f) ? nrm[c] / mag : 0.0f;
    s.tri_face[t] = static_cast<uint8_t>(fid);
    cur.tri++;
  }
  cur.fid++;
}

// Pair up faces whose unit normals are exact negatives (see ShapeDev::slab); runs once per shape after the last EmitFace.
template <class S>
HALO_GEOM_HD void FinalizeSlabs(S& s, const ShapeCursor& cur) {
  s.face_cnt = cur.fid;
  s.tri_cnt = cur.tri;
  const float (*fn)[4] = cur.fn != nullptr ? cur.fn : s.face;   // the builder's copy where there is one (ShapeCursor)
  bool used[kMaxFaces];
  for (int i = 0; i < kMaxFaces; i++) used[i] = false;
  int ns = 0, n1 = 0;
  for (int i = 0; i < cur.fid; i++) {
    if (used[i]) continue;
    int mate = -1;
    for (int j = i + 1; j < cur.fid && mate < 0; j++)
      if (!used[j] && fn[i][0] == -fn[j][0] && fn[i][1] == -fn[j][1] && fn[i][2] == -fn[j][2]) mate = j;
    if (mate < 0) {
      s.single[n1++] = static_cast<uint8_t>(i);
      continue;
    }
    used[mate] = true;
    float* r = s.slab[ns++];
    r[0] = fn[i][0];
    r[1] = fn[i][1];
    r[2] = fn[i][2];
    r[3] = fn[i][3];
    r[4] = fn[mate][3];
    int32_t ip = i, im = mate;
    uint32_t bp, bm;
    bp = static_cast<uint32_t>(ip);
    bm = static_cast<uint32_t>(im);
    reinterpret_cast<uint32_t*>(r)[5] = bp;
    reinterpret_cast<uint32_t*>(r)[6] = bm;
    r[7] = 0.0f;
  }
  s.slab_cnt = ns;
  s.single_cnt = n1;
}

// ---------------------------------------------------------------------------------------------------
// prism
// ---------------------------------------------------------------------------------------------------
struct Pt2 {
  double x, y;
};

// intersection of half-plane boundaries i and j (Cramer, geo3d_closedform.cpp:27-35)
HALO_GEOM_HD bool Meet(int i, int j, const double r[6], Pt2& out) {
  const double det = Cos6(i) * Sin6(j) - Sin6(i) * Cos6(j);
  if (det == 0.0) return false;
  out.x = (r[i] * Sin6(j) - r[j] * Sin6(i)) / det;
  out.y = (Cos6(i) * r[j] - Cos6(j) * r[i]) / det;
  return true;
}

struct HexSection {
  Pt2 ring[6];      // CCW corners, one per adjacent pair of present sides
  int n = 0;
  bool present[6] = {false, false, false, false, false, false};
  bool bounded = false;
};

// 2-D intersection of the six half-planes cos(i*60)x + sin(i*60)y <= r[i] (SolveHexCrossSection,
// geo3d_closedform.cpp:124-302): enumerate non-parallel pairs, keep feasible corners, dedupe within
// tol = 5*eps*max|r|, a side is present iff >= 2 corners sit on it, then walk present sides in order.
HALO_GEOM_HD void SolveHex(const double r[6], HexSection& hs) {
  double scale = 0.0;
  for (int i = 0; i < 6; i++) scale = fmax(scale, fabs(r[i]));
  const double tol = 5.0 * static_cast<double>(kGeomFloatEps) * scale;
  Pt2 cand[12];
  int nc = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i + 1; j < 6; j++) {
      if (j == i + 3) continue;
      Pt2 q{0.0, 0.0};
      Meet(i, j, r, q);
      bool ok = true;
      for (int m = 0; m < 6 && ok; m++)
        if (m != i && m != j && Cos6(m) * q.x + Sin6(m) * q.y > r[m] + tol) ok = false;
      if (!ok) continue;
      bool dup = false;
      for (int c = 0; c < nc; c++)
        if (sqrt((cand[c].x - q.x) * (cand[c].x - q.x) + (cand[c].y - q.y) * (cand[c].y - q.y)) <= tol) {
          dup = true;
          break;
        }
      if (!dup && nc < 12) cand[nc++] = q;
    }
  int sides[6];
  int n = 0;
  for (int i = 0; i < 6; i++) {
    int on = 0;
    for (int c = 0; c < nc; c++)
      if (fabs(Cos6(i) * cand[c].x + Sin6(i) * cand[c].y - r[i]) <= tol) on++;
    hs.present[i] = on >= 2;
    if (hs.present[i]) sides[n++] = i;
  }
  bool opposite_adjacent = false;
  for (int k = 0; k < n; k++) {
    const int dlt = sides[k] - sides[(k + 1) % n];
    if (dlt == 3 || dlt == -3) opposite_adjacent = true;
  }
  hs.bounded = n >= 3 && !opposite_adjacent;
  hs.n = 0;
  if (!hs.bounded) return;
  for (int k = 0; k < n; k++) {
    Pt2 q{0.0, 0.0};
    Meet(sides[k], sides[(k + 1) % n], r, q);
    hs.ring[hs.n++] = q;
  }
}

// ComputeClosedFormPrism + AdaptClosedFormPrismToCrystalGeom. false = empty crystal (counts stay 0).
template <class S>
HALO_GEOM_HD bool BuildPrismShape(float h, const float dist[6], S& out) {
  ClearShape(out);
  if (!(h > kGeomFloatEps)) return false;
  const double k_r = kGeomSqrt3 / 4.0, k_d = kGeomSqrt3 / 8.0;
  double r[6];
  for (int i = 0; i < 6; i++) r[i] = k_r * static_cast<double>(dist[i]);
  HexSection hs;
  SolveHex(r, hs);
  const int n = hs.n;
  if (n < 3) return false;  // IsValidClosedFormPrism crystal.cpp:77-79
  float c[6][2];
  for (int k = 0; k < n; k++) {
    c[k][0] = static_cast<float>(hs.ring[k].x);
    c[k][1] = static_cast<float>(hs.ring[k].y);
  }
  const float zt = 0.5f * h, zb = -0.5f * h;
  ShapeCursor cur;
  float fn_rows[kMaxFaces][4];
  cur.fn = fn_rows;
  float loop[HALO_MAX_FACE_VTX][3];
  if (hs.bounded) {  // basal faces (numbers 1, 2)
    const float plane_t[4] = {0.0f, 0.0f, 1.0f, -zt}, nrm_t[3] = {0.0f, 0.0f, 1.0f};
    for (int k = 0; k < n; k++) {
      loop[k][0] = c[k][0];
      loop[k][1] = c[k][1];
      loop[k][2] = zt;
    }
    EmitFace(out, cur, plane_t, nrm_t, 1, loop, n);
    const float plane_b[4] = {0.0f, 0.0f, -1.0f, -zt}, nrm_b[3] = {0.0f, 0.0f, -1.0f};
    for (int k = 0; k < n; k++) {
      loop[k][0] = c[n - 1 - k][0];
      loop[k][1] = c[n - 1 - k][1];
      loop[k][2] = zb;
    }
    EmitFace(out, cur, plane_b, nrm_b, 2, loop, n);
  }
  // side faces (numbers 3..8): rectangle between ring corners k-1 and k for the k-th present side
  int k = 0;
  for (int i = 0; i < 6; i++) {
    if (!hs.present[i]) continue;
    const float nrm[3] = {static_cast<float>(Cos6(i)), static_cast<float>(Sin6(i)), 0.0f};
    const float plane[4] = {0.5f * static_cast<float>(Cos6(i)), 0.5f * static_cast<float>(Sin6(i)), 0.0f,
                            -static_cast<float>(k_d * static_cast<double>(dist[i]))};
    const float* a = c[(k - 1 + n) % n];
    const float* b = c[k];
    loop[0][0] = a[0]; loop[0][1] = a[1]; loop[0][2] = zb;
    loop[1][0] = b[0]; loop[1][1] = b[1]; loop[1][2] = zb;
    loop[2][0] = b[0]; loop[2][1] = b[1]; loop[2][2] = zt;
    loop[3][0] = a[0]; loop[3][1] = a[1]; loop[3][2] = zt;
    EmitFace(out, cur, plane, nrm, 3 + i, loop, 4);
    k++;
  }
  FinalizeSlabs(out, cur);
  return out.face_cnt > 0;
}

// ---------------------------------------------------------------------------------------------------
// pyramid family (Crystal::CreatePyramid crystal.cpp:379-426).  Plane set, cone slope, wedge legality and the
// basal cut follow the reference (FillHexCrystalCoef geo3d.cpp:346-512; ComputeClosedFormPyramid
// geo3d_closedform.cpp:1404-1420); the solid is assembled as a half-space intersection: every feasible
// concurrence of three planes is a vertex, a face is the CCW-sorted set of vertices on its plane.
// ---------------------------------------------------------------------------------------------------
struct Plane3 {
  double a, b, c, d;
};
// Round 4: plane evaluations and the 3 x 3 solve are explicit fma chains.  This file compiles with contraction OFF on host and device so
// that both round alike; an explicit fma() is one IEEE operation on either side (v_fma_f64; vfmadd or libm's fma on the host), so the
// tables stay bit-equal — and the device generator, whose time is the twenty-plane feasibility scan of ~100 candidate vertices in fp64,
// issues three operations per plane where it issued six.
HALO_GEOM_HD double EvalPlane4(double a, double b, double c, double d, double x0, double x1, double x2) { return fma(a, x0, fma(b, x1, fma(c, x2, d))); }
HALO_GEOM_HD double EvalPlane(const Plane3& p, const double x[3]) { return EvalPlane4(p.a, p.b, p.c, p.d, x[0], x[1], x[2]); }
HALO_GEOM_HD double Minor2(double a, double b, double c, double d) { return fma(a, b, -(c * d)); }   // a b - c d

HALO_GEOM_HD bool Concurrence(const Plane3& p, const Plane3& q, const Plane3& r, double out[3]) {
  const double m_bc = Minor2(q.b, r.c, q.c, r.b), m_ac = Minor2(q.a, r.c, q.c, r.a), m_ab = Minor2(q.a, r.b, q.b, r.a);
  const double det = fma(p.a, m_bc, fma(-p.b, m_ac, p.c * m_ab));
  if (fabs(det) < 1e-9) return false;
  const double dx = -p.d, dy = -q.d, dz = -r.d;
  const double inv = 1.0 / det;   // one fp64 division per concurrence (a long sequence on the device), not three
  const double n_yc = Minor2(dy, r.c, q.c, dz), n_yb = Minor2(dy, r.b, q.b, dz), n_az = Minor2(q.a, dz, dy, r.a), n_bz = Minor2(q.b, dz, dy, r.b);
  out[0] = fma(dx, m_bc, fma(-p.b, n_yc, p.c * n_yb)) * inv;
  out[1] = fma(p.a, n_yc, fma(-dx, m_ac, p.c * n_az)) * inv;
  out[2] = fma(p.a, n_bz, fma(-p.b, n_az, dx * m_ab)) * inv;
  return true;
}

// extreme z over the feasible vertices of one cone's six planes = its natural apex
HALO_GEOM_HD bool ConeApexZ(const Plane3* cone, double tol, int sign, double& z) {
  bool found = false;
  double x[3];
  for (int i = 0; i < 6; i++)
    for (int j = i + 1; j < 6; j++)
      for (int k = j + 1; k < 6; k++) {
        if (!Concurrence(cone[i], cone[j], cone[k], x)) continue;
        bool ok = true;
        for (int m = 0; m < 6 && ok; m++) ok = EvalPlane(cone[m], x) <= tol;
        if (!ok) continue;
        if (!found || sign * x[2] > sign * z) z = x[2];
        found = true;
      }
  return found;
}

HALO_GEOM_HD double PyrMergeRadius(double max_abs_dist) {   // one definition for the serial builder and the team builder
  return 5.0 * static_cast<double>(kGeomFloatEps) * (0.25 * 1.7320508075688772935) * fmax(max_abs_dist, 1e-3);
}
constexpr int kPyrMaxVerts = 40;   // a hexagonal prism capped by two truncated hexagonal pyramids has 24 corners; the reference's pools peak at 24

// Plane of slot s (2..7 prism sides, 8..13 upper cone, 14..19 lower cone) as FillHexCrystalCoef states it (geo3d.cpp:346-512);
// a1 / a2 = cone slopes, half = h2 / 2, k8 = sqrt3 / 8.  One definition for the serial builder and the team builder.
HALO_GEOM_HD Plane3 PyrRawPlaneOne(int s, double a1, double a2, double half, double k8, float dist_i) {   // dist_i = dist[(s - 2) % 6]
  if (s < 8) {
    const int i = s - 2;
    return Plane3{0.5 * Cos6(i), 0.5 * Sin6(i), 0.0, -k8 * static_cast<double>(dist_i)};
  }
  if (s < 14) {
    const int i = s - 8;
    return Plane3{0.5 * a1 * Cos6(i), 0.5 * a1 * Sin6(i), k8, -k8 * (half + a1 * static_cast<double>(dist_i))};
  }
  const int i = s - 14;
  return Plane3{0.5 * a2 * Cos6(i), 0.5 * a2 * Sin6(i), -k8, -k8 * (half + a2 * static_cast<double>(dist_i))};
}
HALO_GEOM_HD Plane3 PyrRawPlane(int s, double a1, double a2, double half, double k8, const float dist[6]) {
  return PyrRawPlaneOne(s, a1, a2, half, k8, dist[(s - 2) % 6]);
}
HALO_GEOM_HD Plane3 PyrUnitPlane(const Plane3& raw) {
  const double inv = 1.0 / sqrt(raw.a * raw.a + raw.b * raw.b + raw.c * raw.c);
  return Plane3{raw.a * inv, raw.b * inv, raw.c * inv, raw.d * inv};
}
constexpr int kPyrFaceNumber[20] = {1, 2, 3, 4, 5, 6, 7, 8, 13, 14, 15, 16, 17, 18, 23, 24, 25, 26, 27, 28};

// The vertices of one face, given as indices `on[0..cnt)` into verts, put into CCW order seen from outside (in place; the
// loop starts at the face's first vertex).  Returns cnt, or 0 when the face degenerates to a point.  `ang` is caller-provided
// scratch of cnt doubles.  One definition for the serial builder and the team builder.
template <class IndexT>
HALO_GEOM_HD int PyrOrderFace(const double (*verts)[3], IndexT* on, int cnt, const Plane3& unit, double tol, double* ang) {
  // centroid of the face's vertices: only the angular ORDER around it is used below (the loop starts at the face's first
  // vertex whatever its value), so it is summed first and divided once — three fp64 divisions per face, not per vertex
  double c[3] = {0, 0, 0};
  for (int q = 0; q < cnt; q++)
    for (int a = 0; a < 3; a++) c[a] += verts[on[q]][a];
  for (int a = 0; a < 3; a++) c[a] /= static_cast<double>(cnt);
  const double n[3] = {unit.a, unit.b, unit.c};
  double e1[3] = {verts[on[0]][0] - c[0], verts[on[0]][1] - c[1], verts[on[0]][2] - c[2]};
  const double l1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
  if (l1 <= tol) return 0;
  for (int a = 0; a < 3; a++) e1[a] /= l1;
  const double e2[3] = {n[1] * e1[2] - n[2] * e1[1], n[2] * e1[0] - n[0] * e1[2], n[0] * e1[1] - n[1] * e1[0]};
  for (int q = 0; q < cnt; q++) {
    const double r[3] = {verts[on[q]][0] - c[0], verts[on[q]][1] - c[1], verts[on[q]][2] - c[2]};
    // sort key: a pseudo-angle that grows with atan2(y, x) mapped to [0, 2 pi) — t = |y| / (|x| + |y|) per quadrant, in
    // [0, 4) — so the order is the CCW order an atan2 would give (the vertices of a face are at least a merge radius, ~2e-5 of the
    // crystal's width, apart after the duplicate filter: far beyond any rounding of either key) at the price of one division
    double a = 0.0;
    if (q != 0) {
      const double y = r[0] * e2[0] + r[1] * e2[1] + r[2] * e2[2], x = r[0] * e1[0] + r[1] * e1[1] + r[2] * e1[2];
      const double ax = fabs(x), ay = fabs(y);
      const double t = (ax + ay > 0.0) ? ay / (ax + ay) : 0.0;
      a = (y >= 0.0) ? (x >= 0.0 ? t : 2.0 - t) : (x < 0.0 ? 2.0 + t : 4.0 - t);
    }
    ang[q] = a;
  }
  for (int q = 1; q < cnt; q++) {  // stable insertion sort by angle
    const double ka = ang[q];
    const IndexT kv = on[q];
    int p = q - 1;
    while (p >= 0 && ang[p] > ka) {
      ang[p + 1] = ang[p];
      on[p + 1] = on[p];
      p--;
    }
    ang[p + 1] = ka;
    on[p + 1] = kv;
  }
  return cnt;
}

// cot_u / cot_l = sqrt3/4 / tan(wedge) for a legal wedge, negative = that cone absent (the caller evaluates tan once)
HALO_GEOM_HD bool BuildPyramidShape(double cot_u, double cot_l, float h1, float h2, float h3, const float dist[6], ShapeDev& out) {
  ClearShape(out);
  const bool upper = h1 > kGeomFloatEps && cot_u >= 0.0;
  const bool lower = h3 > kGeomFloatEps && cot_l >= 0.0;
  if (!upper && !lower && h2 < kGeomFloatEps) return false;
  const double k8 = static_cast<double>(kGeomSqrt3) / 8.0, half = 0.5 * static_cast<double>(h2);
  const double a1 = upper ? cot_u : -1.0;
  const double a2 = lower ? cot_l : -1.0;
  Plane3 raw[20], unit[20];
  bool active[20];
  for (int s = 0; s < 20; s++) {
    active[s] = false;
    raw[s] = unit[s] = Plane3{0.0, 0.0, 0.0, 0.0};
  }
  for (int s = 2; s < 20; s++) {
    active[s] = (s < 8) || (s < 14 ? upper : lower);
    if (active[s]) raw[s] = PyrRawPlane(s, a1, a2, half, k8, dist);
  }
  double scale = fabs(half);
  for (int s = 2; s < 20; s++) {
    if (!active[s]) continue;
    unit[s] = PyrUnitPlane(raw[s]);
    scale = fmax(scale, fabs(unit[s].d));
  }
  const double tol = 5.0 * static_cast<double>(kGeomFloatEps) * fmax(scale, 1e-3);
  // The duplicate radius is LATERAL (geo3d_closedform.cpp:77-96 GapToleranceForScale, :393 LateralMergeTol, :702-710 InsertOrFindVertex, :927
  // kApexMergeTol): 5 * kFloatEps x the scale of the m = 0 cross section, (sqrt3/4) max |dist_i| — the ruler the reference applies at the apex
  // and the upper bound of the rulers it applies at every other inset.  (Until round 5: 2 tol, i.e. scaled by the largest plane constant — a
  // steep wedge puts a cone plane's |d| far above the crystal's width, and the radius swallowed slivers the reference resolves:
  // test_closed_form_pyramid.cpp:1579-1663 t10298 came out with 11 faces where the reference demands >= 12.)
  double max_dist = 0.0;
  for (int i = 0; i < 6; i++) max_dist = fmax(max_dist, fabs(static_cast<double>(dist[i])));
  const double merge = PyrMergeRadius(max_dist);
  double z_top = half, z_bot = -half, apex = 0.0;
  if (upper) {
    if (!ConeApexZ(unit + 8, tol, +1, apex)) return false;
    z_top = half + static_cast<double>(h1) * (apex - half);
  }
  if (lower) {
    if (!ConeApexZ(unit + 14, tol, -1, apex)) return false;
    z_bot = -half + static_cast<double>(h3) * (apex + half);
  }
  raw[0] = unit[0] = Plane3{0.0, 0.0, 1.0, -z_top};
  raw[1] = unit[1] = Plane3{0.0, 0.0, -1.0, z_bot};
  active[0] = active[1] = true;

  double verts[kPyrMaxVerts][3];
  int nv = 0;
  // Vertices = feasible concurrences of three planes.  Of the C(20,3) = 1140 triples only ~100 can meet in a point OF THE
  // SOLID, and which ones follows from the plane set alone (cone slope a > 0):
  //   * above the shoulder (z > h2/2) upper-cone plane i is tighter than prism plane i (its bound on cos x + sin y shrinks with
  //     z) and the lower cone is looser still, so only upper-cone planes and the top basal plane can be tight there; mirrored
  //     below -h2/2; strictly between the shoulders only the vertical prism planes are tight, and three vertical planes
  //     never meet in a point;
  //   * AT a shoulder, cone plane i and prism plane i share the horizontal line of side i, so a corner of the cross-section
  //     there is the concurrence of prism i, prism j and the cone planes i, j.
  // Hence every vertex is found by one of: three planes of the same cone (2 x 20 triples); a basal plane with two planes of
  // its cone, or with two prism planes when that cone is absent (2 x 15); two prism planes i < j with cone plane i (2 x 15).
  // The triples are visited in lexicographic order and each list above contains the lexicographically FIRST triple of every
  // vertex, so the kept vertices, their order and their coordinates are those of the exhaustive enumeration this replaces
  // (which spent 1140 solves and feasibility scans per crystal to find them; the duplicate filter stays for the coincident
  // triples that remain).
  double pre[20][4];
  uint32_t act_mask = 0u;
  for (int s = 0; s < 20; s++) {
    pre[s][0] = unit[s].a;
    pre[s][1] = unit[s].b;
    pre[s][2] = unit[s].c;
    pre[s][3] = unit[s].d;
    if (active[s]) act_mask |= 1u << s;
  }
  // Which planes a kept vertex lies on: the planes that pass through it EXACTLY (|distance| <= 1e-9 of the crystal's size: the concurrence is
  // computed in double) — its own three and whatever else meets there — united over the candidates the duplicate filter folds into it.
  // (Until round 3 a face claimed every kept vertex within 2 tol of its plane: a corner 6e-5 off a neighbouring plane — an apex ridge 1.2e-4
  // long — joined that face too and tilted its fan off the plane; tables that are no polytope, and the reference's next-face strategies part
  // ways on those: DESIGN 5, tests/test_gpu_fuzz.py.)
  uint32_t vmask[kPyrMaxVerts];
  const double tight = 1e-9 * fmax(scale, 1e-3);
  auto try_triple = [&](int i, int j, int k) {
    double x[3];
    if (!Concurrence(unit[i], unit[j], unit[k], x)) return;
    bool ok = true;
    uint32_t mk = (1u << i) | (1u << j) | (1u << k);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int m = 0; m < 20; m++)   // EvalPlane(unit[m], x) <= tol
      if ((act_mask >> m) & 1u) {
        const double ev = EvalPlane4(pre[m][0], pre[m][1], pre[m][2], pre[m][3], x[0], x[1], x[2]);
        ok = ok && (ev <= tight);
        if (fabs(ev) <= tight) mk |= 1u << m;
      }
    if (!ok) return;
    for (int v = 0; v < nv; v++) {
      const double dx = verts[v][0] - x[0], dy = verts[v][1] - x[1], dz = verts[v][2] - x[2];
      if (fabs(dx) > 2.0 * merge || fabs(dy) > 2.0 * merge || fabs(dz) > 2.0 * merge) continue;   // farther than the radius for certain: no sqrt
      if (sqrt(dx * dx + dy * dy + dz * dz) <= merge) {
        vmask[v] |= mk;
        return;   // duplicate
      }
    }
    if (nv < kPyrMaxVerts) {
      verts[nv][0] = x[0];
      verts[nv][1] = x[1];
      verts[nv][2] = x[2];
      vmask[nv] = mk;
      nv++;
    }
  };
  // The argument above is about the exact solid.  The feasibility test has a tolerance, and where two rings of corners come within a few
  // tolerances of each other (a cap a ten-thousandth of the way to its apex, a side face about to vanish) triples that meet OUTSIDE the
  // exact solid pass it as well; the exhaustive enumeration meets them — early, and they then absorb the true corners on both sides as
  // duplicates — the short lists do not, and keep two rings whose corners each face then claims in part: a table that is no polytope
  // (fan triangles != 2 V - 4).  So the short lists are checked by that count and the exhaustive enumeration — the definition — runs
  // when it fails (found by tests/test_gpu_fuzz.py, seed 488: one crystal in ~2500 of a recipe whose cap height crosses zero;
  // tools/pyr_topology_scan.py: 11 of 30000 ordinary parameter draws disagreed with the exhaustive builder before, 1 now — and there the
  // exhaustive table is no polytope itself).
  int on[20][HALO_MAX_FACE_VTX];
  int on_n[20];
  int present = 0;
  bool polytope = false;
  for (int pass = 0; pass < 2; pass++) {
  nv = 0;
  if (pass == 0) {
  for (int b = 0; b < 2; b++) {   // i = 0, 1: a basal plane with two planes of its cone (or two prism planes without one)
    const int lo = (b == 0) ? (upper ? 8 : 2) : (lower ? 14 : 2);
    for (int j = lo; j < lo + 6; j++)
      for (int k = j + 1; k < lo + 6; k++) try_triple(b, j, k);
  }
  for (int i = 2; i < 8; i++)     // i = prism plane: with a later prism plane and cone plane i of either cone
    for (int j = i + 1; j < 8; j++) {
      if (upper) try_triple(i, j, 8 + (i - 2));
      if (lower) try_triple(i, j, 14 + (i - 2));
    }
  for (int c = 0; c < 2; c++) {   // i in a cone: three planes of the same cone
    if (!(c == 0 ? upper : lower)) continue;
    const int lo = (c == 0) ? 8 : 14;
    for (int i = lo; i < lo + 6; i++)
      for (int j = i + 1; j < lo + 6; j++)
        for (int k = j + 1; k < lo + 6; k++) try_triple(i, j, k);
  }
  } else {
    for (int i = 0; i < 20; i++) {
      if (!active[i]) continue;
      for (int j = i + 1; j < 20; j++) {
        if (!active[j]) continue;
        for (int k = j + 1; k < 20; k++)
          if (active[k]) try_triple(i, j, k);
      }
    }
  }
  // a face is present with >= 3 vertices on its plane; the crystal needs >= 4 present faces
  // (IsValidClosedFormPyramid crystal.cpp:93-101), so the loops are ordered in a first pass and emitted in a second
  present = 0;
  // membership of every vertex in every plane, vertex-major: one load of the vertex, the 20 planes from their register copies
  // (same expression as EvalPlane, same values; each face's list comes out in ascending vertex order as before)
  int member_cnt[20];
  {
    int cnt_reg[20];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < 20; s++) cnt_reg[s] = 0;
    for (int v = 0; v < nv; v++) {
      const double x0 = verts[v][0], x1 = verts[v][1], x2 = verts[v][2];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int s = 0; s < 20; s++)
        if (((act_mask >> s) & 1u) && ((vmask[v] >> s) & 1u) && cnt_reg[s] < HALO_MAX_FACE_VTX)
          on[s][cnt_reg[s]++] = v;
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < 20; s++) member_cnt[s] = cnt_reg[s];
  }
  for (int s = 0; s < 20; s++) {
    on_n[s] = 0;
    if (!active[s]) continue;
    const int cnt = member_cnt[s];
    if (cnt < 3) continue;
    double ang[HALO_MAX_FACE_VTX];
    if (PyrOrderFace(verts, on[s], cnt, unit[s], tight, ang) == 0) continue;   // (only a face whose vertices all coincide is no face: a sliver a few 1e-5 across is one, t10298)
    on_n[s] = cnt;
    present++;
  }
  int tris = 0;
  for (int s = 0; s < 20; s++) tris += on_n[s] > 0 ? on_n[s] - 2 : 0;
  polytope = tris == 2 * nv - 4 && present >= 4;
  if (polytope) break;   // a polytope: the short lists found it
  }
  // A table that fails Euler's count even after the exhaustive enumeration (2 of 10^4 deliberately degenerate draws: a concurrence whose
  // determinant sits just above the 1e-9 gate) is REFUSED: the empty crystal, whose rays carry no weight — what the reference does with a
  // sample its builder rejects (MakeCrystal, simulator.cpp:448; crystal.cpp:77-79) — rather than a body the next-face search can leave
  // through a gap.  Host, device team kernel and oracle refuse the same samples.
  if (!polytope) return false;
  ShapeCursor cur;
  float fn_rows[kMaxFaces][4];
  cur.fn = fn_rows;
  float loop[HALO_MAX_FACE_VTX][3];
  for (int s = 0; s < 20; s++) {
    if (on_n[s] == 0) continue;
    const float plane[4] = {static_cast<float>(raw[s].a), static_cast<float>(raw[s].b), static_cast<float>(raw[s].c), static_cast<float>(raw[s].d)};
    const float nrm[3] = {static_cast<float>(unit[s].a), static_cast<float>(unit[s].b), static_cast<float>(unit[s].c)};
    for (int q = 0; q < on_n[s]; q++)
      for (int a = 0; a < 3; a++) loop[q][a] = static_cast<float>(verts[on[s][q]][a]);
    EmitFace(out, cur, plane, nrm, kPyrFaceNumber[s], loop, on_n[s]);
  }
  FinalizeSlabs(out, cur);
  return out.face_cnt > 0;
}

// ---------------------------------------------------------------------------------------------------
// one sampled crystal instance (MakeCrystal simulator.cpp:448 with SyncGroupSampler :361-393)
// ---------------------------------------------------------------------------------------------------
struct CrystalRecipe {   // HaloCrystal with the wedge trig already evaluated (host, once per dispatch)
  HaloCrystal c;
  double cot_u, cot_l;   // sqrt3/4 / tan(wedge); negative = illegal wedge (cone absent)
  // draw plan (PlanShapeScalar, filled by host::MakeRecipe): scalar q of [h0, h1, h2, d0..d5] is the draw of distribution
  // plan_src[q] (0..2 heights, 3..8 face distances) taken at stream slot plan_slot[q]; 0xFF = the crystal kind has no scalar q
  uint8_t plan_slot[9], plan_src[9];
};

HALO_GEOM_HD bool BuildPyramidDispatch(const CrystalRecipe& rc, const float sc[9], const float dist[6], ShapeDev& out) {
  return BuildPyramidShape(rc.cot_u, rc.cot_l, fabsf(sc[0]), fabsf(sc[1]), fabsf(sc[2]), dist, out);
}
HALO_GEOM_HD bool BuildPyramidDispatch(const CrystalRecipe&, const float*, const float*, ShapePrism& out) {
  ClearShape(out);
  return false;
}

// The shape scalars of crystal instance `shape_index`, slot order [h0, h1, h2, d0..d5] (prism: h0 only), raw draws (heights
// are folded with fabs by the callers).  Members of a sync group share the draw of the group's first member.
HALO_GEOM_HD void DrawShapeScalars(uint32_t seed, const CrystalRecipe& rc, uint64_t shape_index, float sc[9]) {
  const HaloCrystal& c = rc.c;
  const uint32_t lo = static_cast<uint32_t>(shape_index & 0xFFFFFFFFull);
  const uint32_t hi = static_cast<uint32_t>(shape_index >> 32);
  ScalarStream rng{(hi == 0u) ? (seed ^ kNonceShape) : ((seed ^ kNonceShape) ^ PcgHash32(hi)), lo * 1000003u, 0u};
  // first member of a sync group draws, later members reuse the raw value
  int grp[9];
  float val[9];
  int cached = 0;
  for (int i = 0; i < 9; i++) sc[i] = 0.0f;
  const int n_h = (c.kind == HALO_CRYSTAL_PRISM) ? 1 : 3;
  for (int i = 0; i < n_h + 6; i++) {
    const int slot = (i < n_h) ? i : 3 + (i - n_h);
    const HaloDist& d = (i < n_h) ? c.height[i] : c.face_dist[i - n_h];
    const int group = c.sync_group[slot];
    float v = 0.0f;
    bool have = false;
    if (group != 0)
      for (int q = 0; q < cached; q++)
        if (grp[q] == group) {
          v = val[q];
          have = true;
          break;
        }
    if (!have) {
      v = Draw(rng, d);
      if (group != 0) {
        grp[cached] = group;
        val[cached] = v;
        cached++;
      }
    }
    sc[slot] = v;
  }
}

// RNG slots one draw of `d` consumes (Uniform: 1, Box-Muller Gaussian: 2; see Draw)
HALO_GEOM_HD uint32_t DrawSlots(const HaloDist& d) {
  switch (d.type) {
    case HALO_DIST_UNIFORM:
    case HALO_DIST_ZIGZAG:
    case HALO_DIST_LAPLACIAN: return 1u;
    case HALO_DIST_GAUSS:
    case HALO_DIST_GAUSS_LEGACY: return 2u;
    default: return 0u;
  }
}
// Where ONE of the nine scalars of DrawShapeScalars comes from (slot `want` of [h0, h1, h2, d0..d5]): walks the same draw
// sequence without drawing — `src` = the distribution whose draw supplies it (itself, or the first member in draw order of its
// sync group), `at` = the stream slot that draw starts at.  False for a slot the crystal kind does not have.  The walk depends
// on the crystal description alone, so the host does it once per dispatch (CrystalRecipe::plan_*) and the team generators draw
// their scalars side by side, one lane each, straight from the plan.
HALO_GEOM_HD bool PlanShapeScalar(const HaloCrystal& c, int want, uint32_t& at, int& src) {
  const int n_h = (c.kind == HALO_CRYSTAL_PRISM) ? 1 : 3;
  // the slot whose draw `want` takes: itself, or the first member (in draw order) of its sync group
  int target = -1;
  const int want_group = (want >= 0 && want < 9) ? c.sync_group[want] : 0;
  bool want_exists = false;
  for (int i = 0; i < n_h + 6; i++) {
    const int slot = (i < n_h) ? i : 3 + (i - n_h);
    if (slot == want) want_exists = true;
    if (target < 0 && (slot == want || (want_group != 0 && c.sync_group[slot] == want_group))) target = slot;
  }
  if (!want_exists || target < 0) return false;
  int seen_groups[9];
  int n_seen = 0;
  uint32_t pos = 0u;
  for (int i = 0; i < n_h + 6; i++) {
    const int slot = (i < n_h) ? i : 3 + (i - n_h);
    const HaloDist& d = (i < n_h) ? c.height[i] : c.face_dist[i - n_h];
    const int group = c.sync_group[slot];
    bool have = false;
    if (group != 0)
      for (int q = 0; q < n_seen; q++) have = have || (seen_groups[q] == group);
    if (have) continue;                 // a later member of a group: reuses, draws nothing
    if (slot == target) {
      at = pos;
      src = (i < n_h) ? i : 3 + (i - n_h);
      return true;
    }
    pos += DrawSlots(d);
    if (group != 0) seen_groups[n_seen++] = group;
  }
  return false;
}
HALO_GEOM_HD void FillDrawPlan(CrystalRecipe& rc) {
  for (int q = 0; q < 9; q++) {
    uint32_t at = 0u;
    int src = 0;
    const bool ok = PlanShapeScalar(rc.c, q, at, src);
    rc.plan_slot[q] = ok ? static_cast<uint8_t>(at) : static_cast<uint8_t>(0xFF);
    rc.plan_src[q] = ok ? static_cast<uint8_t>(src) : static_cast<uint8_t>(0);
  }
}
// scalar `want` of crystal instance `shape_index` by the recipe's plan (the value DrawShapeScalars puts in sc[want])
HALO_GEOM_HD float DrawShapeScalarOne(uint32_t seed, const CrystalRecipe& rc, uint64_t shape_index, int want) {
  if (want < 0 || want >= 9 || rc.plan_slot[want] == 0xFF) return 0.0f;
  const uint32_t lo = static_cast<uint32_t>(shape_index & 0xFFFFFFFFull);
  const uint32_t hi = static_cast<uint32_t>(shape_index >> 32);
  ScalarStream rng{(hi == 0u) ? (seed ^ kNonceShape) : ((seed ^ kNonceShape) ^ PcgHash32(hi)), lo * 1000003u, rc.plan_slot[want]};
  const int src = rc.plan_src[want];
  return Draw(rng, src < 3 ? rc.c.height[src] : rc.c.face_dist[src - 3]);
}

// S = ShapeDev (any crystal) or ShapePrism (prisms only: a pyramid recipe yields the empty shape)
template <class S>
HALO_GEOM_HD bool MakeShapeDev(uint32_t seed, const CrystalRecipe& rc, uint64_t shape_index, S& out) {
  const HaloCrystal& c = rc.c;
  float sc[9];
  DrawShapeScalars(seed, rc, shape_index, sc);
  float dist[6];
  for (int i = 0; i < 6; i++) dist[i] = sc[3 + i];
  if (c.kind == HALO_CRYSTAL_PRISM) return BuildPrismShape(fabsf(sc[0]), dist, out);  // heights fold, distances stay signed
  return BuildPyramidDispatch(rc, sc, dist, out);
}

}  // namespace geom
}  // namespace halo

#endif
