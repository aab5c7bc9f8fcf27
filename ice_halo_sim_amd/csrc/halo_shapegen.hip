// halo_shapegen.hip — device crystal generator: one thread samples one crystal instance and writes
// its kernel tables.  The geometry code is halo_geom.h, the same functions the host uses, compiled here with
// -ffp-contract=off so that identical scalars give bit-identical tables on both sides.
//
// Replaces, for stochastic crystals, the host loop MakeCrystal (simulator.cpp:448) → closed-form geometry
// (geo3d_closedform.cpp:124-302,1318-1407) → PopulateFromCfGeom (crystal.cpp:304-347) and the upload of its result;
// the reference's device counterpart is the per-ray geometry stage of its CUDA backend (cuda_trace_backend.cu:2577-2800).
// geom_clock consecutive rays share shape k (legacy semantics, simulator.cpp:1244-1275), so the pool has n/geom_clock
// entries and the trace kernel reads shape tid/geom_clock.
#include <hip/hip_runtime.h>

#include "halo_geom.h"

namespace halo {

constexpr int kGenBlock = 64;

// S = ShapeDev (any crystal; 4.1 KB records) or ShapePrism (prism pools; 1.4 KB records).  Every record gets its counts
// written (an invalid draw leaves them 0 = empty crystal) and rows beyond the counts are never read, so the pool needs no
// clearing.
#ifndef HALO_GEN_WAVES
#define HALO_GEN_WAVES 2
#endif
#ifndef HALO_GEN_WAVES_PRISM
#define HALO_GEN_WAVES_PRISM 1
#endif
template <class S>
// (waves per SIMD the general builder is compiled for: measured below; the pyramid feasibility scans keep the 20 fp64 planes
// in registers)
__global__ void __launch_bounds__(kGenBlock, (sizeof(S) == sizeof(ShapeDev) ? HALO_GEN_WAVES : HALO_GEN_WAVES_PRISM)) halo_shapegen_kernel(S* __restrict__ pool, uint32_t n, uint32_t seed, const geom::CrystalRecipe rc,
                                                                   uint64_t first_index) {
  const uint32_t k = blockIdx.x * kGenBlock + threadIdx.x;
  if (k >= n) return;
  geom::MakeShapeDev(seed, rc, first_index + k, pool[k]);
}

// ------------------------------------------------------------------------------------------------------------------
// Pyramid family: one TEAM of 32 lanes builds one crystal (two teams per wave64, eight per workgroup).
//
// The serial builder (geom::BuildPyramidShape, one thread per crystal) keeps ~4 KB of per-lane arrays in scratch — planes,
// vertices, per-face vertex lists — and ran at two waves per SIMD; at 1.4 ms per 125 K crystals it cost as much as the trace
// of their 4 M rays.  Here the same steps are spread over a team and the arrays live in LDS:
//   planes       lane s < 20 builds raw / unit plane s                                  (geom::PyrRawPlane / PyrUnitPlane)
//   cone apexes  lane t < 20 solves triple t of its cone, team max                      (geom::Concurrence)
//   vertices     the ~100 candidate triples in lexicographic order, 32 per round, one solve + feasibility scan per lane;
//                feasible candidates are then taken IN ORDER (ballot, lowest lane first), broadcast, tested against the kept
//                vertices (one per lane) and appended — the serial duplicate filter, unchanged
//   faces        lane s: vertices on plane s, CCW order                                 (geom::PyrOrderFace)
//   tables       lane s emits face row + fan triangles at offsets from a team prefix sum (geom::EmitFace), then pairs the
//                opposite faces (geom::FinalizeSlabs' rule) with ballots
// Every number is produced by the same expression on the same operands as in the serial builder, reductions are max / any
// (order-free), and candidates are filtered in the serial order, so the record is bit-identical to the host's
// (tests/test_gpu_parity.py::test_device_crystal_generator_equals_host_builder).
constexpr int kTeam = 32, kTeamsPerBlock = 8, kTeamBlock = kTeam * kTeamsPerBlock;

struct TeamLds {
  geom::Plane3 unit[20];
  double verts[geom::kPyrMaxVerts][3];
  double ang[20][HALO_MAX_FACE_VTX];
  uint8_t on[20][HALO_MAX_FACE_VTX];
  float fn[20][4];      // emitted face rows by compact id (unit normal, plane constant)
  int tri_cnt[20];      // fan triangles of slot s (0 when absent)
};

__device__ __forceinline__ uint32_t team_ballot(bool p) {
  const unsigned long long b = __ballot(p);
  return static_cast<uint32_t>(b >> (threadIdx.x & 32u));
}
__device__ __forceinline__ double team_bcast(double v, int src) {   // value of team lane `src`
  return __shfl(v, static_cast<int>((threadIdx.x & 32u) | static_cast<uint32_t>(src)));
}
__device__ __forceinline__ double team_max(double v) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}

// candidate triple number t of the restricted enumeration (see geom::BuildPyramidShape), lexicographic order
__device__ __forceinline__ bool team_triple(int t, bool upper, bool lower, int& i, int& j, int& k) {
  const int pair_a[15] = {0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 3, 3, 4}, pair_b[15] = {1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};
  const int tri_a[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2, 3}, tri_b[20] = {1, 1, 1, 1, 2, 2, 2, 3, 3, 4, 2, 2, 2, 3, 3, 4, 3, 3, 4, 4},
            tri_c[20] = {2, 3, 4, 5, 3, 4, 5, 4, 5, 5, 3, 4, 5, 4, 5, 5, 4, 5, 5, 5};
  if (t < 30) {   // basal plane b with two planes of its cone (or two prism planes without one)
    const int b = t / 15, q = t % 15;
    const int lo = (b == 0) ? (upper ? 8 : 2) : (lower ? 14 : 2);
    i = b;
    j = lo + pair_a[q];
    k = lo + pair_b[q];
    return true;
  }
  t -= 30;
  const int per_pair = (upper ? 1 : 0) + (lower ? 1 : 0);
  if (t < 15 * per_pair) {   // prism planes i < j with cone plane i of the upper, then the lower cone
    const int q = t / per_pair, which = t % per_pair;
    i = 2 + pair_a[q];
    j = 2 + pair_b[q];
    k = ((which == 0 && upper) ? 8 : 14) + pair_a[q];
    return true;
  }
  t -= 15 * per_pair;
  if (upper) {
    if (t < 20) {
      i = 8 + tri_a[t];
      j = 8 + tri_b[t];
      k = 8 + tri_c[t];
      return true;
    }
    t -= 20;
  }
  if (lower && t < 20) {
    i = 14 + tri_a[t];
    j = 14 + tri_b[t];
    k = 14 + tri_c[t];
    return true;
  }
  return false;
}

__global__ void __launch_bounds__(kTeamBlock, 4) halo_pyrgen_team_kernel(ShapeDev* __restrict__ pool, uint32_t n, uint32_t seed, const geom::CrystalRecipe rc,
                                                                        uint64_t first_index) {
  __shared__ TeamLds s_team[kTeamsPerBlock];
  const int lane = static_cast<int>(threadIdx.x & 31u);
  TeamLds& T = s_team[threadIdx.x >> 5];
  const uint32_t crystal = blockIdx.x * kTeamsPerBlock + (threadIdx.x >> 5);
  const bool live = crystal < n;                       // team-uniform
  ShapeDev& out = pool[live ? crystal : 0u];
  // --- shape scalars: lane q < 9 draws scalar q (a Gaussian draw is a logf + cosf), the team shares them ---
  float sc[9];
  {
    const float mine = lane < 9 ? geom::DrawShapeScalarOne(seed, rc, first_index + (live ? crystal : 0u), lane) : 0.0f;
#pragma unroll
    for (int q = 0; q < 9; q++) sc[q] = __shfl(mine, static_cast<int>((threadIdx.x & 32u) | static_cast<uint32_t>(q)));
  }
  float dist[6];
  for (int i = 0; i < 6; i++) dist[i] = sc[3 + i];
  const float h1 = fabsf(sc[0]), h2 = fabsf(sc[1]), h3 = fabsf(sc[2]);
  const double cot_u = rc.cot_u, cot_l = rc.cot_l;
  const bool upper = h1 > geom::kGeomFloatEps && cot_u >= 0.0;
  const bool lower = h3 > geom::kGeomFloatEps && cot_l >= 0.0;
  bool valid = live && !(!upper && !lower && h2 < geom::kGeomFloatEps);
  const double k8 = static_cast<double>(geom::kGeomSqrt3) / 8.0, half = 0.5 * static_cast<double>(h2);
  const double a1 = upper ? cot_u : -1.0;
  const double a2 = lower ? cot_l : -1.0;
  // --- planes: lane s owns slot s ---
  const int s = lane;
  const bool side_active = s >= 2 && s < 20 && ((s < 8) || (s < 14 ? upper : lower));
  geom::Plane3 raw = geom::Plane3{0.0, 0.0, 0.0, 0.0}, unit = raw;
  if (side_active) {
    raw = geom::PyrRawPlane(s, a1, a2, half, k8, dist);
    unit = geom::PyrUnitPlane(raw);
  }
  const double scale = team_max(fmax(fabs(half), side_active ? fabs(unit.d) : 0.0));
  const double tol = 5.0 * static_cast<double>(geom::kGeomFloatEps) * fmax(scale, 1e-3);
  if (s < 20) T.unit[s] = unit;
  // --- cone apexes: extreme z over the feasible concurrences of each cone's own six planes ---
  double z_top = half, z_bot = -half;
  for (int c = 0; c < 2; c++) {
    if (!(c == 0 ? upper : lower)) continue;
    const int lo = (c == 0) ? 8 : 14;
    const double sign = (c == 0) ? 1.0 : -1.0;
    int i, j, k;
    bool found = false;
    double zc = 0.0;
    if (lane < 20 && team_triple(30 + 15 * ((upper ? 1 : 0) + (lower ? 1 : 0)) + ((c == 1 && upper) ? 20 : 0) + lane, upper, lower, i, j, k)) {
      double x[3];
      if (geom::Concurrence(T.unit[i], T.unit[j], T.unit[k], x)) {
        bool ok = true;
        for (int m = 0; m < 6 && ok; m++) ok = geom::EvalPlane(T.unit[lo + m], x) <= tol;
        if (ok) {
          found = true;
          zc = x[2];
        }
      }
    }
    const bool any = team_ballot(found) != 0u;
    const double best = sign * team_max(found ? sign * zc : -1e300);
    if (!any) valid = false;
    if (c == 0) z_top = half + static_cast<double>(h1) * (best - half);
    else z_bot = -half + static_cast<double>(h3) * (best + half);
  }
  if (s == 0) raw = unit = geom::Plane3{0.0, 0.0, 1.0, -z_top};
  if (s == 1) raw = unit = geom::Plane3{0.0, 0.0, -1.0, z_bot};
  if (s < 2) T.unit[s] = unit;
  const bool active = s < 2 || side_active;
  const uint32_t act_mask = team_ballot(active);
  // --- vertices: candidate triples in lexicographic order, 32 per round; kept in serial order ---
  // (the kept vertices live in registers while the list grows — lane v holds vertex v and vertex 32 + v — and go to LDS once, for
  // the face phase: the serial filter below runs ~25 times per crystal and used to read and write the list in LDS each time)
  int nv = 0;
  double k0[3] = {0.0, 0.0, 0.0}, k1[3] = {0.0, 0.0, 0.0};
  const int total = 30 + 15 * ((upper ? 1 : 0) + (lower ? 1 : 0)) + (upper ? 20 : 0) + (lower ? 20 : 0);
  for (int base = 0; base < total; base += kTeam) {
    int i, j, k;
    double x[3] = {0.0, 0.0, 0.0};
    bool feasible = false;
    if (valid && base + lane < total && team_triple(base + lane, upper, lower, i, j, k)) {
      if (geom::Concurrence(T.unit[i], T.unit[j], T.unit[k], x)) {
        bool ok = true;
        for (int m = 0; m < 20; m++)   // EvalPlane(unit[m], x) <= tol over the active planes
          if ((act_mask >> m) & 1u) ok = ok && (T.unit[m].a * x[0] + T.unit[m].b * x[1] + T.unit[m].c * x[2] + T.unit[m].d <= tol);
        feasible = ok;
      }
    }
    uint32_t todo = team_ballot(feasible);
    while (todo != 0u) {   // (the two teams of a wave may differ: the loop runs for the longer list)
      const int src = __ffs(todo) - 1;
      todo &= todo - 1u;
      const double cx = team_bcast(x[0], src), cy = team_bcast(x[1], src), cz = team_bcast(x[2], src);
      bool dup = false;
      if (lane < nv) {
        const double dx = k0[0] - cx, dy = k0[1] - cy, dz = k0[2] - cz;
        if (!(fabs(dx) > 4.0 * tol || fabs(dy) > 4.0 * tol || fabs(dz) > 4.0 * tol) && sqrt(dx * dx + dy * dy + dz * dz) <= 2.0 * tol) dup = true;
      }
      if (lane + kTeam < nv) {
        const double dx = k1[0] - cx, dy = k1[1] - cy, dz = k1[2] - cz;
        if (!(fabs(dx) > 4.0 * tol || fabs(dy) > 4.0 * tol || fabs(dz) > 4.0 * tol) && sqrt(dx * dx + dy * dy + dz * dz) <= 2.0 * tol) dup = true;
      }
      if (team_ballot(dup) == 0u && nv < geom::kPyrMaxVerts) {
        if (lane == (nv & (kTeam - 1))) {
          if (nv < kTeam) {
            k0[0] = cx, k0[1] = cy, k0[2] = cz;
          } else {
            k1[0] = cx, k1[1] = cy, k1[2] = cz;
          }
        }
        nv++;
      }
    }
  }
  if (lane < nv)
    for (int a = 0; a < 3; a++) T.verts[lane][a] = k0[a];
  if (lane + kTeam < nv)
    for (int a = 0; a < 3; a++) T.verts[lane + kTeam][a] = k1[a];
  // --- faces: vertices on plane s (ascending vertex order), then CCW order ---
  int cnt = 0;
  if (valid && active) {
    for (int v = 0; v < nv; v++)
      if (fabs(unit.a * T.verts[v][0] + unit.b * T.verts[v][1] + unit.c * T.verts[v][2] + unit.d) <= 2.0 * tol && cnt < HALO_MAX_FACE_VTX) T.on[s][cnt++] = static_cast<uint8_t>(v);
  }
  int on_n = 0;
  if (valid && active && cnt >= 3) on_n = geom::PyrOrderFace(T.verts, T.on[s], cnt, unit, tol, T.ang[s]);
  const uint32_t present = team_ballot(on_n > 0);
  if (__popc(present) < 4) valid = false;
  // --- tables ---
  if (!valid) {
    if (live && lane == 0) out.face_cnt = out.tri_cnt = out.slab_cnt = out.single_cnt = 0;
    return;   // (the other team of this wave carries on: nothing below synchronises across teams)
  }
  const int my_tris = on_n > 0 ? on_n - 2 : 0;
  if (s < 20) T.tri_cnt[s] = my_tris;
  int tri_start = 0, tri_total = 0;
  for (int q = 0; q < 20; q++) {
    const int c = T.tri_cnt[q];
    if (q < s) tri_start += c;
    tri_total += c;
  }
  const int fid = __popc(present & ((1u << s) - 1u));
  float nrm[3] = {0.0f, 0.0f, 0.0f};
  if (on_n > 0) {
    geom::ShapeCursor cur;
    cur.fid = fid;
    cur.tri = min(tri_start, static_cast<int>(sizeof(out.tri_na) / 16u));
    const float plane[4] = {static_cast<float>(raw.a), static_cast<float>(raw.b), static_cast<float>(raw.c), static_cast<float>(raw.d)};
    nrm[0] = static_cast<float>(unit.a);
    nrm[1] = static_cast<float>(unit.b);
    nrm[2] = static_cast<float>(unit.c);
    float loop[HALO_MAX_FACE_VTX][3];
    for (int q = 0; q < on_n; q++)
      for (int a = 0; a < 3; a++) loop[q][a] = static_cast<float>(T.verts[T.on[s][q]][a]);
    geom::EmitFace(out, cur, plane, nrm, geom::kPyrFaceNumber[s], loop, on_n);
    T.fn[fid][0] = nrm[0];
    T.fn[fid][1] = nrm[1];
    T.fn[fid][2] = nrm[2];
    T.fn[fid][3] = cur.d_last;
  }
  // opposite-face slabs (geom::FinalizeSlabs): face i pairs with the first later face whose unit normal is its exact negative
  const int face_cnt = __popc(present);
  int mate = -1;
  bool is_minus = false;
  if (on_n > 0) {
    for (int j2 = fid + 1; j2 < face_cnt && mate < 0; j2++)
      if (nrm[0] == -T.fn[j2][0] && nrm[1] == -T.fn[j2][1] && nrm[2] == -T.fn[j2][2]) mate = j2;
    for (int j2 = 0; j2 < fid && !is_minus; j2++)
      if (T.fn[j2][0] == -nrm[0] && T.fn[j2][1] == -nrm[1] && T.fn[j2][2] == -nrm[2]) {
        // j2 takes this face only if no face between them already matched j2's normal — distinct faces of a convex solid
        // never share a normal, so the first match is the only one
        is_minus = true;
      }
  }
  const bool is_plus = on_n > 0 && mate >= 0 && !is_minus;
  const bool is_single = on_n > 0 && !is_plus && !is_minus;
  const uint32_t plus_mask = team_ballot(is_plus), single_mask = team_ballot(is_single);
  const uint32_t below = (1u << s) - 1u;
  if (is_plus) {
    float* r = out.slab[__popc(plus_mask & below)];
    r[0] = nrm[0];
    r[1] = nrm[1];
    r[2] = nrm[2];
    r[3] = T.fn[fid][3];
    r[4] = T.fn[mate][3];
    reinterpret_cast<uint32_t*>(r)[5] = static_cast<uint32_t>(fid);
    reinterpret_cast<uint32_t*>(r)[6] = static_cast<uint32_t>(mate);
    r[7] = 0.0f;
  }
  if (is_single) out.single[__popc(single_mask & below)] = static_cast<uint8_t>(fid);
  if (lane == 0) {
    out.face_cnt = face_cnt;
    out.tri_cnt = min(tri_total, static_cast<int>(sizeof(out.tri_na) / 16u));
    out.slab_cnt = __popc(plus_mask);
    out.single_cnt = __popc(single_mask);
  }
}

hipError_t launch_shapegen(void* pool, bool prism_records, uint32_t n, uint32_t seed, const geom::CrystalRecipe& rc, uint64_t first_index,
                           hipStream_t stream, bool serial_pyramid) {
  if (n == 0) return hipSuccess;
  const dim3 grid((n + kGenBlock - 1) / kGenBlock), block(kGenBlock);
  if (prism_records) hipLaunchKernelGGL(halo_shapegen_kernel<ShapePrism>, grid, block, 0, stream, static_cast<ShapePrism*>(pool), n, seed, rc, first_index);
  else if (rc.c.kind == HALO_CRYSTAL_PYRAMID && !serial_pyramid)
    hipLaunchKernelGGL(halo_pyrgen_team_kernel, dim3((n + kTeamsPerBlock - 1) / kTeamsPerBlock), dim3(kTeamBlock), 0, stream, static_cast<ShapeDev*>(pool), n, seed, rc,
                       first_index);
  else hipLaunchKernelGGL(halo_shapegen_kernel<ShapeDev>, grid, block, 0, stream, static_cast<ShapeDev*>(pool), n, seed, rc, first_index);
  return hipGetLastError();
}

}  // namespace halo
