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
//   scalars      lane q < 9 draws scalar q from the recipe's draw plan (geom::DrawShapeScalarOne); lane s takes its face distance by shuffle
//   planes       lane s < 20 builds raw / unit plane s                                  (geom::PyrRawPlaneOne / PyrUnitPlane)
//   cone apexes  lane t < 20 solves triple t of its cone, team max; the feasible ones are parked for the vertex phase (geom::Concurrence)
//   vertices     the 60 basal / prism-pair triples in lexicographic order, then the parked cone survivors, 32 per round: one solve + one scan of
//                the planes per lane — exact feasibility (no plane has the point outside by more than 1e-9 of the crystal's size) and the set
//                of planes through it — then the serial duplicate filter evaluated in parallel: every feasible lane tests its candidate
//                against the kept vertices and the round's other candidates (float pre-test, fp64 only for near pairs), the lowest of each
//                duplicate group is kept, in lane order = list order (see the loop for why that IS the serial filter), and a duplicate's
//                planes are OR-ed into the vertex it folds into (LDS)
//   faces        lane v holds vertex v's incidence mask, one ballot per plane hands lane s the vertices on plane s; CCW order by float
//                pseudo-angle keys (order_face_fast: geom::PyrOrderFace's order without its divisions; the serial ordering where keys are close)
//   check        fan triangles = 2 V - 4 and >= 4 faces (Euler), else the exhaustive enumeration of all 1140 triples runs (second inlined copy)
//   tables       lane s emits face row + fan triangles at offsets from a team prefix sum (geom::EmitFace), then pairs the
//                opposite faces (geom::FinalizeSlabs' rule: the one slot that can hold the exact negative normal) with ballots
// Every number is produced by the same expression on the same operands as in the serial builder, reductions are max / any
// (order-free), and candidates are kept in the serial order, so the record is bit-identical to the host's
// (tests/test_gpu_parity.py::test_device_crystal_generator_equals_host_builder).
// Round 3: 4.28 -> 2.75 ms per 781 K crystals (measured per phase with early exits: draw plan + no scratch tables 0.96 -> 0.24 ms,
// parallel duplicate filter and branch-free plane scan 1.30 -> 1.14 ms, slab pairing 0.31 -> ~0.05 ms; face collection ~0.65 ms until the
// incidence masks replaced the per-vertex plane evaluation).
constexpr int kTeam = 32, kTeamsPerBlock = 8, kTeamBlock = kTeam * kTeamsPerBlock;

struct TeamLds {
  geom::Plane3 unit[20];
  double verts[geom::kPyrMaxVerts][3];
  uint32_t vmask[geom::kPyrMaxVerts];         // planes through kept vertex v (its own three, whatever else meets there, united over its duplicates)
  union {
    struct {   // vertex phase
      double cand[kTeam][3];                  // the round's feasible candidates
      double park[geom::kPyrMaxVerts][3];     // feasible concurrences of the cones' own triples (<= 20 per cone), parked by the apex phase
      float4 cand_f[kTeam];                   // float copies for the distance pre-test (w = unused)
      float4 kept_f[geom::kPyrMaxVerts];
      uint32_t park_m[geom::kPyrMaxVerts];    // the three planes of a parked concurrence
      uint32_t rmask[kTeam];                  // a round's incidence masks, gathered on the lanes that are kept
    } vtx;
    struct {   // face phase
      double key[20][HALO_MAX_FACE_VTX];      // pseudo-angle sort keys (float keys of order_face_fast in the front half; doubles when geom::PyrOrderFace runs)
      uint8_t on[20][HALO_MAX_FACE_VTX];
      float fn[20][4];      // emitted face rows by compact id (unit normal, plane constant)
      int tri_cnt[20];      // fan triangles of slot s (0 when absent)
    } f;
  };
};

__device__ __forceinline__ uint32_t team_ballot(bool p) {
  const unsigned long long b = __ballot(p);
  return static_cast<uint32_t>(b >> (threadIdx.x & 32u));
}
__device__ __forceinline__ double team_bcast(double v, int src) {   // value of team lane `src`
  return __shfl(v, static_cast<int>((threadIdx.x & 32u) | static_cast<uint32_t>(src)));
}
// LDS written by one lane of a team and read by others: the wave runs in lockstep, but nothing makes the compiler keep LDS accesses of
// different lanes in program order across divergent regions — publish with a workgroup-scope fence + a wave barrier (like stage_shape's
// callers in halo_trace.inl).
__device__ __forceinline__ void team_publish() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
// sqrt(d2) <= lim, as the serial builders evaluate it — but the fp64 square root (a long software sequence on this part) only when d2 is
// within a few ulps of lim^2, where its rounding could decide; everywhere else the comparison of the squares gives the same answer.
__device__ __forceinline__ bool within(double d2, double lim) {
  const double l2 = lim * lim;
  if (d2 > l2 * (1.0 + 1e-12)) return false;
  if (d2 < l2 * (1.0 - 1e-12)) return true;
  return sqrt(d2) <= lim;
}
__device__ __forceinline__ double team_max(double v) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}

// geom::PyrOrderFace for the team kernel: the same ORDER without its eight fp64 divisions and its square root.  The serial form divides
// the vertex sum by the count (centroid), normalises the first spoke, and sorts by a pseudo-angle |y| / (|x| + |y|) per quadrant — but only
// the cyclic order around the centroid is used, and that is invariant under a positive scaling of the spokes and of the frame: here the
// spokes are cnt * (v - centroid) = cnt * v - sum, the frame is the first spoke as it is, and the sort key is the float form of the same
// pseudo-angle (error ~2e-7).  That decides the order only where the keys are clearly apart: two corners 1e-4 apart on a sliver face can
// subtend 1e-8 at the centroid.  So the function REFUSES (returns -1, list untouched) when two neighbouring keys — or the last key and the
// full turn — are closer than 1e-4, or when the face is nearly a point, and the caller then runs geom::PyrOrderFace itself: the table is
// the serial builder's in every case (tests/test_gpu_parity.py::test_device_crystal_generator_equals_host_builder: one crystal in 4000
// of its "full apexes" recipe takes the refusal).
__device__ __forceinline__ int order_face_fast(const double (*verts)[3], uint8_t* on, int cnt, const geom::Plane3& unit, double tol, float* key) {
  double c[3] = {0, 0, 0};
  for (int q = 0; q < cnt; q++)
    for (int a = 0; a < 3; a++) c[a] += verts[on[q]][a];
  const double k = static_cast<double>(cnt);
  const double e1[3] = {k * verts[on[0]][0] - c[0], k * verts[on[0]][1] - c[1], k * verts[on[0]][2] - c[2]};
  const double lim = k * tol;
  if (e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2] < 4.0 * lim * lim) return -1;   // |v0 - centroid| within 2 tol: the serial test decides
  const double n[3] = {unit.a, unit.b, unit.c};
  const double e2[3] = {n[1] * e1[2] - n[2] * e1[1], n[2] * e1[0] - n[0] * e1[2], n[0] * e1[1] - n[1] * e1[0]};
  const uint32_t saved[3] = {reinterpret_cast<const uint32_t*>(on)[0], reinterpret_cast<const uint32_t*>(on)[1], reinterpret_cast<const uint32_t*>(on)[2]};
  static_assert(HALO_MAX_FACE_VTX == 12, "a face's vertex list is three dwords");
  key[0] = 0.0f;
  for (int q = 1; q < cnt; q++) {
    const double r[3] = {k * verts[on[q]][0] - c[0], k * verts[on[q]][1] - c[1], k * verts[on[q]][2] - c[2]};
    const double yd = r[0] * e2[0] + r[1] * e2[1] + r[2] * e2[2], xd = r[0] * e1[0] + r[1] * e1[1] + r[2] * e1[2];
    const float x = static_cast<float>(xd), y = static_cast<float>(yd);   // (products of crystal-size lengths: far inside float's range)
    const float ax = fabsf(x), ay = fabsf(y);
    const float t = (ax + ay > 0.0f) ? ay * __builtin_amdgcn_rcpf(ax + ay) : 0.0f;   // (1 ulp: the keys only have to order)
    key[q] = (y >= 0.0f) ? (x >= 0.0f ? t : 2.0f - t) : (x < 0.0f ? 2.0f + t : 4.0f - t);
  }
  for (int q = 1; q < cnt; q++) {  // stable insertion sort by key
    const float ka = key[q];
    const uint8_t kv = on[q];
    int p = q - 1;
    while (p >= 0 && key[p] > ka) {
      key[p + 1] = key[p];
      on[p + 1] = on[p];
      p--;
    }
    key[p + 1] = ka;
    on[p + 1] = kv;
  }
  bool close = 4.0f - key[cnt - 1] < 1e-4f;
  for (int q = 1; q < cnt; q++) close = close || (key[q] - key[q - 1] < 1e-4f);
  if (close) {
    for (int a = 0; a < 3; a++) reinterpret_cast<uint32_t*>(on)[a] = saved[a];
    return -1;
  }
  return cnt;
}

template <class... T>
constexpr unsigned long long PackOctal(T... v) {
  unsigned long long out = 0ull;
  int sh = 0;
  ((out |= static_cast<unsigned long long>(v) << sh, sh += 3), ...);
  return out;
}
template <class... T>
constexpr unsigned long long PackOctal15(T... v) {
  static_assert(sizeof...(v) == 15, "15 pairs");
  return PackOctal(v...);
}
template <class... T>
constexpr unsigned long long PackOctal20(T... v) {
  static_assert(sizeof...(v) == 20, "20 triples");
  return PackOctal(v...);
}
// candidate triple number t of the restricted enumeration (see geom::BuildPyramidShape), lexicographic order
__device__ __forceinline__ bool team_triple(int t, bool upper, bool lower, int& i, int& j, int& k) {
  // pairs (a < b) and triples (a < b < c) of 0..5 in lexicographic order, packed 3 bits per entry (register constants: an array indexed by
  // the lane would be a load from constant memory in the middle of a dependent chain)
  //   pair_a = 0,0,0,0,0,1,1,1,1,2,2,2,3,3,4   pair_b = 1,2,3,4,5,2,3,4,5,3,4,5,4,5,5
  //   tri_a  = 0 x10, 1 x6, 2 x3, 3            tri_b  = 1,1,1,1,2,2,2,3,3,4,2,2,2,3,3,4,3,3,4,4      tri_c = 2,3,4,5,3,4,5,4,5,5,3,4,5,4,5,5,4,5,5,5
  constexpr unsigned long long kPairA = PackOctal15(0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 3, 3, 4), kPairB = PackOctal15(1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5);
  constexpr unsigned long long kTriA = PackOctal20(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2, 3), kTriB = PackOctal20(1, 1, 1, 1, 2, 2, 2, 3, 3, 4, 2, 2, 2, 3, 3, 4, 3, 3, 4, 4),
                               kTriC = PackOctal20(2, 3, 4, 5, 3, 4, 5, 4, 5, 5, 3, 4, 5, 4, 5, 5, 4, 5, 5, 5);
  auto pick = [](unsigned long long packed, int idx) { return static_cast<int>((packed >> (3 * idx)) & 7ull); };
  if (t < 30) {   // basal plane b with two planes of its cone (or two prism planes without one)
    const int b = t / 15, q = t % 15;
    const int lo = (b == 0) ? (upper ? 8 : 2) : (lower ? 14 : 2);
    i = b;
    j = lo + pick(kPairA, q);
    k = lo + pick(kPairB, q);
    return true;
  }
  t -= 30;
  const int per_pair = (upper ? 1 : 0) + (lower ? 1 : 0);
  if (t < 15 * per_pair) {   // prism planes i < j with cone plane i of the upper, then the lower cone
    const int q = t / per_pair, which = t % per_pair;
    i = 2 + pick(kPairA, q);
    j = 2 + pick(kPairB, q);
    k = ((which == 0 && upper) ? 8 : 14) + pick(kPairA, q);
    return true;
  }
  t -= 15 * per_pair;
  if (upper) {
    if (t < 20) {
      i = 8 + pick(kTriA, t);
      j = 8 + pick(kTriB, t);
      k = 8 + pick(kTriC, t);
      return true;
    }
    t -= 20;
  }
  if (lower && t < 20) {
    i = 14 + pick(kTriA, t);
    j = 14 + pick(kTriB, t);
    k = 14 + pick(kTriC, t);
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
  // (no per-lane array of the nine: lane s takes the one face distance its plane uses straight from the lane that drew it)
  const float mine = lane < 9 ? geom::DrawShapeScalarOne(seed, rc, first_index + (live ? crystal : 0u), lane) : 0.0f;
  const int team_base = static_cast<int>(threadIdx.x & 32u);
  const float h1 = fabsf(__shfl(mine, team_base | 0)), h2 = fabsf(__shfl(mine, team_base | 1)), h3 = fabsf(__shfl(mine, team_base | 2));
  const float my_dist = __shfl(mine, team_base | (3 + (lane + 4) % 6));   // (s - 2) mod 6 for s = lane >= 2
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
    raw = geom::PyrRawPlaneOne(s, a1, a2, half, k8, my_dist);
    unit = geom::PyrUnitPlane(raw);
  }
  const double scale = team_max(fmax(fabs(half), side_active ? fabs(unit.d) : 0.0));
  const double tol = 5.0 * static_cast<double>(geom::kGeomFloatEps) * fmax(scale, 1e-3);
  const double tight = 1e-9 * fmax(scale, 1e-3);   // "on the plane" for a concurrence computed in double (geom::BuildPyramidShape)
  const double merge = geom::PyrMergeRadius(team_max((s >= 2 && s < 8) ? fabs(static_cast<double>(my_dist)) : 0.0));   // the lateral duplicate radius (geom::BuildPyramidShape)
  if (s < 20) T.unit[s] = unit;
  team_publish();
  // --- cone apexes: extreme z over the feasible concurrences of each cone's own six planes ---
  // The 20 triples of a cone are also the last candidates of the vertex enumeration below, and a concurrence that violates one of its own
  // cone's planes cannot be a vertex: the survivors of this phase (a handful of the 40) are parked — in LDS the face phase reuses later, so free until the
  // vertex list is final — in their list order, and the vertex phase takes them from there instead of solving and scanning all 40 again.
  double z_top = half, z_bot = -half;
  int ns = 0;   // parked survivors (team-uniform)
  static_assert(geom::kPyrMaxVerts >= 40, "40 cone triples can be parked");
  double (*const cand_lds)[3] = T.vtx.cand;
  double (*const park)[3] = T.vtx.park;
  for (int c = 0; c < 2; c++) {
    if (!(c == 0 ? upper : lower)) continue;
    const int lo = (c == 0) ? 8 : 14;
    const double sign = (c == 0) ? 1.0 : -1.0;
    int i, j, k;
    bool found = false;
    double zc = 0.0;
    double xs[3] = {0.0, 0.0, 0.0};
    if (lane < 20 && team_triple(30 + 15 * ((upper ? 1 : 0) + (lower ? 1 : 0)) + ((c == 1 && upper) ? 20 : 0) + lane, upper, lower, i, j, k)) {
      if (geom::Concurrence(T.unit[i], T.unit[j], T.unit[k], xs)) {
        bool ok = true;
        for (int m = 0; m < 6 && ok; m++) ok = geom::EvalPlane(T.unit[lo + m], xs) <= tol;
        if (ok) {
          found = true;
          zc = xs[2];
        }
      }
    }
    const uint32_t fmask = team_ballot(found);
    if (found) {
      const int slot = ns + __popc(fmask & ((1u << lane) - 1u));
      for (int a = 0; a < 3; a++) park[slot][a] = xs[a];
      T.vtx.park_m[slot] = (1u << i) | (1u << j) | (1u << k);
    }
    ns += __popc(fmask);
    const bool any = fmask != 0u;
    const double best = sign * team_max(found ? sign * zc : -1e300);
    if (!any) valid = false;
    if (c == 0) z_top = half + static_cast<double>(h1) * (best - half);
    else z_bot = -half + static_cast<double>(h3) * (best + half);
  }
  if (s == 0) raw = unit = geom::Plane3{0.0, 0.0, 1.0, -z_top};
  if (s == 1) raw = unit = geom::Plane3{0.0, 0.0, -1.0, z_bot};
  if (s < 2) T.unit[s] = unit;
  team_publish();
  const bool active = s < 2 || side_active;
  const uint32_t act_mask = team_ballot(active);
  // --- vertices: candidate triples in lexicographic order, 32 per round; kept in serial order ---
  // The serial filter keeps a candidate unless a vertex kept before it lies within the merge radius.  Taking the feasible candidates one at a time
  // (broadcast, test against the kept list, append) cost ~40 instructions per candidate, ~35 candidates per crystal: a third of the
  // kernel.  Here a round's feasible candidates go to LDS, every feasible lane tests its own against the vertices kept in earlier rounds
  // (a hit drops it, as in the serial filter) and against the round's other candidates, gathering the mask D of those within the radius; among
  // the candidates that survive the first test the serial filter keeps exactly the lowest of each group when the groups are cliques
  // (D equal for all members: every duplicate of a vertex is a duplicate of its other duplicates) — checked, and when it does not hold
  // the greedy pass runs on the masks, which is the serial filter itself.  Kept candidates are appended in lane order = list order.
  // Two passes at most (geom::BuildPyramidShape): the short candidate lists first; if what they give is no polytope (fan triangles != 2 V - 4:
  // corners of two rings within a few tolerances of each other, where triples that meet outside the exact solid pass the feasibility test
  // too) the exhaustive enumeration of all C(20,3) triples — the definition — runs instead, 36 rounds of 32.
  int nv = 0, on_n = 0, tri_start = 0, tri_total = 0;
  uint32_t present = 0u;
  const uint32_t below = (1u << lane) - 1u;
  auto build = [&](const bool exhaustive) __attribute__((always_inline)) {
  nv = 0;
  const int total_nc = exhaustive ? 1140 : 30 + 15 * ((upper ? 1 : 0) + (lower ? 1 : 0));   // basal and prism-pair triples; the cone triples follow as parked survivors
  const int total = exhaustive ? 1140 : total_nc + ns;
  for (int base = 0; base < total; base += kTeam) {
    int i, j, k;
    double x[3] = {0.0, 0.0, 0.0};
    bool feasible = false;
    const int cand = base + lane;
    bool solved = false;
    uint32_t mk = 0u;   // planes through the candidate
    if (exhaustive) {
      if (cand < total) {   // triple number `cand` of 0 <= i < j < k < 20, lexicographic
        int rem = cand;
        i = 0;
        while (rem >= (19 - i) * (18 - i) / 2) {
          rem -= (19 - i) * (18 - i) / 2;
          i++;
        }
        j = i + 1;
        while (rem >= 19 - j) {
          rem -= 19 - j;
          j++;
        }
        k = j + 1 + rem;
        if (((act_mask >> i) & (act_mask >> j) & (act_mask >> k) & 1u) != 0u) solved = geom::Concurrence(T.unit[i], T.unit[j], T.unit[k], x);
      }
    } else {
      if (cand < total_nc && team_triple(cand, upper, lower, i, j, k)) solved = geom::Concurrence(T.unit[i], T.unit[j], T.unit[k], x);
      if (cand >= total_nc && cand < total) {
        for (int a = 0; a < 3; a++) x[a] = park[cand - total_nc][a];
        mk = T.vtx.park_m[cand - total_nc];
        solved = true;
      }
    }
    if (solved) {
      if (mk == 0u) mk = (1u << i) | (1u << j) | (1u << k);
      bool ok = true;
      // EvalPlane(unit[m], x) <= tight over the active planes (exact feasibility: the concurrence is computed in double), and the planes that
      // pass through x — over all twenty slots without a branch: the slots of an absent cone hold the zero plane, which evaluates to 0 and is
      // masked out of the incidences afterwards
#pragma unroll 5
      for (int m = 0; m < 20; m++) {
        const double ev = geom::EvalPlane4(T.unit[m].a, T.unit[m].b, T.unit[m].c, T.unit[m].d, x[0], x[1], x[2]);
        ok = ok & (ev <= tight);
        mk |= (fabs(ev) <= tight ? 1u : 0u) << m;
      }
      mk &= act_mask;
      feasible = ok;
    }
    const uint32_t fm = team_ballot(feasible);
    // distance pre-test in float: a pair within the merge radius in fp64 is within thr in float whatever the vertices' magnitude (the float copies
    // are off by <= 2^-24 of each coordinate, |difference error| <= 1.2e-7 * (|x|+|y|+|z|) =: m), so a pair the pre-test calls far IS far
    // and only the pairs it calls near (real duplicates, nearly always) take the fp64 test of the serial filter
    const float fx = static_cast<float>(x[0]), fy = static_cast<float>(x[1]), fz = static_cast<float>(x[2]);
    const float thr = 1.5f * static_cast<float>(merge) + 1e-6f * (fabsf(fx) + fabsf(fy) + fabsf(fz));
    const float thr2 = thr * thr;
    if (feasible) {
      for (int a = 0; a < 3; a++) cand_lds[lane][a] = x[a];
      T.vtx.cand_f[lane] = make_float4(fx, fy, fz, 0.0f);
    }
    team_publish();
    bool dupk = false;
    int dup_v = 0;   // the FIRST kept vertex it duplicates takes its incidences (the serial filter stops there)
    uint32_t D = feasible ? (1u << lane) : 0u;   // (a candidate is within the radius of itself)
    if (feasible) {
      for (int v = 0; v < nv; v++) {
        const float4 o = T.vtx.kept_f[v];
        const float ex = o.x - fx, ey = o.y - fy, ez = o.z - fz;
        if (!dupk && ex * ex + ey * ey + ez * ez <= thr2) {
          const double dx = T.verts[v][0] - x[0], dy = T.verts[v][1] - x[1], dz = T.verts[v][2] - x[2];
          if (within(dx * dx + dy * dy + dz * dz, merge)) {
            dupk = true;
            dup_v = v;
          }
        }
      }
      for (uint32_t m = fm & ~(1u << lane); m != 0u; m &= m - 1u) {
        const int e = __ffs(m) - 1;
        const float4 o = T.vtx.cand_f[e];
        const float ex = o.x - fx, ey = o.y - fy, ez = o.z - fz;
        if (ex * ex + ey * ey + ez * ez <= thr2) {
          const double dx = cand_lds[e][0] - x[0], dy = cand_lds[e][1] - x[1], dz = cand_lds[e][2] - x[2];
          if (within(dx * dx + dy * dy + dz * dz, merge)) D |= 1u << e;
        }
      }
    }
    const bool in_r = feasible && !dupk;
    const uint32_t rm = team_ballot(in_r);
    D &= rm;
    const int rep = in_r ? __ffs(D) - 1 : lane;   // (D holds the lane's own bit)
    const uint32_t d_rep = static_cast<uint32_t>(__shfl(static_cast<int>(D), static_cast<int>((threadIdx.x & 32u) | static_cast<uint32_t>(rep))));
    uint32_t keep = team_ballot(in_r && rep == lane);
    if (__ballot(in_r && d_rep != D) != 0ull) {   // not cliques (vertices about a radius apart in a chain): the greedy pass, in list order
      keep = 0u;
      for (int e = 0; e < kTeam; e++) {
        const uint32_t d_e = static_cast<uint32_t>(__shfl(static_cast<int>(D), static_cast<int>((threadIdx.x & 32u) | static_cast<uint32_t>(e))));
        if (((rm >> e) & 1u) && (d_e & keep) == 0u) keep |= 1u << e;
      }
    }
    // incidences: a duplicate's planes go to the vertex the serial filter would have stopped at — the first earlier-round vertex within
    // the radius, else the lowest kept candidate of this round within it
    const bool kept = ((keep >> lane) & 1u) != 0u;
    if (kept) T.vtx.rmask[lane] = mk;
    team_publish();
    if (feasible && dupk) atomicOr(&T.vmask[dup_v], mk);
    if (in_r && !kept) atomicOr(&T.vtx.rmask[__ffs(D & keep) - 1], mk);
    team_publish();
    if (kept) {
      const int pos = nv + __popc(keep & below);
      if (pos < geom::kPyrMaxVerts) {
        for (int a = 0; a < 3; a++) T.verts[pos][a] = x[a];
        T.vtx.kept_f[pos] = make_float4(fx, fy, fz, 0.0f);
        T.vmask[pos] = T.vtx.rmask[lane];
      }
    }
    nv = min(nv + __popc(keep), geom::kPyrMaxVerts);
    team_publish();
  }
  // --- faces: vertices on plane s (ascending vertex order), then CCW order ---
  // Lane v holds the incidence mask of vertex v (gathered while the vertices were found); a ballot per plane hands lane s the set of
  // vertices on plane s.
  uint32_t on_lo = 0u, on_hi = 0u;
  {
    const bool two = __ballot(nv > kTeam) != 0ull;   // more than 32 vertices in either team of the wave: lane v also holds vertex 32 + v
    const uint32_t pm0 = (valid && lane < nv) ? T.vmask[lane] : 0u;
    const uint32_t pm1 = (two && valid && lane + kTeam < nv) ? T.vmask[lane + kTeam] : 0u;
#pragma unroll
    for (int m = 0; m < 20; m++) {
      const uint32_t b0 = team_ballot((pm0 >> m) & 1u);
      if (s == m) on_lo = b0;
    }
    if (two) {
#pragma unroll
      for (int m = 0; m < 20; m++) {
        const uint32_t b1 = team_ballot((pm1 >> m) & 1u);
        if (s == m) on_hi = b1;
      }
    }
  }
  int cnt = 0;
  if (valid && active) {   // the first HALO_MAX_FACE_VTX of them, ascending
    for (uint32_t m = on_lo; m != 0u && cnt < HALO_MAX_FACE_VTX; m &= m - 1u) T.f.on[s][cnt++] = static_cast<uint8_t>(__ffs(m) - 1);
    for (uint32_t m = on_hi; m != 0u && cnt < HALO_MAX_FACE_VTX; m &= m - 1u) T.f.on[s][cnt++] = static_cast<uint8_t>(kTeam + __ffs(m) - 1);
  }
  on_n = 0;
  if (valid && active && cnt >= 3) {
    on_n = order_face_fast(T.verts, T.f.on[s], cnt, unit, tol, reinterpret_cast<float*>(T.f.key[s]));
    if (on_n < 0) on_n = geom::PyrOrderFace(T.verts, T.f.on[s], cnt, unit, tight, T.f.key[s]);   // too close for float keys: the serial ordering itself
  }
  present = team_ballot(on_n > 0);
  if (s < 20) T.f.tri_cnt[s] = on_n > 0 ? on_n - 2 : 0;
  team_publish();
  tri_start = tri_total = 0;
  for (int q = 0; q < 20; q++) {
    const int c = T.f.tri_cnt[q];
    if (q < s) tri_start += c;
    tri_total += c;
  }
  };
  if (valid) build(false);
  if (valid && !(tri_total == 2 * nv - 4 && __popc(present) >= 4)) {   // no polytope: rare, and the two copies of the phases keep the usual path's registers to itself
    team_publish();   // (the face phase's LDS is the vertex phase's staging area)
    build(true);
  }
  if (!(tri_total == 2 * nv - 4 && __popc(present) >= 4)) valid = false;   // still no polytope: refused, the empty crystal (geom::BuildPyramidShape)
  // --- tables ---
  if (!valid) {
    if (live && lane == 0) out.face_cnt = out.tri_cnt = out.slab_cnt = out.single_cnt = 0;
    return;   // (the other team of this wave carries on: nothing below synchronises across teams)
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
    // geom::EmitFace with the corner loop read where it lies (vertex list and face order in LDS) instead of through a per-lane copy of
    // 16 x 3 floats (192 B of scratch per lane): the same statements on the same values
    {
      const int fid_e = cur.fid;
      const float len = sqrtf(plane[0] * plane[0] + plane[1] * plane[1] + plane[2] * plane[2]);
      const float dn = (len > geom::kGeomFloatEps) ? plane[3] / len : 0.0f;
      out.face[fid_e][0] = nrm[0];
      out.face[fid_e][1] = nrm[1];
      out.face[fid_e][2] = nrm[2];
      out.face[fid_e][3] = dn;
      cur.d_last = dn;
      out.face_number[fid_e] = static_cast<uint8_t>(geom::kPyrFaceNumber[s]);
      float v0[3];
      for (int a = 0; a < 3; a++) v0[a] = static_cast<float>(T.verts[T.f.on[s][0]][a]);
      for (int k = 1; k + 1 < on_n && on_n >= 3 && cur.tri < static_cast<int>(sizeof(out.tri_na) / 16u); k++) {
        const int t = cur.tri;
        float v[9];
        for (int a = 0; a < 3; a++) {
          v[a] = v0[a];
          v[3 + a] = static_cast<float>(T.verts[T.f.on[s][k]][a]);
          v[6 + a] = static_cast<float>(T.verts[T.f.on[s][k + 1]][a]);
        }
        for (int a = 0; a < 9; a++) out.tri_v[t][a] = v[a];
        const float ea[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]};
        const float eb[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
        const float cn[3] = {-eb[1] * ea[2] + ea[1] * eb[2], eb[0] * ea[2] - ea[0] * eb[2], -eb[0] * ea[1] + ea[0] * eb[1]};  // Cross3 math.cpp:36
        const float mag = sqrtf(cn[0] * cn[0] + cn[1] * cn[1] + cn[2] * cn[2]);
        out.tri_na[t][3] = mag / 2.0f;
        for (int c = 0; c < 3; c++) out.tri_na[t][c] = (mag > 0.0f) ? cn[c] / mag : 0.0f;
        out.tri_face[t] = static_cast<uint8_t>(fid_e);
        cur.tri++;
      }
      cur.fid++;
    }
    T.f.fn[fid][0] = nrm[0];
    T.f.fn[fid][1] = nrm[1];
    T.f.fn[fid][2] = nrm[2];
    T.f.fn[fid][3] = cur.d_last;
  }
  team_publish();
  // opposite-face slabs (geom::FinalizeSlabs): face i pairs with the first later face whose unit normal is its exact negative
  const int face_cnt = __popc(present);
  // The serial rule scans the emitted faces for the first later one whose normal is the exact negative.  A face of this family has at
  // most one such partner and it can only sit in one slot: the other basal face, the prism side three steps round, the other cone's face
  // three steps round (equal cone slopes) — every other plane points somewhere else in the horizontal, and two present faces of a convex
  // solid never share a normal (a legal wedge keeps a cone face's normal away from the basal one even in float).  So: look at that slot.
  const int opp = s < 2 ? 1 - s : (s < 8 ? 2 + (s + 1) % 6 : (s < 14 ? 14 + (s - 5) % 6 : 8 + (s - 11) % 6));   // (s - 2 + 3) % 6 etc.
  const float ox = __shfl(nrm[0], team_base | opp), oy = __shfl(nrm[1], team_base | opp), oz = __shfl(nrm[2], team_base | opp);
  const bool paired = on_n > 0 && s < 20 && ((present >> opp) & 1u) && nrm[0] == -ox && nrm[1] == -oy && nrm[2] == -oz;
  const int mate = (paired && opp > s) ? __popc(present & ((1u << opp) - 1u)) : -1;
  const bool is_minus = paired && opp < s;
  const bool is_plus = on_n > 0 && mate >= 0 && !is_minus;
  const bool is_single = on_n > 0 && !is_plus && !is_minus;
  const uint32_t plus_mask = team_ballot(is_plus), single_mask = team_ballot(is_single);
  if (is_plus) {
    float* r = out.slab[__popc(plus_mask & below)];
    r[0] = nrm[0];
    r[1] = nrm[1];
    r[2] = nrm[2];
    r[3] = T.f.fn[fid][3];
    r[4] = T.f.fn[mate][3];
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

// ------------------------------------------------------------------------------------------------------------------
// Prism pools: one TEAM of 8 lanes builds one crystal (eight teams per wave64, sixteen per workgroup), the record assembled in LDS and
// written out with coalesced 16-byte stores.
//
// The serial builder above (one thread per crystal) spent its time waiting: a chain of dependent fp64 steps per lane, 640 B of scratch
// per lane for its candidate / ring / loop arrays, and 1360-byte records 64 lanes wide written 4 bytes at a time — 11.5 % VALU-active,
// 3.6x the records' bytes at the memory controller (profiles/r02_bench4_pmc_*), 1.13 ms per 781 K crystals, a fifth of configs[4]'s
// step.  The same steps across a team (geom::SolveHex / BuildPrismShape / EmitFace / FinalizeSlabs, statement by statement):
//   scalars     lane q draws scalar q (height, six face distances)                         (geom::DrawShapeScalarOne)
//   corners     lane t intersects side pairs t and t + 8 (12 in all) and tests them against the other four sides; the feasible ones are then taken IN ORDER
//               (ballot, lowest lane first), broadcast and tested against the kept corners — the serial duplicate filter, unchanged
//   sides       lane i < 6 counts the kept corners on side i; ring corner k = lane k's intersection of consecutive present sides
//   triangles   lane t (and t + 8, t + 16) builds fan triangle t of the face it belongs to; lanes 0..7 the face rows; lanes 2..4 the slabs
// Every number comes from the same expression on the same operands as on the host (this file is compiled without contraction), so
// the record is bit-identical to the host builder's (tests/test_gpu_parity.py::test_device_crystal_generator_equals_host_builder).
constexpr int kPTeam = 8, kPTeamsPerBlock = 16, kPTeamBlock = kPTeam * kPTeamsPerBlock;   // 128 threads: 16 crystals, 22.8 KB of LDS
struct PrismTeamLds {
  __attribute__((aligned(16))) ShapePrism rec;
  float c[6][2];   // ring corners (float, as the tables take them)
};
__device__ __forceinline__ uint32_t pteam_ballot(bool p) {
  const unsigned long long b = __ballot(p);
  return static_cast<uint32_t>(b >> (threadIdx.x & 56u)) & 0xFFu;
}
__device__ __forceinline__ double pteam_bcast(double v, int src) { return __shfl(v, src, kPTeam); }

__global__ void __launch_bounds__(kPTeamBlock) halo_prismgen_team_kernel(ShapePrism* __restrict__ pool, uint32_t n_crystals, uint32_t seed, const geom::CrystalRecipe rc,
                                                                        uint64_t first_index) {
  __shared__ PrismTeamLds s_team[kPTeamsPerBlock];
  const int lane = static_cast<int>(threadIdx.x & 7u);
  PrismTeamLds& T = s_team[threadIdx.x >> 3];
  const uint32_t crystal = blockIdx.x * kPTeamsPerBlock + (threadIdx.x >> 3);
  const bool live = crystal < n_crystals;   // team-uniform
  {   // the record starts as zeros (rows beyond the counts are never read, but the pool then holds no stale bytes either)
    float4* z = reinterpret_cast<float4*>(&T.rec);
    for (uint32_t i = static_cast<uint32_t>(lane); i < sizeof(ShapePrism) / 16u; i += kPTeam) z[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  // --- shape scalars: lane 0 draws the height (slot 0), lanes 1..6 the six face distances (slots 3..8) ---
  float h_raw, dist[6];
  {
    const float mine = lane < 7 ? geom::DrawShapeScalarOne(seed, rc, first_index + (live ? crystal : 0u), lane == 0 ? 0 : lane + 2) : 0.0f;
    h_raw = __shfl(mine, 0, kPTeam);
#pragma unroll
    for (int i = 0; i < 6; i++) dist[i] = __shfl(mine, 1 + i, kPTeam);
  }
  const float h = fabsf(h_raw);
  bool valid = live && (h > geom::kGeomFloatEps);
  const double k_r = geom::kGeomSqrt3 / 4.0, k_d = geom::kGeomSqrt3 / 8.0;
  double r[6];
#pragma unroll
  for (int i = 0; i < 6; i++) r[i] = k_r * static_cast<double>(dist[i]);
  // --- geom::SolveHex ---
  double scale = 0.0;
#pragma unroll
  for (int i = 0; i < 6; i++) scale = fmax(scale, fabs(r[i]));
  const double tol = 5.0 * static_cast<double>(geom::kGeomFloatEps) * scale;
  // (the 60-degree tables by selects: a table indexed by a lane-varying value would live in scratch)
  auto c6 = [](int i) { return i == 0 ? 1.0 : i == 1 ? 0.5 : i == 2 ? -0.5 : i == 3 ? -1.0 : i == 4 ? -0.5 : 0.5; };
  auto s6 = [](int i) {
    const double sv = 0.86602540378443864676;
    return (i == 1 || i == 2) ? sv : (i == 4 || i == 5) ? -sv : 0.0;
  };
  auto rsel = [&](int i) {   // r[i] without a private array
    double v = 0.0;
#pragma unroll
    for (int m = 0; m < 6; m++) v = (m == i) ? r[m] : v;
    return v;
  };
  // candidate t: the t-th pair i < j, j != i + 3, in the serial loop's order: (0,1) (0,2) (0,4) (0,5) (1,2) (1,3) (1,5) (2,3) (2,4) (3,4) (3,5) (4,5),
  // a nibble each; lane t takes candidates t and t + 8
  geom::Pt2 q[2] = {{0.0, 0.0}, {0.0, 0.0}};
  bool feasible[2] = {false, false};
#pragma unroll
  for (int round = 0; round < 2; round++) {
    const int t = lane + 8 * round;
    if (valid && t < 12) {
      const int i = static_cast<int>((0x433221110000ull >> (4 * t)) & 15ull), j = static_cast<int>((0x554435325421ull >> (4 * t)) & 15ull);
      const double ri = rsel(i), rj = rsel(j);
      const double det = c6(i) * s6(j) - s6(i) * c6(j);   // geom::Meet (never 0 for these pairs)
      q[round].x = (ri * s6(j) - rj * s6(i)) / det;
      q[round].y = (c6(i) * rj - c6(j) * ri) / det;
      bool ok = true;
#pragma unroll
      for (int m = 0; m < 6; m++)
        if (m != i && m != j && geom::Cos6(m) * q[round].x + geom::Sin6(m) * q[round].y > r[m] + tol) ok = false;
      feasible[round] = ok;
    }
  }
  // kept corners: lane c holds corners c and c + 8 (at most 12); the feasible candidates join in the serial order
  int nc = 0;
  double kx[2] = {0.0, 0.0}, ky[2] = {0.0, 0.0};
#pragma unroll
  for (int round = 0; round < 2; round++) {
    uint32_t todo = pteam_ballot(feasible[round]);
    while (__ballot(todo != 0u) != 0ull) {   // (the eight teams of a wave may differ: the loop runs for the longest list)
      const bool mine_left = todo != 0u;
      const int src = mine_left ? __ffs(todo) - 1 : 0;
      todo &= todo - 1u;
      const double cx = pteam_bcast(q[round].x, src), cy = pteam_bcast(q[round].y, src);
      const bool dup = mine_left && ((lane < nc && within((kx[0] - cx) * (kx[0] - cx) + (ky[0] - cy) * (ky[0] - cy), tol)) ||
                                     (lane + kPTeam < nc && within((kx[1] - cx) * (kx[1] - cx) + (ky[1] - cy) * (ky[1] - cy), tol)));
      if (mine_left && pteam_ballot(dup) == 0u && nc < 12) {
        if (lane == (nc & (kPTeam - 1))) {
          if (nc < kPTeam) {
            kx[0] = cx;
            ky[0] = cy;
          } else {
            kx[1] = cx;
            ky[1] = cy;
          }
        }
        nc++;
      }
    }
  }
  // a side is present iff at least two kept corners sit on it
  int on = 0;
  {
    int nc_max = nc;
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) nc_max = max(nc_max, __shfl_xor(nc_max, off));   // uniform trip count over the wave's teams
    const double ci = c6(lane), si = s6(lane), rr = rsel(lane);   // (lanes 6, 7: unused values)
    for (int c = 0; c < nc_max; c++) {
      const double cx = c < kPTeam ? pteam_bcast(kx[0], c) : pteam_bcast(kx[1], c - kPTeam);
      const double cy = c < kPTeam ? pteam_bcast(ky[0], c) : pteam_bcast(ky[1], c - kPTeam);
      if (c < nc && lane < 6 && fabs(ci * cx + si * cy - rr) <= tol) on++;
    }
  }
  const uint32_t pmask = pteam_ballot(valid && lane < 6 && on >= 2);   // present sides
  const int n = __popc(pmask);
  // walk the present sides in order: side_k = the k-th present side
  auto kth = [&](int k) {   // index of the k-th set bit of pmask (k < n)
    uint32_t m = pmask;
    for (int t = 0; t < k; t++) m &= m - 1u;
    return m ? __ffs(m) - 1 : 0;
  };
  const int sk = lane < n ? kth(lane) : 0, sk1 = lane < n ? kth((lane + 1) % max(n, 1)) : 0;
  const bool opp = lane < n && ((sk - sk1) == 3 || (sk - sk1) == -3);
  const bool bounded = n >= 3 && pteam_ballot(opp) == 0u;
  valid = valid && bounded;   // (n < 3 or unbounded: the empty crystal, counts stay 0)
  if (valid && lane < n) {    // ring corner k: consecutive present sides meet
    const double ri = rsel(sk), rj = rsel(sk1);
    const double det = c6(sk) * s6(sk1) - s6(sk) * c6(sk1);
    geom::Pt2 g{0.0, 0.0};
    if (det != 0.0) {
      g.x = (ri * s6(sk1) - rj * s6(sk)) / det;
      g.y = (c6(sk) * rj - c6(sk1) * ri) / det;
    }
    T.c[lane][0] = static_cast<float>(g.x);
    T.c[lane][1] = static_cast<float>(g.y);
  }
  team_publish();
  // --- geom::BuildPrismShape's emission ---
  if (valid) {
    const float zt = 0.5f * h, zb = -0.5f * h;
    ShapePrism& R = T.rec;
    // face rows: lane 0 top (number 1), lane 1 bottom (2), lane 2 + i side i (3 + i) at compact id 2 + rank(i)
    float dn_mine = 0.0f;
    int fid_mine = -1;
    {
      const int i = lane - 2;
      const bool side = lane >= 2;
      if (!side || ((pmask >> i) & 1u)) {
        float plane[4], nrm[3];
        if (!side) {
          plane[0] = 0.0f, plane[1] = 0.0f, plane[2] = lane == 0 ? 1.0f : -1.0f, plane[3] = -zt;
          nrm[0] = 0.0f, nrm[1] = 0.0f, nrm[2] = plane[2];
          fid_mine = lane;
        } else {
          nrm[0] = static_cast<float>(c6(i)), nrm[1] = static_cast<float>(s6(i)), nrm[2] = 0.0f;
          plane[0] = 0.5f * static_cast<float>(c6(i)), plane[1] = 0.5f * static_cast<float>(s6(i)), plane[2] = 0.0f;
          float di = 0.0f;
#pragma unroll
          for (int m = 0; m < 6; m++) di = (m == i) ? dist[m] : di;
          plane[3] = -static_cast<float>(k_d * static_cast<double>(di));
          fid_mine = 2 + __popc(pmask & ((1u << i) - 1u));
        }
        const float len = sqrtf(plane[0] * plane[0] + plane[1] * plane[1] + plane[2] * plane[2]);   // geom::EmitFace
        dn_mine = (len > geom::kGeomFloatEps) ? plane[3] / len : 0.0f;
        R.face[fid_mine][0] = nrm[0];
        R.face[fid_mine][1] = nrm[1];
        R.face[fid_mine][2] = nrm[2];
        R.face[fid_mine][3] = dn_mine;
        R.face_number[fid_mine] = static_cast<uint8_t>(side ? 3 + i : 1 + lane);
      }
    }
    // fan triangles, in the serial order: top (n - 2), bottom (n - 2), then two per present side
    const int nb = n - 2, total = 4 * n - 4;
    for (int t = lane; t < total; t += kPTeam) {
      float l0[3], l1[3], l2[3];
      int fid;
      if (t < 2 * nb) {
        const bool top = t < nb;
        const int k = (top ? t : t - nb) + 1;
        const int i0 = top ? 0 : n - 1, i1 = top ? k : n - 1 - k, i2 = top ? k + 1 : n - 2 - k;
        const float z = top ? zt : zb;
        l0[0] = T.c[i0][0], l0[1] = T.c[i0][1], l0[2] = z;
        l1[0] = T.c[i1][0], l1[1] = T.c[i1][1], l1[2] = z;
        l2[0] = T.c[i2][0], l2[1] = T.c[i2][1], l2[2] = z;
        fid = top ? 0 : 1;
      } else {
        const int sidx = (t - 2 * nb) >> 1, half = (t - 2 * nb) & 1;
        const float* a = T.c[(sidx - 1 + n) % n];
        const float* b = T.c[sidx];
        // loop = (a, zb), (b, zb), (b, zt), (a, zt); triangles (0, 1, 2) and (0, 2, 3)
        l0[0] = a[0], l0[1] = a[1], l0[2] = zb;
        if (half == 0) {
          l1[0] = b[0], l1[1] = b[1], l1[2] = zb;
          l2[0] = b[0], l2[1] = b[1], l2[2] = zt;
        } else {
          l1[0] = b[0], l1[1] = b[1], l1[2] = zt;
          l2[0] = a[0], l2[1] = a[1], l2[2] = zt;
        }
        fid = 2 + sidx;
      }
      float* v = R.tri_v[t];
      for (int a3 = 0; a3 < 3; a3++) {
        v[a3] = l0[a3];
        v[3 + a3] = l1[a3];
        v[6 + a3] = l2[a3];
      }
      const float ea[3] = {l1[0] - l0[0], l1[1] - l0[1], l1[2] - l0[2]};
      const float eb[3] = {l2[0] - l0[0], l2[1] - l0[1], l2[2] - l0[2]};
      const float nrm[3] = {-eb[1] * ea[2] + ea[1] * eb[2], eb[0] * ea[2] - ea[0] * eb[2], -eb[0] * ea[1] + ea[0] * eb[1]};  // Cross3 math.cpp:36
      const float mag = sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
      R.tri_na[t][3] = mag / 2.0f;
      for (int c3 = 0; c3 < 3; c3++) R.tri_na[t][c3] = (mag > 0.0f) ? nrm[c3] / mag : 0.0f;
      R.tri_face[t] = static_cast<uint8_t>(fid);
    }
    // geom::FinalizeSlabs: slab 0 = the basal pair, then sides i < 3 whose opposite i + 3 is present too, in side order; the rest single
    const uint32_t pair = pmask & (pmask >> 3) & 7u;              // bit i: sides i and i + 3 both present
    const uint32_t single = pmask & ~(pair | (pair << 3));
    const float dn_opp = __shfl(dn_mine, (lane + 3) & 7, kPTeam);   // side i + 3 sits in lane i + 5
    const float dn_bot = __shfl(dn_mine, 1, kPTeam);
    const int fid_opp = __shfl(fid_mine, (lane + 3) & 7, kPTeam);
    if (lane == 0) {
      float* s0 = R.slab[0];
      s0[0] = 0.0f, s0[1] = 0.0f, s0[2] = 1.0f, s0[3] = dn_mine, s0[4] = dn_bot;
      reinterpret_cast<uint32_t*>(s0)[5] = 0u;
      reinterpret_cast<uint32_t*>(s0)[6] = 1u;
      s0[7] = 0.0f;
      R.face_cnt = 2 + n;
      R.tri_cnt = min(total, static_cast<int>(sizeof(R.tri_na) / 16u));
      R.slab_cnt = 1 + __popc(pair);
      R.single_cnt = __popc(single);
    }
    if (lane >= 2) {
      const int i = lane - 2;
      if (i < 3 && ((pair >> i) & 1u)) {
        float* sr = R.slab[1 + __popc(pair & ((1u << i) - 1u))];
        sr[0] = static_cast<float>(c6(i)), sr[1] = static_cast<float>(s6(i)), sr[2] = 0.0f, sr[3] = dn_mine, sr[4] = dn_opp;
        reinterpret_cast<uint32_t*>(sr)[5] = static_cast<uint32_t>(fid_mine);
        reinterpret_cast<uint32_t*>(sr)[6] = static_cast<uint32_t>(fid_opp);
        sr[7] = 0.0f;
      }
      if ((single >> i) & 1u) R.single[__popc(single & ((1u << i) - 1u))] = static_cast<uint8_t>(fid_mine);
    }
  }
  team_publish();
  if (live) {   // the record leaves in 16-byte pieces, 128 contiguous bytes per team and round
    const float4* src = reinterpret_cast<const float4*>(&T.rec);
    float4* dst = reinterpret_cast<float4*>(pool + crystal);
    for (uint32_t i = static_cast<uint32_t>(lane); i < sizeof(ShapePrism) / 16u; i += kPTeam) dst[i] = src[i];
  }
}

hipError_t launch_shapegen(void* pool, bool prism_records, uint32_t n, uint32_t seed, const geom::CrystalRecipe& rc, uint64_t first_index,
                           hipStream_t stream, bool serial_pyramid) {
  if (n == 0) return hipSuccess;
  const dim3 grid((n + kGenBlock - 1) / kGenBlock), block(kGenBlock);
  if (prism_records && rc.c.kind == HALO_CRYSTAL_PRISM && !serial_pyramid)
    hipLaunchKernelGGL(halo_prismgen_team_kernel, dim3((n + kPTeamsPerBlock - 1) / kPTeamsPerBlock), dim3(kPTeamBlock), 0, stream, static_cast<ShapePrism*>(pool), n, seed, rc,
                       first_index);
  else if (prism_records) hipLaunchKernelGGL(halo_shapegen_kernel<ShapePrism>, grid, block, 0, stream, static_cast<ShapePrism*>(pool), n, seed, rc, first_index);
  else if (rc.c.kind == HALO_CRYSTAL_PYRAMID && !serial_pyramid)
    hipLaunchKernelGGL(halo_pyrgen_team_kernel, dim3((n + kTeamsPerBlock - 1) / kTeamsPerBlock), dim3(kTeamBlock), 0, stream, static_cast<ShapeDev*>(pool), n, seed, rc,
                       first_index);
  else hipLaunchKernelGGL(halo_shapegen_kernel<ShapeDev>, grid, block, 0, stream, static_cast<ShapeDev*>(pool), n, seed, rc, first_index);
  return hipGetLastError();
}

}  // namespace halo
