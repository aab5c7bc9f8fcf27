// halo_trace.inl — gfx950 (CDNA4 / MI355X) kernels of the ice-halo trace hot path.
//
// One fused kernel per (scattering layer, crystal entry) dispatch:
//     root generation | layer hop  →  entry Fresnel  →  ≤ max_hits-1 interior interactions
//     →  emit gate  →  lens projection  →  CIE-XYZ accumulation  |  continuation append
// Ray state lives in VGPRs for its whole life; the crystal plane/slab/fan tables and the latitude LUT are
// staged once per workgroup (or, for sampled crystals, once per half-wave pass) into LDS and read back as
// wave-wide broadcasts (ds_read_b128, conflict-free); the one-shape production kernels queue their exits per wave and run
// projection + accumulation on full batches; HBM sees only the accumulation traffic (the hit log's 8-byte records for big
// launches, else atomics on line-decorrelated planes or binned hit lists) and — for multi-scatter layers — the 20-byte SoA
// continuation record, appended with wave64 ballot compaction into sharded regions and gathered by the next
// layer through the Feistel permutation, so neither the reference's 80 B/ray root buffers nor its separate
// gen / transit / shuffle kernels exist here.
//
// What each block restates (reference /root/reference, file:line):
//   PCG streams, orientation, sun cone, entry pick   src/core/shared/pcg_shared.h:193-624,
//                                                     cuda_trace_backend.cu:1417-1616 (gen), :1220-1403 (transit)
//   Fresnel split / refraction                        src/core/optics.cpp:18-53, shared/optics_shared.h:17-24
//   convex slab traversal                             src/core/optics.cpp:64-158, shared/traversal_shared.h:61-71
//   hit loop + emit gate (legacy semantics)           src/core/simulator.cpp:585-762, 1308-1336
//   projection (11 lenses)                            src/core/shared/projection_shared.h:42-375
//   XYZ accumulation + landed weight                  shared/accum_shared.h:40-47, cuda_trace_backend.cu:433-480
//   continuation permutation                          pcg_shared.h:550-603, cuda_trace_backend.cu:1633-1657
#pragma once
#include <hip/hip_runtime.h>

#include "halo_device.h"

namespace halo {

#define HD __device__ __forceinline__

// Phase probe (tools/phase_probe.py; builds with -DHALO_PROBE only — never the shipped library): every wave stamps the shader
// clock at phase boundaries and the kernel adds the per-wave cycle sums to g_halo_probe.  With several waves interleaving on
// a SIMD a phase's elapsed time is its share of the SIMD's issue slots, which is what "where do the cycles go" asks.
#ifdef HALO_PROBE
__device__ unsigned long long g_halo_probe[16];
struct Probe {
  uint64_t t0;
  uint32_t acc[16];
};
HD void probe_start(Probe& pr) {
  __builtin_amdgcn_sched_barrier(0);
  pr.t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
}
template <int K>
HD void probe_mark(Probe& pr) {
  __builtin_amdgcn_sched_barrier(0);
  const uint64_t t = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
  pr.acc[K] += static_cast<uint32_t>(t - pr.t0);
  pr.t0 = t;
}
#define PROBE_MARK(pr, K) probe_mark<K>(pr)
#else
struct Probe {};
#define PROBE_MARK(pr, K) ((void)0)
#endif
enum { kPhStream = 0, kPhOrient = 1, kPhRotation = 2, kPhSun = 3, kPhEntry = 4, kPhFresnel = 5, kPhEmitGate = 6, kPhProject = 7,
       kPhAccum = 8, kPhSlab = 9, kPhKernelFixed = 10, kPhTotal = 11, kPhStage = 12, kPhFlush = 13, kPhPrologue = 14, kPhFinalDrain = 15 };

constexpr float kPiF = 3.14159265358979323846f;   // LM_PI_F  (lm_shims.h:84)
constexpr float kPi2F = 1.5707963267948966f;      // LM_PI_2F (lm_shims.h:85)
constexpr float kSlabEps = 1e-5f;                 // traversal_shared.h:46

// Separately-rounded multiply / add that the compiler cannot re-fuse into an FMA (the backend fuses any
// fmul+fadd pair under -ffp-contract=fast, whatever the source says).  Used only where the reference's formula
// cancels catastrophically and a fused product would move the result by orders of magnitude more than an ulp.
HD float mul_rn(float a, float b) {
  float r;
  asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
HD float add_rn(float a, float b) {
  float r;
  asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// 1-ulp hardware reciprocal / square root (v_rcp_f32, v_sqrt_f32) for the inner-loop quotients whose last bit does
// not steer a discrete decision; IEEE division costs ~10 VALU ops here and the loop had nine of them per hit.
// Every sum of products on the ray's way is spelled as the fma chain it is evaluated as (round 4: the instantiations of one kernel must trace
// bit-identical rays whatever code surrounds the expression).  HALO_STRICT builds — the tested variant `libhalo_hip_strict.so`, compiled with
// -ffp-contract=off and HALO_FRESNEL=1 — spell the same chains as separately rounded products and sums in the same order: the REFERENCE's
// roundings on a host without FMA contraction, which is what the oracle computes; tests/test_gpu_strict_variant.py holds that build to the
// unconditioned per-ray bars.
#ifndef HALO_STRICT
#define HALO_STRICT 0
#endif
#if HALO_STRICT
#define HALO_FMA(a, b, c) ((a) * (b) + (c))
#else
#define HALO_FMA(a, b, c) fmaf((a), (b), (c))
#endif
HD float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
HD float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
HD float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

// The Fresnel split's quotients and root (HALO_FRESNEL: 0 = hardware approximations, 1 = IEEE division / square root, 2 = the
// approximations with one FMA refinement step — the correctly rounded quotient for operands in the normal range, which these are).
// dd = (1 - rr^2) / cos^2 + rr^2 cancels near the critical angle: a 1-ulp reciprocal there moves the transmitted weight by 1e-3.
#ifndef HALO_FRESNEL
#define HALO_FRESNEL 0
#endif
HD float fresnel_div(float a, float b) {
#if HALO_FRESNEL == 0
  return a * fast_rcp(b);
#elif HALO_FRESNEL == 1
  return __fdiv_rn(a, b);
#else
  const float y = fast_rcp(b);
  const float q = a * y;
  return fmaf(fmaf(-b, q, a), y, q);
#endif
}
HD float fresnel_sqrt(float x) {   // x >= 0
#if HALO_FRESNEL == 0
  return fast_sqrt(x);
#elif HALO_FRESNEL == 1
  return __fsqrt_rn(x);
#else
  const float xc = fmaxf(x, 1e-30f);   // rsq(0) is inf; sqrt(1e-30) is 1e-15, below half an ulp of everything it is added to here
  const float y = fast_rsq(xc);
  const float sq = xc * y;
  return fmaf(fmaf(-sq, sq, xc), 0.5f * y, sq);
#endif
}

// sin and cos together for |x| below a few turns (every angle on this path is): two-term Cody-Waite reduction by
// pi/2 with FMAs, then the classic degree-7 / degree-8 minimax kernels on [-pi/4, pi/4].  ~1 ulp; replaces the
// library sincosf whose general-argument reduction dominated root generation.
HD void sincos_small(float x, float* sn, float* cs) {
  const float k = rintf(x * 0.6366197466850281f);
  float r = fmaf(k, -1.5707963705062866f, x);
  r = fmaf(k, 4.371138828673793e-08f, r);
  const int q = static_cast<int>(k);
  const float z = r * r;
  const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
  const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                        fmaf(-0.5f, z, 1.0f));
  const bool swap = (q & 1) != 0;
  float so = swap ? cp : sp;
  float co = swap ? sp : cp;
  // quadrant signs as sign-bit flips (bit 1 of q and of q + 1 moved to bit 31): a shift and one three-input bit op each, no compare / select
  const uint32_t uq = static_cast<uint32_t>(q);
  *sn = __uint_as_float(__float_as_uint(so) ^ ((uq << 30) & 0x80000000u));
  *cs = __uint_as_float(__float_as_uint(co) ^ (((uq + 1u) << 30) & 0x80000000u));
}

// ------------------------------------------------------------------------------------------------
// counter-based RNG (pcg_shared.h:193-274)
// ------------------------------------------------------------------------------------------------
HD uint32_t pcg_hash(uint32_t x) {
  x = x * 747796405u + 2891336453u;
  x = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
  return (x >> 22u) ^ x;
}

struct Stream {
  uint32_t seed;
  uint32_t key;   // global_idx * 1000003u, hoisted
  uint32_t slot;
};

HD Stream make_stream(uint32_t seed, uint32_t lo, uint32_t hi, uint32_t tid) {
  // pcg_advance_hi + pcg_seed_with_high (pcg_shared.h:257-268): the 64-bit ray index = hi:lo + tid
  uint32_t g = lo + tid;
  uint32_t h = hi + ((g < lo) ? 1u : 0u);
  Stream s;
  s.seed = (h == 0u) ? seed : (seed ^ pcg_hash(h));
  s.key = g * 1000003u;
  s.slot = 0u;
  return s;
}

HD float uniform(Stream& s) {
  uint32_t h = pcg_hash(s.seed ^ pcg_hash(s.key + s.slot));
  s.slot++;
  return static_cast<float>(h >> 8) * (1.0f / 16777216.0f);
}

HD float gaussian(Stream& s) {  // Box-Muller, pcg_shared.h:277-281
  float u1 = fmaxf(uniform(s), 1e-7f);
  float u2 = uniform(s);
  return sqrtf(-2.0f * logf(u1)) * cosf(2.0f * kPiF * u2);
}

HD float get_dist(Stream& s, uint32_t dtype, float mean, float spread) {  // pcg_shared.h:290-308
  if (dtype == HALO_DIST_NONE) return mean;
  if (dtype == HALO_DIST_UNIFORM) return HALO_FMA(uniform(s) - 0.5f, spread, mean);
  if (dtype == HALO_DIST_GAUSS || dtype == HALO_DIST_GAUSS_LEGACY) return HALO_FMA(gaussian(s), spread, mean);
  if (dtype == HALO_DIST_ZIGZAG) return fabsf(HALO_FMA(spread, sinf(uniform(s) * 2.0f * kPiF), mean));
  float u = uniform(s);
  float sgn = (u < 0.5f) ? -1.0f : 1.0f;
  float arg = fmaxf(1.0f - 2.0f * fabsf(u - 0.5f), 1e-30f);
  return HALO_FMA(-(spread * sgn), logf(arg), mean);
}

// 4-round balanced Feistel + cycle walk on [0, n)  (pcg_shared.h:550-603)
HD uint32_t feistel_bijection(uint32_t i, uint32_t n, uint32_t seed) {
  if (n <= 1u) return i;
  if (n == 2u) return i ^ 1u;
  uint32_t bits = 32u - __clz(n - 1u);  // smallest b with 2^b >= n (n >= 3)
  if (bits > 30u) bits = 30u;
  if (bits & 1u) bits++;
  const uint32_t half_bits = bits >> 1u;
  const uint32_t hm = (1u << half_bits) - 1u;
  uint32_t cur = i;
  for (uint32_t guard = 0u; guard < 64u; guard++) {
    uint32_t L = (cur >> half_bits) & hm;
    uint32_t R = cur & hm;
    const uint32_t rc[4] = {0x9E3779B9u, 0x85EBCA6Bu, 0xC2B2AE35u, 0x27D4EB2Fu};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t f = pcg_hash(seed ^ R ^ rc[k]) & hm;
      uint32_t nr = L ^ f;
      L = R;
      R = nr;
    }
    uint32_t out = (L << half_bits) | R;
    if (out < n) return out;
    cur = out;
  }
  return cur % n;
}

// ------------------------------------------------------------------------------------------------
// orientation (pcg_shared.h:311-483)
// ------------------------------------------------------------------------------------------------
HD void normalize_latitude(float phi, float& phi_out, bool& flip) {
  float theta = kPi2F - phi;
  theta = fmodf(theta, 2.0f * kPiF);
  if (theta < 0.0f) theta += 2.0f * kPiF;
  flip = theta > kPiF;
  if (flip) theta = 2.0f * kPiF - theta;
  phi_out = kPi2F - theta;
}

// LUT in LDS: [0..256] theta, [257..513] cdf, [514..770] flip
HD float invert_lat_lut(float xi, const float* lut) {
  const float* th = lut;
  const float* cdf = lut + kLutNodes;
  xi = fminf(fmaxf(xi, cdf[0]), cdf[kLutNodes - 1]);
  // The reference's bisection (lo = 0, hi = 256; mid = (lo + hi) / 2; cdf[mid] <= xi ? lo = mid : hi = mid; 8 rounds) has hi - lo = 256 >> round
  // whatever the data, so mid is always lo + (128 >> round) and hi need not be carried: the same probes, the same comparisons, the same lo —
  // in three instructions a round (compare, select, add; the read is at a constant offset from lo) where carrying both ends took seven.
  static_assert(kLutNodes == 257, "the bisection below is written for 256 intervals");
  uint32_t lo4 = 0u;   // lo as a byte offset: the probe's address is lo4 plus a constant, no shift per round
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const uint32_t step4 = 512u >> it;
    lo4 += (*reinterpret_cast<const float*>(reinterpret_cast<const char*>(cdf) + lo4 + step4) <= xi) ? step4 : 0u;
  }
  const uint32_t lo = lo4 >> 2;
  float c0 = cdf[lo], c1 = cdf[lo + 1u];
  float denom = c1 - c0;
  float w = denom > 0.0f ? (xi - c0) * fast_rcp(denom) : 0.0f;
  return HALO_FMA(w, th[lo + 1u] - th[lo], th[lo]);
}

HD uint32_t lat_lut_bin(float theta, const float* lut) {
  float span = lut[kLutNodes - 1] - lut[0];
  float t = span > 0.0f ? (theta - lut[0]) * fast_rcp(span) : 0.0f;
  int idx = static_cast<int>(t * static_cast<float>(kLutNodes - 1));
  idx = idx < 0 ? 0 : idx;
  idx = idx > kLutNodes - 2 ? kLutNodes - 2 : idx;
  return static_cast<uint32_t>(idx);
}

HD void sample_lat_lon_roll(Stream& s, const DispatchParams& P, const float* lut, float& lon, float& lat, float& roll) {
  float phi = 0.0f;
  bool flip = false;
  lon = 0.0f;
  if (P.lat_path == kLatFullSphere) {
    float u = uniform(s) * 2.0f - 1.0f;
    u = fminf(fmaxf(u, -1.0f), 1.0f);
    phi = asinf(u);
    lon = uniform(s) * 2.0f * kPiF;
  } else if (P.lat_path == kLatNoRandom) {
    phi = P.lat_mean_rad;
  } else if (P.lat_path == kLatGaussLegacy) {
    float raw = get_dist(s, HALO_DIST_GAUSS_LEGACY, P.lat_mean_rad, P.lat_std_rad);
    normalize_latitude(raw, phi, flip);
  } else {  // kLatLut
    float xi = uniform(s);
    float colat = invert_lat_lut(xi, lut);
    phi = kPi2F - colat;
    uint32_t bin = lat_lut_bin(colat, lut);
    const float p_flip = lut[2 * kLutNodes + bin];
    if (p_flip > 0.0f) {
      flip = uniform(s) < p_flip;
    } else {
      s.slot++;  // u < 0 is false for every u: keep the stream aligned, skip the two hashes
    }
  }
  if (P.lat_path != kLatFullSphere) lon = get_dist(s, P.az_type, P.az_mean_rad, P.az_std_rad);
  roll = get_dist(s, P.roll_type, P.roll_mean_rad, P.roll_std_rad);
  if (flip) {
    lon += kPiF;
    roll += kPiF;
  }
  lat = phi;
}

// R = Rz(lon - pi) * Ry(lat - pi/2) * Rz(roll), row-major (pcg_shared.h:441-483).  Evaluated in the sparse
// form the dense axis-angle / 3x3 chain reduces to; `k = (1 - c) + c` keeps the reference's diagonal term.
HD void build_crystal_rotation(float lon, float lat, float roll, float* R) {
  float s1, c1, s2, c2, s3, c3;
  sincos_small(roll, &s1, &c1);
  sincos_small(lat - kPi2F, &s2, &c2);
  sincos_small(lon - kPiF, &s3, &c3);
  float k1 = (1.0f - c1) + c1, k2 = (1.0f - c2) + c2, k3 = (1.0f - c3) + c3;
  // t = Ry * Rz(roll)
  float t00 = c2 * c1, t01 = c2 * (-s1), t02 = s2 * k1;
  float t10 = k2 * s1, t11 = k2 * c1;  // t12 = 0
  float t20 = (-s2) * c1, t21 = (-s2) * (-s1), t22 = c2 * k1;
  // (every sum of products on the ray's way is written as the fma chain it is evaluated as: which of a*b + c*d's two products gets fused is
  // otherwise the backend's choice, and it follows the code AROUND the expression — instantiations of one kernel then trace rays that differ
  // in the last bit, and one in a few million of them takes the other side of a decision; see dot3_fma)
  R[0] = HALO_FMA(-s3, t10, c3 * t00);
  R[1] = HALO_FMA(-s3, t11, c3 * t01);
  R[2] = c3 * t02;
  R[3] = HALO_FMA(c3, t10, s3 * t00);
  R[4] = HALO_FMA(c3, t11, s3 * t01);
  R[5] = s3 * t02;
  R[6] = k3 * t20;
  R[7] = k3 * t21;
  R[8] = k3 * t22;
}

HD void apply_inverse(const float* R, float x, float y, float z, float* o) {  // o = R^T v
  o[0] = HALO_FMA(R[6], z, HALO_FMA(R[3], y, R[0] * x));
  o[1] = HALO_FMA(R[7], z, HALO_FMA(R[4], y, R[1] * x));
  o[2] = HALO_FMA(R[8], z, HALO_FMA(R[5], y, R[2] * x));
}

// ------------------------------------------------------------------------------------------------
// projection (projection_shared.h:42-375); proj_type is dispatch-uniform → scalar branches
// ------------------------------------------------------------------------------------------------
struct XY {
  float x, y;
  bool valid;
};

HD XY fisheye_equal_area(float dx, float dy, float dz, float rs) {
  // v_rsq_f32 (1 ulp) instead of an IEEE sqrt + divide (~25 instructions, once per emitted exit): moves a hit by <= 1e-4 px
  float k = rs * fast_rsq(1.0f + fminf(fmaxf(dz, -1.0f + 1e-6f), 1.0f));
  return {k * dx, k * dy, true};
}
HD XY fisheye_equidistant(float dx, float dy, float dz, float rs) {
  float rho = sqrtf(dx * dx + dy * dy);
  if (rho < 1e-10f) return {0.0f, 0.0f, true};
  float theta = acosf(fminf(fmaxf(dz, -1.0f), 1.0f));
  float sc = rs * theta / (kPi2F * rho);
  return {sc * dx, sc * dy, true};
}
HD XY fisheye_stereographic(float dx, float dy, float dz, float rs) {
  float rho = sqrtf(dx * dx + dy * dy);
  if (rho < 1e-10f) return {0.0f, 0.0f, true};
  float theta = acosf(fminf(fmaxf(dz, -1.0f), 1.0f));
  float sc = rs * tanf(theta / 2.0f) / rho;
  return {sc * dx, sc * dy, true};
}
HD XY fisheye_orthographic(float dx, float dy, float dz, float rs) {
  if (dz < 0.0f) return {0.0f, 0.0f, false};
  return {rs * dx, rs * dy, true};
}
HD XY dual_forward(int t, float sx, float sy, float z, float rs) {
  if (t == HALO_LENS_DUAL_FISHEYE_EQUAL_AREA) return fisheye_equal_area(sx, sy, z, rs);
  if (t == HALO_LENS_DUAL_FISHEYE_EQUIDISTANT) return fisheye_equidistant(sx, sy, z, rs);
  if (t == HALO_LENS_DUAL_FISHEYE_STEREOGRAPHIC) return fisheye_stereographic(sx, sy, z, rs);
  return fisheye_orthographic(sx, sy, z, rs);
}
HD void dual_to_pixel(float xn, float yn, bool upper, int w, int h, float& fx, float& fy) {
  int half_w = w / 2;
  int short_res = half_w < h ? half_w : h;
  float r = static_cast<float>(short_res) / 2.0f;
  float cy = static_cast<float>(h) / 2.0f;
  float cx = upper ? static_cast<float>(w) / 2.0f - r : static_cast<float>(w) / 2.0f + r;
  fx = HALO_FMA(upper ? -yn : yn, r, cx);
  fy = HALO_FMA(xn, r, cy);
}

struct Hits {
  int px0, py0, px1, py1;
  int count;
};

HD Hits project_exit(const ProjDev& p, float wx, float wy, float wz, int lens = -1, int vis = -1) {   // lens / vis >= 0: the dispatch's lens / visible range, known at compile time
  Hits r;
  r.count = 0;
  r.px0 = r.py0 = r.px1 = r.py1 = 0;
  const int t = lens >= 0 ? lens : p.proj_type;
  const int vr = vis >= 0 ? vis : p.visible_range;
  if (t == HALO_LENS_LINEAR || t == HALO_LENS_FISHEYE_EQUAL_AREA || t == HALO_LENS_FISHEYE_EQUIDISTANT ||
      t == HALO_LENS_FISHEYE_STEREOGRAPHIC || t == HALO_LENS_FISHEYE_ORTHOGRAPHIC) {
    if ((vr == HALO_VISIBLE_UPPER && wz > 0.0f) || (vr == HALO_VISIBLE_LOWER && wz < 0.0f)) return r;
    // (every sum of products below is spelled as the chain it is evaluated as — like the cull, exit_may_land — so that the instantiations that
    //  take the projection from SGPRs and the ones that take it from LDS into VGPRs (HALO_PROJ_LDS) put a hit on the same side of a pixel edge:
    //  left to the compiler's contraction the two differed on one hit in 5 M, tests/test_gpu_production_routes.py::test_hit_log_route_equals_…)
    float cx = HALO_FMA(p.rot[6], -wz, HALO_FMA(p.rot[3], -wy, p.rot[0] * (-wx)));
    float cy = HALO_FMA(p.rot[7], -wz, HALO_FMA(p.rot[4], -wy, p.rot[1] * (-wx)));
    float cz = HALO_FMA(p.rot[8], -wz, HALO_FMA(p.rot[5], -wy, p.rot[2] * (-wx)));
    XY xy;
    if (t == HALO_LENS_LINEAR) {
      if (cz <= 0.0f) return r;
      xy = {cx / cz, cy / cz, true};
    } else {
      if (cz <= 0.0f) return r;
      if (t == HALO_LENS_FISHEYE_EQUAL_AREA) xy = fisheye_equal_area(cx, cy, cz, 1.0f);
      else if (t == HALO_LENS_FISHEYE_EQUIDISTANT) xy = fisheye_equidistant(cx, cy, cz, 1.0f);
      else if (t == HALO_LENS_FISHEYE_STEREOGRAPHIC) xy = fisheye_stereographic(cx, cy, cz, 1.0f);
      else xy = fisheye_orthographic(cx, cy, cz, 1.0f);
    }
    if (!xy.valid) return r;
    xy.x = -xy.x;
    const float* pre = reinterpret_cast<const float*>(&p + 1);   // ProjLds' / DispatchParams::proj_pre's four floats, directly behind the ProjDev
    const float half_w = pre[0], half_h = pre[1], shift_x = pre[2], shift_y = pre[3];
    r.px0 = static_cast<int>(floorf(HALO_FMA(xy.x, p.scale, half_w) + 0.5f + shift_x));
    r.py0 = static_cast<int>(floorf(HALO_FMA(xy.y, p.scale, half_h) + 0.5f + shift_y));
    r.count = 1;
    return r;
  }
  if (t == HALO_LENS_RECTANGULAR) {
    float lon = atan2f(-wy, -wx) - p.az0;
    float lat = asinf(fminf(fmaxf(-wz, -1.0f), 1.0f));
    while (lon < -kPiF) lon += 2.0f * kPiF;
    while (lon > kPiF) lon -= 2.0f * kPiF;
    int raw_x = static_cast<int>(floorf(HALO_FMA(lon, p.scale, static_cast<float>(p.img_w) / 2.0f) + 0.5f));
    r.px0 = ((raw_x % p.img_w) + p.img_w) % p.img_w;
    r.py0 = static_cast<int>(floorf(HALO_FMA(-lat, p.scale, static_cast<float>(p.img_h) / 2.0f) + 0.5f));
    r.count = 1;
    return r;
  }
  if (t == HALO_LENS_DUAL_FISHEYE_EQUAL_AREA || t == HALO_LENS_DUAL_FISHEYE_EQUIDISTANT ||
      t == HALO_LENS_DUAL_FISHEYE_STEREOGRAPHIC || t == HALO_LENS_DUAL_FISHEYE_ORTHOGRAPHIC) {
    float sx = -wx, sy = -wy, sz = -wz;
    bool upper = (sz >= 0.0f);
    float z_hemi = upper ? sz : -sz;
    XY xy = dual_forward(t, sx, sy, z_hemi, p.r_scale);
    float fx, fy;
    dual_to_pixel(xy.x, xy.y, upper, p.img_w, p.img_h, fx, fy);
    r.px0 = static_cast<int>(floorf(fx + 0.5f));
    r.py0 = static_cast<int>(floorf(fy + 0.5f));
    r.count = 1;
    if (p.max_abs_dz > 0.0f && fabsf(sz) < p.max_abs_dz) {
      XY xy2 = dual_forward(t, sx, sy, -z_hemi, p.r_scale);
      dual_to_pixel(xy2.x, xy2.y, !upper, p.img_w, p.img_h, fx, fy);
      r.px1 = static_cast<int>(floorf(fx + 0.5f));
      r.py1 = static_cast<int>(floorf(fy + 0.5f));
      r.count = 2;
    }
    return r;
  }
  if (t == HALO_LENS_GLOBE) {
    const float kGlobeCameraD = 4.0f;
    float cx = HALO_FMA(p.rot[6], -wz, HALO_FMA(p.rot[3], -wy, p.rot[0] * (-wx)));
    float cy = HALO_FMA(p.rot[7], -wz, HALO_FMA(p.rot[4], -wy, p.rot[1] * (-wx)));
    float cz = HALO_FMA(p.rot[8], -wz, HALO_FMA(p.rot[5], -wy, p.rot[2] * (-wx)));
    if (cz >= -1.0f / kGlobeCameraD) return r;
    float denom = kGlobeCameraD + cz;
    r.px0 = static_cast<int>(floorf(HALO_FMA(-cx / denom, p.scale, static_cast<float>(p.img_w) / 2.0f) + 0.5f + static_cast<float>(p.lens_shift_x)));
    r.py0 = static_cast<int>(floorf(HALO_FMA(cy / denom, p.scale, static_cast<float>(p.img_h) / 2.0f) + 0.5f + static_cast<float>(p.lens_shift_y)));
    r.count = 1;
    return r;
  }
  return r;
}

// ------------------------------------------------------------------------------------------------
// accumulation
// ------------------------------------------------------------------------------------------------
HD void atomic_add_f32(float* addr, float v) {
  // hardware global_atomic_add_f32 (no CAS loop); the accumulator is ordinary coarse-grained device memory
  unsafeAtomicAdd(addr, v);
}

// Measured on MI355X (tools/atomic_bench.hip): scattered fp32 atomics retire at ~20.7 G/s whatever the scope or
// buffer size, but same-cache-line atomics serialize — and a third of all exits land on the ~20 pixels of the
// sun disc (rays crossing two parallel faces keep their direction).  What keeps that off the fabric (the plane
// layout that decorrelates lines is MonoSlot in halo_device.h):
//  * MONO: in a discrete-wavelength session every exit's XYZ is cmf(lambda)*w, so the kernel accumulates the
//    scalar w into a one-channel plane (1 atomic per hit, not 3) and halo_fold_kernel applies the CMF once per
//    session (AccumXyzToPixel accum_shared.h:40-47 distributes over the sum);
//  * a per-workgroup two-way pixel cache in LDS: the first pixel to claim a slot accumulates there with
//    ds_add_f32 for the rest of the kernel and is flushed once; pixels that lose the claim go straight to HBM.
//    Frequent pixels claim early with overwhelming probability, which is all the cache is for.
typedef float float2v __attribute__((ext_vector_type(2)));
HD float2v pk_dot(float2v X, float2v Y, float2v Z, float gx, float gy, float gz) {   // (n.d, n.p) of one plane: explicit packed fma chain (see dot3_fma)
  const float2v vx = {gx, gx}, vy = {gy, gy}, vz = {gz, gz};
#if HALO_STRICT
  return Z * vz + (Y * vy + X * vx);
#else
  return __builtin_elementwise_fma(Z, vz, __builtin_elementwise_fma(Y, vy, X * vx));
#endif
}

// SMALLC: binned kernels of shape-pool dispatches halve the one-channel cache (their LDS also holds the pool slots and the
// hit buffer; misses are cheap there) to stay at 4 workgroups per CU
#ifndef HALO_CACHE_WAYS
#define HALO_CACHE_WAYS 4   // ways of the X/Y/Z pixel cache's sets; 2: the two-way cache of rounds 2-4 (A/B knob).  The scalar caches have two.
#endif
#ifndef HALO_CACHE_LOG2
#define HALO_CACHE_LOG2 11
#endif
template <bool MONO, bool SMALLC>
struct CacheGeom {  // 2048 one-channel slots (16 KB; 1024 with SMALLC) or 1024 three-channel slots (16 KB)
  static constexpr int kLog2 = (MONO && !SMALLC) ? HALO_CACHE_LOG2 : 10;
  static constexpr int kN = 1 << kLog2;
};

template <bool MONO, bool SMALLC>
struct PixCache {
  __attribute__((aligned(16))) uint32_t tag[CacheGeom<MONO, SMALLC>::kN];     // ((plane << 23) | pixel) + 1, 0 = free; a set of four is read as one uint4
  float val[CacheGeom<MONO, SMALLC>::kN * (MONO ? 1 : 3)];
};

// Binned accumulation (discrete-wavelength sessions, big launches).  A launch of n rays puts tens of hits on EVERY pixel,
// but no workgroup sees a pixel twice — the reuse only exists chip-wide, and one global atomic per hit caps the kernel at
// ~21 G hits/s.  So hits are exchanged through per-tile lists in HBM instead: a workgroup stages its hits {slot, w} in LDS,
// and when the buffer is half full bins them by image tile (16384 slots) — one returning global atomic per tile per
// flush reserves the segment, the 8-byte stores of a tile land in one or two lines — and halo_bin_accumulate_kernel then
// sums each tile's list in a 64 KB LDS tile and adds it to the plane with plain stores.  Lists that run over (a tile much
// hotter than average) fall back to the direct atomic, so capacity is a speed matter only.
constexpr int kAccDirect = 0, kAccBin = 1, kAccLog = 2, kAccNone = 3, kAccLogFinal = 4;   // halo_trace_kernel ACC (None: a layer whose every exit continues — nothing lands)
constexpr int kHitBuf = 1536;                 // staged hits per workgroup (16 KB)
constexpr uint32_t kBinTileLog2 = 14u;         // slots per tile: 64 KB of fp32 in the accumulate pass
constexpr int kBinMaxTiles = 512;
constexpr int kBinCntStride = 16;              // tile counters 64 B apart
struct HitBuffer {
  uint2 h[kHitBuf];
  uint32_t n;
  uint32_t cur[kBinMaxTiles];   // per tile: hit count of this flush, then the write cursor inside the tile's reserved segment
};
template <bool ON>
struct HitSlot {
  HitBuffer b;
};
template <>
struct HitSlot<false> {
  uint32_t unused;
};
// Exit queue (production one-shape kernels).  An interaction hands the image at most one exit per lane, and most of them go
// nowhere: configs[1] emits 4.7 exits per ray of which 1.6 are in frame (`visible: upper` and the camera's half space cull the
// rest), TIR lanes emit none — yet projection, pixel cache and miss path ran once per interaction with a third of their lanes
// on.  So an interaction only PUSHES its exits that pass the cheap culls {world direction, weight, wavelength index} onto the
// wave's queue in LDS (ballot-compacted), and the wave pops 64 of them at a time: the expensive half of the emit runs ~2.5
// times per ray pass instead of 7, with full lanes.  The count lives in a register — every lane still in the interaction
// loop takes part in every push, so theirs agree — and is parked in LDS between passes (lanes that left a pass early hold a
// stale copy).
constexpr uint32_t kExitQ = 128u;   // < 64 left after a drain + <= 64 pushed by one interaction
struct ExitQueue {
  float x[kExitQ], y[kExitQ], z[kExitQ], w[kExitQ];
  uint8_t wl[kExitQ];   // wavelength-pool entry (HALO_WL_POOL_MAX = 255)
  uint32_t n;
};
struct ExitQueueMask {   // raypath colour (kModeColor): the exit's component mask rides along
  uint32_t lo[kExitQ], hi[kExitQ];
};
template <bool ON>
struct ExitQueueMasks {
  ExitQueueMask q[kBlock / 64];
};
template <>
struct ExitQueueMasks<false> {
  uint32_t unused;
};
template <bool ON>
struct ExitQueues {
  ExitQueue q[kBlock / 64];
};
template <>
struct ExitQueues<false> {
  uint32_t unused;
};
// The projection's constants followed by the four int -> float conversions the pixel formulas of the fisheye family begin with (half the
// image's width and height, the lens shifts), made once per dispatch on the host (DispatchParams::proj_pre lies directly behind ::proj) instead
// of once per exit: the same operations on the same integers, the same floats.  The exit queue's drain reads the whole of it from its LDS copy
// (HALO_PROJ_LDS).  project_exit takes a ProjDev and reads the four floats BEHIND it: it is only ever handed one of these two.
struct ProjLds {
  ProjDev p;
  float half_w, half_h, shift_x, shift_y;
};
static_assert(offsetof(ProjLds, half_w) == sizeof(ProjDev) && offsetof(DispatchParams, proj_pre) == offsetof(DispatchParams, proj) + sizeof(ProjDev),
              "the four floats lie directly behind the ProjDev, in LDS and in the dispatch record");
template <bool MONO, bool SMALLC>
struct AccCtx {
  int lens, vis;     // >= 0: instantiated for this lens / visible range (the projection's dispatch folds away)
  bool nogate;       // instantiated for prob <= 0: no candidate ever passes the gate, the gate stream and its code fold away
  bool last;         // kAccLogFinal kernels: the scene's last layer — no candidate continues, the append code is compiled out
  bool none;         // kAccNone kernels: every outgoing candidate continues (prob >= 1, not the last layer), nothing is projected
  ExitQueue* q;      // this wave's exit queue; nullptr = project and accumulate at the emit site
  ExitQueueMask* qm; // ... its colour-mask planes (kModeColor)
  const FastTables __attribute__((address_space(4)))* fast;   // kModeFilter / kModeColor: the dispatch's filter + colour predicates (dispatch-uniform, scalar loads)
  const uint32_t* fast_ee;  // ... the first kFastEeLds entry/exit matrices, staged in LDS
  PixCache<MONO, SMALLC>* cache;
  HitBuffer* hits;   // nullptr = accumulate directly
  uint32_t* log_n;   // hit-log kernels: the workgroup's log cursor (LDS); nullptr otherwise
  const ProjLds* proj;   // HALO_PROJ_LDS: the projection's constants staged in LDS for the exit queue's drain (nullptr = the dispatch record's)
};

// Accumulation planes (halo_device.h MonoSlot): plane `pl`, privatised copy of this workgroup, slot of `pix`.
HD float* mono_slot(const DispatchParams& P, uint32_t pl, uint32_t pix) {
  const uint32_t copy = blockIdx.x & P.mono_copy_mask;
  return P.mono + (static_cast<size_t>(pl * (P.mono_copy_mask + 1u) + copy) << (P.mono_s_log2 + 10u)) + MonoSlot(pix, P.mono_s_log2);
}

// Stage one hit record in the workgroup's buffer; false = buffer full (the caller adds the hit directly).
HD bool stage_hit(HitBuffer* hb, uint32_t key, float w) {
  // one LDS atomic per wave, not per lane (64 lanes on one address would serialise)
  const uint64_t mask = __ballot(1);
  const uint32_t lane = __lane_id();
  const uint32_t leader = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(mask))) - 1u;
  uint32_t first = 0u;
  if (lane == leader) first = atomicAdd(&hb->n, static_cast<uint32_t>(__popcll(mask)));
  first = __shfl(first, static_cast<int>(leader));
  const uint32_t pos = first + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
  if (pos >= static_cast<uint32_t>(kHitBuf)) return false;
  hb->h[pos] = make_uint2(key, __float_as_uint(w));
  return true;
}

// Where a record goes that found its log region or its tile list full: the planes' fp64 twin when the dispatch has one, else the fp32 plane.
HD void overflow_add(const DispatchParams& P, size_t off, float v) {
  if (P.ovf != nullptr) {
    atomicAdd(P.ovf + TwinOffset(off, P.mono_s_log2 + 10u, P.ovf_copies_log2), static_cast<double>(v));
    *P.ovf_flag = 1u;
  } else {
    atomic_add_f32(P.mono + off, v);
  }
}

// Hit log.  Global fp32 atomics execute memory-side on this part, 21 G/s whatever the footprint or the lanes per instruction
// (tools/atomic_rate_bench.hip) — a floor of 2.8 ms under configs[1]'s 58 M cache misses, next to a 2.4 ms trace — while a wave
// appending 8-byte records to a private run of HBM sustains > 200 G records/s.  So a logging kernel ADDS nothing: a hit that
// misses the pixel cache goes to the workgroup's own log region (cursor in LDS, one ds_add_rtn per wave and call), and
// halo_split_kernel / halo_bin_accumulate_range_kernel sum the logs per 16 Ki-slot tile in LDS afterwards.  A full region
// falls back to the direct atomic.  Call with any subset of a wave's lanes active.
template <bool XYZ>   // XYZ: record of an X/Y/Z kernel (slot in one plane | CMF code); else a scalar plane's
HD void log_hit(const DispatchParams& P, uint32_t* log_n, uint32_t slot, float w) {
  const uint64_t mask = __ballot(1);
  // position among the active lanes; the first of them reserves the wave's run
  const uint32_t before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
  uint32_t base = 0u;
  if (before == 0u) base = atomicAdd(log_n, static_cast<uint32_t>(__popcll(mask)));
  const uint32_t idx = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(base))) + before;
  if (idx < P.bin_cap) {
    reinterpret_cast<uint2*>(P.bin_list)[static_cast<size_t>(blockIdx.x) * P.bin_cap + idx] = make_uint2(slot, __float_as_uint(w));
  } else if (XYZ) {   // full region: X, Y, Z directly (copy 0)
    const uint32_t code = slot >> kLogWlShift;
    float cx, cy, cz;
    if (code < P.wl_pool_size) {
      const WlEntryDev e = P.wl_pool[code];
      cx = e.cmf_x, cy = e.cmf_y, cz = e.cmf_z;
    } else {
      cx = code == P.wl_pool_size ? 1.0f : 0.0f, cy = code == P.wl_pool_size + 1u ? 1.0f : 0.0f, cz = code == P.wl_pool_size + 2u ? 1.0f : 0.0f;
    }
    const size_t at = slot & ((1u << kLogWlShift) - 1u);
    if (cx != 0.0f) overflow_add(P, at, cx * w);
    if (cy != 0.0f) overflow_add(P, at + P.log_plane_stride, cy * w);
    if (cz != 0.0f) overflow_add(P, at + 2u * static_cast<size_t>(P.log_plane_stride), cz * w);
  } else {
    overflow_add(P, slot, w);   // copy 0
  }
}

// What a hit of scalar plane `pl` (0, or the ray's wavelength-pool entry) on pixel `pix` is called in the log: its slot in the array of planes.
HD uint32_t log_slot(const DispatchParams& P, uint32_t pl, uint32_t pix) { return (pl << (P.mono_s_log2 + 10u)) + MonoSlot(pix, P.mono_s_log2); }
// X/Y/Z kernels under the hit log: the slot in ONE plane with the CMF code above it (the per-tile pass makes X, Y, Z)
HD uint32_t log_slot_xyz(const DispatchParams& P, uint32_t code, uint32_t pix) { return MonoSlot(pix, P.mono_s_log2) | (code << kLogWlShift); }

// MONO: one scalar per hit into plane 0 (discrete wavelength) or plane wl_idx (illuminant session with one plane per
// pool entry); the CMF is applied by halo_fold_kernel.  !MONO: X, Y, Z into planes 0..2.
template <bool MONO, bool SMALLC>
HD void accumulate(const DispatchParams& P, const AccCtx<MONO, SMALLC>& ctx, uint32_t pix, uint32_t wl_idx, float w, float cx, float cy, float cz) {
  PixCache<MONO, SMALLC>& C = *ctx.cache;
  if (P.aggregate == 2u) return;  // diagnostic: trace + project only
  const uint32_t pl = (MONO && P.mono_by_wl) ? wl_idx : 0u;
  if (P.aggregate == 1u || P.aggregate == 3u) {
    const uint32_t key = ((pl << 23) | pix) + 1u;
    // X/Y/Z caches (round 5): four-way sets, one 16-byte read of the set's tags.  A hot pixel only misses when ALL its ways were claimed by other
    // pixels before its first hit.  With two ways ~0.5 % of the workgroups lost a sun-disc pixel that way, and every record of theirs for it then
    // met the same LDS address in the per-tile pass of the sun's tile — the tail of that pass: configs[4]'s passes 1.32 -> 0.98 ms per launch with
    // four ways.  The probe itself is dearer in instructions than the two compare-and-swaps (a read, eight compares and selects: the trace kernel
    // +2.7 %), which is why the scalar caches keep two ways: their passes measured the same either way, and configs[1]'s kernel lost 2 %.
    // The key is there: the read finds it.  The set is full of other keys: the read says so.  A first sighting with a free way pays a
    // compare-and-swap on top; the loser of a race for the way reads the set again, once.
    if constexpr (!MONO && HALO_CACHE_WAYS == 4) {
    const uint32_t set = ((key * 2654435761u) >> (32 - (CacheGeom<MONO, SMALLC>::kLog2 - 2))) << 2;
    uint32_t slot = 0xFFFFFFFFu;
#pragma unroll
    for (int attempt = 0; attempt < 2; ++attempt) {
      const uint4 t = *reinterpret_cast<const uint4*>(&C.tag[set]);
      const uint32_t found = t.x == key ? 0u : t.y == key ? 1u : t.z == key ? 2u : t.w == key ? 3u : 4u;
      if (found < 4u) {
        slot = set + found;
        break;
      }
      const uint32_t empty = t.x == 0u ? 0u : t.y == 0u ? 1u : t.z == 0u ? 2u : t.w == 0u ? 3u : 4u;
      if (empty == 4u) break;   // the set belongs to four other pixels
      const uint32_t old = atomicCAS(&C.tag[set + empty], 0u, key);
      if (old == 0u || old == key) {
        slot = set + empty;
        break;
      }
    }
    if (slot != 0xFFFFFFFFu) {
      if (MONO) {
        unsafeAtomicAdd(&C.val[slot], w);
      } else {
        unsafeAtomicAdd(&C.val[slot * 3 + 0], cx * w);
        unsafeAtomicAdd(&C.val[slot * 3 + 1], cy * w);
        unsafeAtomicAdd(&C.val[slot * 3 + 2], cz * w);
      }
      return;
    }
    } else {
    // two-way: a key may live in slot s or s^1.  A hot pixel only misses the cache when BOTH were claimed by other pixels
    // before its first hit (~0.2 % of workgroups instead of ~5 % one-way) — and every miss of a hot pixel is an atomic on
    // the same line as all its other misses, chip-wide.
    uint32_t slot = (key * 2654435761u) >> (32 - CacheGeom<MONO, SMALLC>::kLog2);
    uint32_t old = atomicCAS(&C.tag[slot], 0u, key);
    if (old != 0u && old != key) {
      slot ^= 1u;
      old = atomicCAS(&C.tag[slot], 0u, key);
    }
    if (old == 0u || old == key) {
      if (MONO) {
        unsafeAtomicAdd(&C.val[slot], w);
      } else {
        unsafeAtomicAdd(&C.val[slot * 3 + 0], cx * w);
        unsafeAtomicAdd(&C.val[slot * 3 + 1], cy * w);
        unsafeAtomicAdd(&C.val[slot * 3 + 2], cz * w);
      }
      return;
    }
    }
    if (P.aggregate == 3u) return;  // diagnostic: cache only, misses dropped
  }
  if (ctx.log_n != nullptr) {   // hit-log kernels: a scalar record; for X/Y/Z planes it names the ray's pool entry instead of three products
    log_hit<!MONO>(P, ctx.log_n, MONO ? log_slot(P, pl, pix) : log_slot_xyz(P, wl_idx, pix), w);
    return;
  }
  if (MONO) {
    // binned: {slot, w}; the planes of a per-entry-plane session lie back to back, so `slot` addresses them as one array
    if (ctx.hits != nullptr && stage_hit(ctx.hits, (pl << (P.mono_s_log2 + 10u)) + MonoSlot(pix, P.mono_s_log2), w)) return;
    atomic_add_f32(mono_slot(P, pl, pix), w);
  } else {
    atomic_add_f32(mono_slot(P, 0u, pix), cx * w);
    atomic_add_f32(mono_slot(P, 1u, pix), cy * w);
    atomic_add_f32(mono_slot(P, 2u, pix), cz * w);
  }
}

struct RaySums {
  float landed;
  float exit_w;
  uint32_t exit_n;
  uint32_t pix_n;
  uint32_t qn;   // exit queue fill (see ExitQueue)
};

// ------------------------------------------------------------------------------------------------
// emit-gate filters (shared/filter_shared.h:53-315).  Paths are crystal face NUMBERS already, so the reference's
// ApplyGetFn remap (poly index → face number) is the identity here.
// ------------------------------------------------------------------------------------------------
//
// A path of at most 16 faces is ALSO kept as a 128-bit shift register (one byte per face, newest in the low byte), and the
// symmetry reduction then runs on packed values in registers: a sequence is held left-aligned (element 0 in the top byte),
// every transform streams the bytes out of the top of one register pair into the bottom of another, and "lexicographically
// smaller" is an unsigned compare.  The byte-array versions below remain for longer paths (max_hits up to 64).
struct Pk128 {
  uint64_t hi, lo;
};
struct PathView {
  const uint8_t* bytes;   // face numbers of interactions 16 .. kFilterPathCap-1 (the array is only written past the register)
  uint32_t len;           // interactions recorded (may exceed the capacity of both)
  Pk128 reg;              // shift register of the first min(len, 16) interactions, newest in the low byte
};
HD Pk128 pk_shl8(Pk128 v) { return {(v.hi << 8) | (v.lo >> 56), v.lo << 8}; }
HD Pk128 pk_shl_bytes(Pk128 v, uint32_t n) {  // n in [0, 16]
  const uint32_t b = n * 8u;
  if (b == 0u) return v;
  if (b >= 128u) return {0ull, 0ull};
  if (b >= 64u) return {v.lo << (b - 64u), 0ull};
  return {(v.hi << b) | (v.lo >> (64u - b)), v.lo << b};
}
HD bool pk_less(Pk128 a, Pk128 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
HD bool pk_eq(Pk128 a, Pk128 b) { return a.hi == b.hi && a.lo == b.lo; }
HD uint32_t pk_byte(Pk128 v, uint32_t pos) {  // byte `pos` counted from the low end
  return static_cast<uint32_t>((pos < 8u ? v.lo >> (8u * pos) : v.hi >> (8u * (pos - 8u))) & 0xFFull);
}

HD uint8_t path_at(const PathView& pv, uint32_t i) {  // interaction i of a path of any recorded length
  const uint32_t held = pv.len < 16u ? pv.len : 16u;
  return i < 16u ? static_cast<uint8_t>(pk_byte(pv.reg, held - 1u - i)) : pv.bytes[i];
}

HD Pk128 pk_p_shift(Pk128 in, uint32_t len) {  // p_canonical_shift on a left-aligned sequence
  Pk128 out = {0ull, 0ull};
  int first_pri = -1;
  for (uint32_t i = 0; i < len; ++i) {
    uint32_t x = static_cast<uint32_t>(in.hi >> 56);
    in = pk_shl8(in);
    if (x >= 3u) {
      const uint32_t pyr = x / 10u;
      int pri = static_cast<int>(x % 10u);
      if (first_pri < 0) first_pri = pri;
      pri = (pri + 6 - first_pri) % 6 + 3;
      x = pyr * 10u + static_cast<uint32_t>(pri);
    }
    out = pk_shl8(out);
    out.lo |= x;
  }
  return pk_shl_bytes(out, 16u - len);
}
HD Pk128 pk_d_image(Pk128 in, uint32_t len, int32_t sigma_a) {
  Pk128 out = {0ull, 0ull};
  for (uint32_t i = 0; i < len; ++i) {
    uint32_t x = static_cast<uint32_t>(in.hi >> 56);
    in = pk_shl8(in);
    if (x >= 3u) {
      const uint32_t pyr = x / 10u;
      const int pri0 = static_cast<int>(x % 10u) - 3;
      const int np = ((sigma_a - pri0) % 6 + 6) % 6;
      x = pyr * 10u + static_cast<uint32_t>(np + 3);
    }
    out = pk_shl8(out);
    out.lo |= x;
  }
  return pk_shl_bytes(out, 16u - len);
}
HD Pk128 pk_b_image(Pk128 in, uint32_t len, bool& changed) {
  Pk128 out = {0ull, 0ull};
  changed = false;
  for (uint32_t i = 0; i < len; ++i) {
    uint32_t x = static_cast<uint32_t>(in.hi >> 56);
    in = pk_shl8(in);
    if (x <= 2u) {
      x = 3u - x;
      changed = true;
    } else if (x >= 13u && x <= 18u) {
      x += 10u;
      changed = true;
    } else if (x >= 23u && x <= 28u) {
      x -= 10u;
      changed = true;
    }
    out = pk_shl8(out);
    out.lo |= x;
  }
  return pk_shl_bytes(out, 16u - len);
}
// reduce_buffer (below) on a left-aligned packed sequence of len <= 16
HD Pk128 pk_reduce(Pk128 data, uint32_t len, uint8_t symmetry, int32_t sigma_a, bool d_applicable) {
  if (symmetry == 0u) return data;
  if (symmetry & HALO_SYM_P) data = pk_p_shift(data, len);
  if ((symmetry & HALO_SYM_D) && d_applicable) {
    Pk128 img = pk_d_image(data, len, sigma_a);
    if (symmetry & HALO_SYM_P) img = pk_p_shift(img, len);
    if (pk_less(img, data)) data = img;
  }
  if (symmetry & HALO_SYM_B) {
    bool changed;
    const Pk128 img = pk_b_image(data, len, changed);
    if (changed && pk_less(img, data)) data = img;
  }
  return data;
}

HD void p_canonical_shift(uint8_t* data, uint32_t size) {  // filter_shared.h:53-68
  int first_pri = -1;
  for (uint32_t i = 0; i < size; ++i) {
    const uint8_t x = data[i];
    if (x < 3u) continue;
    const uint32_t pyr = x / 10u;
    int pri = static_cast<int>(x % 10u);
    if (first_pri < 0) first_pri = pri;
    pri = (pri + 6 - first_pri) % 6 + 3;
    data[i] = static_cast<uint8_t>(pyr * 10u + static_cast<uint32_t>(pri));
  }
}

HD bool lex_less(const uint8_t* a, const uint8_t* b, uint32_t size) {
  for (uint32_t i = 0; i < size; ++i)
    if (a[i] != b[i]) return a[i] < b[i];
  return false;
}

HD void reduce_buffer(uint8_t* data, uint32_t size, uint8_t symmetry, int32_t sigma_a, bool d_applicable) {  // :79-132
  if (symmetry == 0u) return;
  if (symmetry & HALO_SYM_P) p_canonical_shift(data, size);
  uint8_t scratch[kFilterPathCap];
  if ((symmetry & HALO_SYM_D) && d_applicable) {
    for (uint32_t i = 0; i < size; ++i) {
      const uint8_t x = data[i];
      if (x < 3u) {
        scratch[i] = x;
        continue;
      }
      const uint32_t pyr = x / 10u;
      const int pri0 = static_cast<int>(x % 10u) - 3;
      const int np = ((sigma_a - pri0) % 6 + 6) % 6;
      scratch[i] = static_cast<uint8_t>(pyr * 10u + static_cast<uint32_t>(np + 3));
    }
    if (symmetry & HALO_SYM_P) p_canonical_shift(scratch, size);
    if (lex_less(scratch, data, size))
      for (uint32_t i = 0; i < size; ++i) data[i] = scratch[i];
  }
  if (symmetry & HALO_SYM_B) {
    bool changed = false;
    for (uint32_t i = 0; i < size; ++i) {
      const uint8_t x = data[i];
      if (x <= 2u) {
        scratch[i] = static_cast<uint8_t>(3u - x);
        changed = true;
      } else if (x >= 13u && x <= 18u) {
        scratch[i] = static_cast<uint8_t>(x + 10u);
        changed = true;
      } else if (x >= 23u && x <= 28u) {
        scratch[i] = static_cast<uint8_t>(x - 10u);
        changed = true;
      } else {
        scratch[i] = x;
      }
    }
    if (changed && lex_less(scratch, data, size))
      for (uint32_t i = 0; i < size; ++i) data[i] = scratch[i];
  }
}

// NOT inlined: the emit gate sits at three places of the trace loop and each one evaluates up to 16 filter terms and 16
// colour terms; inlined, one filter kernel grew to 555 KB of code (the plain kernel is 96 KB) and ran out of the
// instruction cache.  One shared copy is called instead; the tables are reached through generic pointers.
__device__ __attribute__((noinline)) bool filter_match_path(uint8_t symmetry, int32_t sigma_a, bool d_applicable, const FilterTermDev& t, const PathView& pv, float wx,
                          float wy, float wz, uint32_t crystal_id) {  // DeviceFilterMatchSimple :238-257
  const uint32_t len = pv.len;
  if (t.type == HALO_FILTER_NONE) return true;
  if (t.type == HALO_FILTER_RAYPATH) {  // :156-178
    if (len != t.canonical_len) return false;
    if (len <= 16u) {  // register path
      const Pk128 red = pk_reduce(pk_shl_bytes(pv.reg, 16u - len), len, symmetry, sigma_a, d_applicable);
      return pk_eq(red, Pk128{t.canon_hi, t.canon_lo});
    }
    uint8_t buf[kFilterPathCap];
    for (uint32_t i = 0; i < len; ++i) buf[i] = path_at(pv, i);
    reduce_buffer(buf, len, symmetry, sigma_a, d_applicable);
    for (uint32_t i = 0; i < len; ++i)
      if (buf[i] != t.canonical[i]) return false;
    return true;
  }
  if (t.type == HALO_FILTER_ENTRY_EXIT) {  // :180-224
    if (len == 0u || len < t.min_len) return false;
    if (t.max_len != 0u && len > t.max_len) return false;
    if (!t.has_entry && !t.has_exit) return true;
    if (len <= 16u) {  // register path: the (entry, exit) pair as a packed sequence of one or two faces
      const uint32_t first = pk_byte(pv.reg, len - 1u), last = static_cast<uint32_t>(pv.reg.lo & 0xFFull);
      uint32_t n2 = 0u;
      uint64_t top = 0ull;
      if (t.has_entry) top |= static_cast<uint64_t>(first) << (56u - 8u * n2++);
      if (t.has_exit) top |= static_cast<uint64_t>(last) << (56u - 8u * n2++);
      const Pk128 red = pk_reduce(Pk128{top, 0ull}, n2, symmetry, sigma_a, d_applicable);
      if (symmetry != 0u && n2 != t.canonical_len) return false;
      return pk_eq(red, Pk128{t.canon_hi, t.canon_lo});
    }
    uint8_t ee[2];
    uint32_t n = 0u;
    if (t.has_entry) ee[n++] = path_at(pv, 0u);
    if (t.has_exit) ee[n++] = path_at(pv, (len < kFilterPathCap ? len : kFilterPathCap) - 1u);
    reduce_buffer(ee, n, symmetry, sigma_a, d_applicable);
    if (symmetry != 0u && n != t.canonical_len) return false;
    for (uint32_t i = 0; i < n; ++i)
      if (ee[i] != t.canonical[i]) return false;
    return true;
  }
  if (t.type == HALO_FILTER_DIRECTION) return t.dir[0] * wx + t.dir[1] * wy + t.dir[2] * wz > t.radii_c;  // :226-229
  if (t.type == HALO_FILTER_CRYSTAL) return crystal_id == t.crystal_id;                                    // :231-233
  return false;
}

// the predicates that never look at the path stay inline; only the two path predicates pay for the call
HD bool filter_match_term(uint8_t symmetry, int32_t sigma_a, bool d_applicable, const FilterTermDev& t, const PathView& pv, float wx,
                          float wy, float wz, uint32_t crystal_id) {
  if (t.type == HALO_FILTER_NONE) return true;
  if (t.type == HALO_FILTER_DIRECTION) return t.dir[0] * wx + t.dir[1] * wy + t.dir[2] * wz > t.radii_c;  // :226-229
  if (t.type == HALO_FILTER_CRYSTAL) return crystal_id == t.crystal_id;                                    // :231-233
  return filter_match_path(symmetry, sigma_a, d_applicable, t, pv, wx, wy, wz, crystal_id);
}

HD bool filter_check(const FilterDev& f, const PathView& pv, float wx, float wy, float wz, uint32_t crystal_id) {
  bool m;
  if (!f.is_complex) {
    m = filter_match_term(f.symmetry, f.sigma_a, f.d_applicable != 0u, f.terms[0], pv, wx, wy, wz, crystal_id);
  } else {  // OR over AND-clauses; an empty complex filter matches nothing (:263-291)
    m = false;
    uint32_t idx = 0u;
    for (uint32_t o = 0u; o < f.or_count && !m; ++o) {
      const uint32_t n = f.and_counts[o];
      bool all = true;
      for (uint32_t a = 0u; a < n && all; ++a) all = filter_match_term(f.symmetry, f.sigma_a, f.d_applicable != 0u, f.terms[idx + a], pv, wx, wy, wz, crystal_id);
      idx += n;
      m = all;
    }
  }
  return (f.action == 0u) ? m : !m;  // Check = Match XOR filter_out (:308-315)
}

// Raypath colour (ApplyLayerColorBits cu:498-527): OR into the carried mask the bit of every predicate of this crystal
// entry that matches the exit.  Non-destructive: runs beside the physical filter, never drops a ray.
HD uint64_t color_bits(const ColorDev& c, uint64_t carried, const PathView& pv, float wx, float wy, float wz, uint32_t crystal_id) {
  uint64_t m = carried;
  for (uint32_t k = 0u; k < c.term_cnt; ++k) {
    const ColorTermDev& ct = c.terms[k];
    if (ct.bit < 64u && filter_match_term(ct.symmetry, ct.sigma_a, ct.d_applicable != 0u, ct.t, pv, wx, wy, wz, crystal_id))
      m |= 1ull << ct.bit;
  }
  return m;
}

// FanColorClassLanes cu:535-556: the exit's Y goes to every class whose rule its mask satisfies (primary AND overlap hits)
HD void fan_lanes(const DispatchParams& P, const ColorDev& c, uint64_t mask, uint32_t pix, float y_val) {
  for (uint32_t k = 0u; k < c.class_cnt; ++k) {
    const uint64_t bits = c.class_bits[k];
    if (bits == 0ull) continue;
    const uint64_t matched = mask & bits;
    const bool ok = c.class_all[k] ? (matched == bits) : (matched != 0ull);
    if (ok) atomicAdd(P.lanes + static_cast<size_t>(k) * P.lane_stride + pix, static_cast<double>(y_val));   // global_atomic_add_f64
  }
}

// ------------------------------------------------------------------------------------------------
// The same predicates in their fast form (halo_device.h FastTables), for max_hits <= 16 — the kModeFilter / kModeColor kernels.
// Nothing is reduced here: the host listed, per raypath term, the sequences whose reduction is the term's canonical form, and
// turned every entry/exit term into a bit matrix over (entry face, exit face).  Terms, counts and the path LENGTH are
// wave-uniform (every lane of an interaction has recorded the same number of faces; the rare stray lane is evaluated by a
// second, uniformly branched call), so the walk over clauses and terms is scalar control flow over scalar loads, and the only
// per-lane work is one 64-bit compare per member, one table word per entry/exit term, three FMAs per direction term.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kFastEeLds = 8u;   // entry/exit matrices staged in LDS (1 KB); a dispatch with more reads the rest from HBM
// The tables are read through the CONSTANT address space: a plain global pointer makes every one of these dispatch-uniform reads a
// vector load + v_readfirstlane behind a full memory wait — the kernel stores to global memory, so the compiler cannot prove the
// tables unclobbered (measured: the all-pass crystal-id filter at 1.37x the plain kernel) — while constant-space loads are scalar
// (s_load through the scalar cache), invariant and hoistable.  The tables are written by the dispatch's H2D copy, before the launch.
typedef const FastTables __attribute__((address_space(4))) FastTablesC;
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint64_t u64x8 __attribute__((ext_vector_type(8)));
typedef uint64_t u64x4 __attribute__((ext_vector_type(4)));
struct FastTermS {   // one term in scalar registers: ONE s_load_dwordx8 (field packing: halo_device.h FastTerm)
  u32x8 r;
  HD uint32_t type() const { return r[0] & 0xFFu; }
  HD uint32_t last() const { return (r[0] >> 8) & 0xFFu; }
  HD uint32_t bit() const { return (r[0] >> 16) & 0xFFu; }
  HD uint32_t len() const { return r[0] >> 24; }
  HD uint32_t min_len() const { return r[1] & 0xFFu; }
  HD uint32_t max_len() const { return (r[1] >> 8) & 0xFFu; }
  HD uint32_t orbit_n() const { return r[1] >> 16; }
  HD uint32_t orbit_off() const { return r[2] & 0xFFFFu; }
  HD uint32_t ee_off() const { return r[2] >> 16; }
  HD uint32_t crystal_id() const { return r[3]; }
  HD float dir(int k) const { return __uint_as_float(r[4 + k]); }
  HD float radii_c() const { return __uint_as_float(r[7]); }
};
static_assert(offsetof(FastTerm, crystal_id) == 12 && offsetof(FastTerm, dir) == 16 && offsetof(FastTerm, radii_c) == 28, "FastTermS mirrors FastTerm");
HD FastTermS load_term(const FastTerm __attribute__((address_space(4)))* t) {
  FastTermS o;
  o.r = *reinterpret_cast<const u32x8 __attribute__((address_space(4)))*>(t);
  return o;
}

HD bool fast_term(FastTablesC& F, const uint32_t* ee_lds, const FastTermS& t, uint32_t L, Pk128 reg, float wx, float wy, float wz, uint32_t crystal_id) {
  const uint32_t type = t.type();
  if (type == HALO_FILTER_NONE) return true;
  if (type == HALO_FILTER_RAYPATH) {  // DeviceFilterMatchSimple :156-178
    if (L != t.len()) return false;
    const uint32_t n = t.orbit_n(), off = t.orbit_off();
    bool m = false;
    if (L <= 8u) {   // eight members per scalar load (lists are padded to a multiple of eight with ~0)
      for (uint32_t k = 0u; k < n; k += 8u) {
        const u64x8 v = *reinterpret_cast<const u64x8 __attribute__((address_space(4)))*>(&F.orbit_lo[off + k]);
#pragma unroll
        for (int j = 0; j < 8; j++) m = m || (reg.lo == v[j]);
      }
    } else {
      for (uint32_t k = 0u; k < n; k += 4u) {
        const u64x4 lo = *reinterpret_cast<const u64x4 __attribute__((address_space(4)))*>(&F.orbit_lo[off + k]);
        const u64x4 hi = *reinterpret_cast<const u64x4 __attribute__((address_space(4)))*>(&F.orbit_hi[off + k]);
#pragma unroll
        for (int j = 0; j < 4; j++) m = m || (reg.lo == lo[j] && reg.hi == hi[j]);
      }
    }
    return m;
  }
  if (type == HALO_FILTER_ENTRY_EXIT) {  // :180-224
    if (L == 0u || L < t.min_len()) return false;
    if (t.max_len() != 0u && L > t.max_len()) return false;
    const uint32_t eo = t.ee_off();
    if (eo == 0xFFFFu) return true;   // neither face constrained
    const uint32_t sh = 8u * (L - 1u);   // scalar
    const uint32_t first = static_cast<uint32_t>((sh < 64u ? reg.lo >> sh : reg.hi >> (sh - 64u)) & 0xFFull);
    const uint32_t last = static_cast<uint32_t>(reg.lo & 0xFFull);
    const uint32_t row = eo < kFastEeLds ? ee_lds[eo * 32u + (first & 31u)] : F.ee[eo][first & 31u];
    return first < 32u && last < 32u && ((row >> last) & 1u) != 0u;
  }
  if (type == HALO_FILTER_DIRECTION) return t.dir(0) * wx + t.dir(1) * wy + t.dir(2) * wz > t.radii_c();  // :226-229
  if (type == HALO_FILTER_CRYSTAL) return crystal_id == t.crystal_id();                                      // :231-233
  return false;
}

// The filter's header: the first eight dwords of FastTables, ONE scalar load at the emit site.  It decides most exits by itself (len_mode:
// every exit of this length fails / passes; or the filter is one direction term whose constants it carries); only when the terms must be
// asked does the walk load them (one more load for the first term).  Measured alternatives, 10 M rays, plain kernel 0.70 ms — all-pass
// crystal filter / one-term direction filter / 3-clause complex filter: header + first term as four loads at the emit 0.785 / 0.839 / 0.981 ms;
// read once per ray pass and held across the interaction loop 0.811 / 0.863 / 1.027 (13 more live scalar registers: spills); requested at
// the top of every interaction, ahead of the Fresnel block 0.806 / 0.856 / 1.016; field by field where the walk needs them (a
// scalar-cache round trip each) 0.823 / 0.828 / 1.263; as vector loads through a plain global pointer 0.970 / 0.980 / 1.594.
struct FastHdr {
  uint64_t len_mode;
  uint32_t term_cnt, action;
  float dir0[3], radii0;
};
HD FastHdr load_fast_hdr(FastTablesC& F) {
  const u32x8 r = *reinterpret_cast<const u32x8 __attribute__((address_space(4)))*>(&F);
  FastHdr h;
  h.len_mode = static_cast<uint64_t>(r[0]) | (static_cast<uint64_t>(r[1]) << 32);
  h.term_cnt = r[2];
  h.action = r[3];
  h.dir0[0] = __uint_as_float(r[4]);
  h.dir0[1] = __uint_as_float(r[5]);
  h.dir0[2] = __uint_as_float(r[6]);
  h.radii0 = __uint_as_float(r[7]);
  return h;
}
static_assert(offsetof(FastTables, len_mode) == 0 && offsetof(FastTables, term_cnt) == 8 && offsetof(FastTables, action) == 12 && offsetof(FastTables, dir0) == 16 &&
              offsetof(FastTables, radii0) == 28, "load_fast_hdr reads the first eight dwords");

// `lm`: what the host already knows about exits of this length (FastTables::len_mode): 0 all fail, 1 all pass, 2 ask the terms, 3 one direction term (header)
HD bool fast_filter(FastTablesC& F, const uint32_t* ee_lds, const FastHdr& H, uint32_t lm, uint32_t L, Pk128 reg, float wx, float wy, float wz, uint32_t crystal_id) {
  if (lm == 3u) {
    const bool m = H.dir0[0] * wx + H.dir0[1] * wy + H.dir0[2] * wz > H.radii0;   // :226-229
    return (H.action == 0u) ? m : !m;
  }
  if (lm != 2u) return lm == 1u;
  // OR over AND-clauses as one flat walk (FastTerm::last closes a clause); every term visited: uniform trip count (:263-291)
  const uint32_t n = H.term_cnt;
  bool m = false, all = true;
  for (uint32_t k = 0u; k < n; ++k) {
    const FastTermS t = load_term(&F.fterm[k]);
    all = fast_term(F, ee_lds, t, L, reg, wx, wy, wz, crystal_id) && all;
    if (t.last()) {
      m = m || all;
      all = true;
    }
  }
  return (H.action == 0u) ? m : !m;  // Check = Match XOR filter_out (:308-315); an empty complex filter matches nothing
}

HD uint64_t fast_color_bits(FastTablesC& F, const uint32_t* ee_lds, uint64_t carried, uint32_t L, Pk128 reg, float wx, float wy, float wz, uint32_t crystal_id) {  // ApplyLayerColorBits cu:498-527
  uint64_t m = carried;
  const uint32_t n = F.color_terms;
  for (uint32_t k = 0u; k < n; ++k) {
    const FastTermS t = load_term(&F.cterm[k]);
    const uint32_t bit = t.bit();
    if (bit < 64u && fast_term(F, ee_lds, t, L, reg, wx, wy, wz, crystal_id)) m |= 1ull << bit;
  }
  return m;
}

HD void fan_lanes_fast(const DispatchParams& P, FastTablesC& F, uint64_t mask, uint32_t pix, float y_val) {  // FanColorClassLanes cu:535-556
  const uint32_t n = F.class_cnt;
  for (uint32_t k = 0u; k < n; ++k) {
    const uint64_t bits = F.class_bits[k];
    if (bits == 0ull) continue;
    const uint64_t matched = mask & bits;
    const bool ok = F.class_all[k] ? (matched == bits) : (matched != 0ull);
    if (ok) atomicAdd(P.lanes + static_cast<size_t>(k) * P.lane_stride + pix, static_cast<double>(y_val));   // global_atomic_add_f64
  }
}

// ------------------------------------------------------------------------------------------------
// the fused kernel.  MODE: 0 = production; 1 = + emit-gate filter, 4 = + filter and raypath colour — both in their fast form
// (path in a 128-bit register, FastTables), built like the production kernels: exit queue, hit log, regular-prism search;
// 2 = + exit capture (tests), 3 = the generic filter / colour kernels (paths up to 64 faces, symmetry reduction on the device)
// ------------------------------------------------------------------------------------------------
// Per-face view of the fan-triangle table of a dispatch's ONE shape (deterministic crystals), built by the workgroup when it
// stages the shape: the entry pick then walks the faces (8 for a prism) and only the triangles of the face it lands in,
// instead of all fan triangles twice.  ok = 0 (shape pools, or a table whose triangles are not grouped face by face in face
// order) selects the flat walk.
struct FaceIndex {
  float area[kMaxFaces];       // sum of the face's fan-triangle areas
  uint8_t tri0[kMaxFaces];     // first fan triangle of the face
  uint8_t tric[kMaxFaces];     // number of fan triangles
  uint32_t ok;
};
template <bool MONO, bool SMALLC>
struct LdsTables {
  float lut[3 * kLutNodes];
  PixCache<MONO, SMALLC> cache;
  uint32_t seg[kContShards + 4];
  FaceIndex fidx;
  __attribute__((aligned(16))) EntryFastDev efast;   // staged only when P.entry_fast != nullptr (full prism, one shape per dispatch)
};
constexpr int kModePlain = 0, kModeFilter = 1, kModeCapture = 2, kModeGeneric = 3, kModeColor = 4;
template <int MODE>
struct ModeTraits {
  static constexpr bool kFast = MODE == kModePlain || MODE == kModeFilter || MODE == kModeColor;   // production-shaped kernels
  static constexpr bool kFastPath = MODE == kModeFilter || MODE == kModeColor;                      // ... that keep a path register
  static constexpr bool kTables = MODE == kModeCapture || MODE == kModeGeneric;                     // FilterDev / ColorDev in LDS, device-side reduction
};
template <bool ON>
struct FilterSlot {
  FilterDev f;
};
template <>
struct FilterSlot<false> {
  uint32_t unused;
};
template <bool ON>
struct ColorSlot {
  ColorDev c;
};
template <>
struct ColorSlot<false> {
  uint32_t unused;
};

static_assert(offsetof(ShapePrism, tri_v) % 16 == 0 && offsetof(ShapePrism, tri_na) % 16 == 0 && offsetof(ShapePrism, slab) % 16 == 0 &&
              offsetof(ShapePrism, tri_face) % 4 == 0 && offsetof(ShapePrism, face_number) % 4 == 0 && offsetof(ShapePrism, single) % 4 == 0,
              "rows are copied as float4 / dwords");
static_assert(offsetof(ShapeDev, tri_v) % 16 == 0 && offsetof(ShapeDev, tri_na) % 16 == 0 && offsetof(ShapeDev, slab) % 16 == 0 &&
              offsetof(ShapeDev, tri_face) % 4 == 0 && offsetof(ShapeDev, face_number) % 4 == 0 && offsetof(ShapeDev, single) % 4 == 0,
              "rows are copied as float4 / dwords");

constexpr int kGeomOne = 0, kGeomPool = 1, kGeomPoolPrism = 2;   // GEOM: one shape per dispatch | pool of ShapeDev | pool of ShapePrism (HBM records and LDS slots)
constexpr int kGeomOneHex = 3;   // one shape per dispatch AND it is a regular hexagonal prism (EntryFastDev::hex_regular): literal normals
template <int GEOM>
struct PoolSlotType {   // type = the LDS slot of a half-wave, rec = the record in the HBM pool
  typedef ShapeSlot48 type;
  typedef ShapeDev rec;
};
template <>
struct PoolSlotType<kGeomPoolPrism> {
  typedef ShapePrism type;
  typedef ShapePrism rec;
};
static_assert(offsetof(ShapeSlot48, tri_na) % 16 == 0 && offsetof(ShapeSlot48, slab) % 16 == 0 &&
              offsetof(ShapeSlot48, tri_face) % 4 == 0 && offsetof(ShapeSlot48, face_number) % 4 == 0 && offsetof(ShapeSlot48, single) % 4 == 0,
              "rows are copied as float4 / dwords");
static_assert(offsetof(ShapeHead, face) == offsetof(ShapeDev, face) && offsetof(ShapeHead, slab) == offsetof(ShapeDev, slab) && offsetof(ShapeHead, tri_cnt) == offsetof(ShapeDev, tri_cnt),
              "ShapeHead is a prefix of ShapeDev");
template <bool ON, typename SlotT, int N = kBlock / 32>
struct PoolSlots {
  SlotT s[N];
};
template <typename SlotT, int N>
struct PoolSlots<false, SlotT, N> {
  uint32_t unused;
};

// 32 lanes copy the rows one pool shape uses into their half-wave's LDS slot (coalesced 16-byte loads)
template <typename SlotT, typename RecT>
HD void stage_shape(SlotT* slot, const RecT* g, uint32_t l32) {
  constexpr uint32_t kF = sizeof(slot->face) / 16u, kS = sizeof(slot->slab) / 32u, kT = sizeof(slot->tri_na) / 16u, kN = sizeof(slot->single);
  const uint32_t fc = min(static_cast<uint32_t>(g->face_cnt), kF), tc = min(static_cast<uint32_t>(g->tri_cnt), kT);
  const uint32_t sc = min(static_cast<uint32_t>(g->slab_cnt), kS), n1 = min(static_cast<uint32_t>(g->single_cnt), kN);
  if (l32 == 0u) {
    slot->face_cnt = static_cast<int32_t>(fc);
    slot->tri_cnt = static_cast<int32_t>(tc);
    slot->slab_cnt = static_cast<int32_t>(sc);
    slot->single_cnt = static_cast<int32_t>(n1);
  }
  const float4* gf = reinterpret_cast<const float4*>(g->face);
  float4* sf = reinterpret_cast<float4*>(slot->face);
  for (uint32_t i = l32; i < fc; i += 32u) sf[i] = gf[i];
  const float4* gs = reinterpret_cast<const float4*>(g->slab);
  float4* ss = reinterpret_cast<float4*>(slot->slab);
  for (uint32_t i = l32; i < 2u * sc; i += 32u) ss[i] = gs[i];
  if constexpr (sizeof(slot->tri_v) == sizeof(void*)) {   // the slot keeps a pointer to the record's corner rows
    if (l32 == 0u) *reinterpret_cast<const float (**)[9]>(&slot->tri_v) = g->tri_v;
  } else {
    const float4* gv = reinterpret_cast<const float4*>(g->tri_v);
    float4* sv = reinterpret_cast<float4*>(const_cast<float(*)[9]>(slot->tri_v));
    for (uint32_t i = l32; i < (tc * 9u + 3u) / 4u; i += 32u) sv[i] = gv[i];
  }
  const float4* gn = reinterpret_cast<const float4*>(g->tri_na);
  float4* sn = reinterpret_cast<float4*>(slot->tri_na);
  for (uint32_t i = l32; i < tc; i += 32u) sn[i] = gn[i];
  const uint32_t* gb = reinterpret_cast<const uint32_t*>(g->tri_face);
  uint32_t* sb = reinterpret_cast<uint32_t*>(slot->tri_face);
  for (uint32_t i = l32; i < (tc + 3u) / 4u; i += 32u) sb[i] = gb[i];
  if (l32 < (fc + 3u) / 4u) reinterpret_cast<uint32_t*>(slot->face_number)[l32] = reinterpret_cast<const uint32_t*>(g->face_number)[l32];
  if (l32 < (n1 + 3u) / 4u) reinterpret_cast<uint32_t*>(slot->single)[l32] = reinterpret_cast<const uint32_t*>(g->single)[l32];
}

#ifndef HALO_N_VGPR
#define HALO_N_VGPR 1
#endif
#ifndef HALO_PROJ_LDS
#define HALO_PROJ_LDS 1    // round 6: the exit queue's drain of the last-layer plain kernels reads the projection's constants from LDS into VGPRs, not from the
                           // kernarg segment into SGPRs: ~40 scalars fewer live across the drain (the loop's own scalar state is no longer spilled to VGPR lanes
                           // around it) and the projection's FMAs take VGPR operands (an SGPR operand makes a VALU instruction 1.56x dearer on this part).  0 = off
#endif
// One exit that goes to the image: project, accumulate, tally (the tail of CollectData, simulator.cpp:719-760).
template <int MODE, bool MONO, bool SMALLC>
HD int land_exit(const DispatchParams& P, const AccCtx<MONO, SMALLC>& cache, const ColorDev* color, uint64_t cmask, float wx, float wy, float wz, float w,
                 float cmf_x, float cmf_y, float cmf_z, uint32_t wl_idx, RaySums& sums, Probe& pr) {
#if HALO_PROJ_LDS
  const ProjDev& pj = cache.proj != nullptr ? cache.proj->p : P.proj;
#else
  const ProjDev& pj = P.proj;
#endif
  Hits h = project_exit(pj, wx, wy, wz, ModeTraits<MODE>::kFast ? cache.lens : -1, ModeTraits<MODE>::kFast ? cache.vis : -1);
  PROBE_MARK(pr, kPhProject);
  int primary = -1;
  // (0 <= x < w as ONE unsigned compare — a negative coordinate is a huge unsigned one, the image's sides are positive — and the three tests
  //  joined without short-circuit: one branch where `&&` made three nested ones, each with its copies of the ray's sums)
  // ... in the kernels instantiated for a one-hit lens.  The others (dual lenses, the generic-lens kernels) keep the short-circuit form: joined,
  // their register allocation puts the Hits into scratch (dual fisheye: 536 bytes per lane, ms_multi_crystal's last layer 12 % slower), while the
  // one-hit kernels spill with the short-circuit form — so the form follows the instantiation (the test folds at compile time).
  const int lens_k = ModeTraits<MODE>::kFast ? cache.lens : -1;
  const bool one_hit = lens_k >= 0 && lens_k != HALO_LENS_DUAL_FISHEYE_EQUAL_AREA && lens_k != HALO_LENS_DUAL_FISHEYE_EQUIDISTANT &&
                       lens_k != HALO_LENS_DUAL_FISHEYE_STEREOGRAPHIC && lens_k != HALO_LENS_DUAL_FISHEYE_ORTHOGRAPHIC;
  bool in0;
  if (one_hit) in0 = (h.count >= 1) & (static_cast<uint32_t>(h.px0) < static_cast<uint32_t>(pj.img_w)) & (static_cast<uint32_t>(h.py0) < static_cast<uint32_t>(pj.img_h));
  else in0 = h.count >= 1 && static_cast<uint32_t>(h.px0) < static_cast<uint32_t>(pj.img_w) && static_cast<uint32_t>(h.py0) < static_cast<uint32_t>(pj.img_h);
  if (in0) {
    uint32_t pix = static_cast<uint32_t>(h.py0) * static_cast<uint32_t>(pj.img_w) + static_cast<uint32_t>(h.px0);
    accumulate<MONO, SMALLC>(P, cache, pix, wl_idx, w, cmf_x, cmf_y, cmf_z);
    if (ModeTraits<MODE>::kTables && color != nullptr) fan_lanes(P, *color, cmask, pix, cmf_y * w);
    if constexpr (MODE == kModeColor) fan_lanes_fast(P, *cache.fast, cmask, pix, cmf_y * w);
    sums.landed += w;  // bump_landed: primary hit only (scatter_accum.hpp:96-108)
    sums.pix_n++;
    primary = static_cast<int>(pix);
  }
  // (the second hit of a dual lens keeps the short-circuit form: joined like the first, the dual-lens kernels put the Hits into scratch — 536 bytes
  //  per lane — and ms_multi_crystal's last layer ran 12 % slower)
  if (h.count == 2 && static_cast<uint32_t>(h.px1) < static_cast<uint32_t>(pj.img_w) && static_cast<uint32_t>(h.py1) < static_cast<uint32_t>(pj.img_h)) {
    uint32_t pix = static_cast<uint32_t>(h.py1) * static_cast<uint32_t>(pj.img_w) + static_cast<uint32_t>(h.px1);
    accumulate<MONO, SMALLC>(P, cache, pix, wl_idx, w, cmf_x, cmf_y, cmf_z);
    if (ModeTraits<MODE>::kTables && color != nullptr) fan_lanes(P, *color, cmask, pix, cmf_y * w);
    if constexpr (MODE == kModeColor) fan_lanes_fast(P, *cache.fast, cmask, pix, cmf_y * w);
    sums.pix_n++;
  }
  PROBE_MARK(pr, kPhAccum);
  return primary;
}

// The culls of project_exit that need no projection: false = this exit cannot land (conservative at cz ~ 0, where the
// projection itself decides).
HD bool exit_may_land(const ProjDev& p, float wx, float wy, float wz, int lens = -1, int vis = -1) {
  const int t = lens >= 0 ? lens : p.proj_type;
  const int vr = vis >= 0 ? vis : p.visible_range;
  if (t == HALO_LENS_LINEAR || t == HALO_LENS_FISHEYE_EQUAL_AREA || t == HALO_LENS_FISHEYE_EQUIDISTANT || t == HALO_LENS_FISHEYE_STEREOGRAPHIC ||
      t == HALO_LENS_FISHEYE_ORTHOGRAPHIC) {
    if ((vr == HALO_VISIBLE_UPPER && wz > 0.0f) || (vr == HALO_VISIBLE_LOWER && wz < 0.0f)) return false;
    return HALO_FMA(p.rot[8], -wz, HALO_FMA(p.rot[5], -wy, p.rot[2] * (-wx))) > -1e-6f;
  }
  return true;
}

// The dispatch record again, from where it lies.  DispatchParams is the kernel's one argument: it sits in the kernarg segment (constant
// memory, scalar loads).  The compiler loads every field a kernel uses ONCE, at the top, and keeps it in an SGPR for the whole kernel — 200
// live scalars against 102 registers: the headline instantiation spilled 104 of them to VGPR lanes and read them back with v_readlane
// (a VALU slot each, plus s_nop hazards) 77 times per wave-ray.  A region that re-reads the record through an OPAQUE pointer to the
// same memory (the empty asm hides where it points, so the loads can neither be hoisted nor merged with the kernel's) gets its
// fields by s_load where it needs them, and the registers are free everywhere else.
#ifndef HALO_RELOAD
#define HALO_RELOAD 3      // bit 0: the exit queue's drain, bit 1: root generation
#endif
HD DispatchParams reload_params() {
  const DispatchParams __attribute__((address_space(4)))* pc =
      (const DispatchParams __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(pc));
  DispatchParams L;
  __builtin_memcpy(&L, pc, sizeof(DispatchParams));   // scalarised: only the fields the region uses are loaded
  return L;
}

// Pop exits off the wave's queue, one per active lane and round, until fewer than 64 are left (`all`: until it is empty).
// Called where every lane that took part in the pushes is active (their counts agree).
template <int MODE, bool MONO, bool SMALLC>
HD void drain_exits(const DispatchParams& P_kernel, const AccCtx<MONO, SMALLC>& cache, RaySums& sums, bool all, Probe& pr) {
#if HALO_RELOAD & 1
  const DispatchParams P = reload_params();
#else
  const DispatchParams& P = P_kernel;
#endif
  ExitQueue& Q = *cache.q;
  const uint64_t m = __ballot(1);
  const uint32_t k = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
  const uint32_t na = static_cast<uint32_t>(__popcll(m));
  uint32_t qn = sums.qn;
  while (all ? qn > 0u : qn >= 64u) {
    if (k < qn) {
      const uint32_t i = qn - 1u - k;
      const uint32_t wl = Q.wl[i];
      float cx = 0.0f, cy = 0.0f, cz = 0.0f;
      if (!MONO || MODE == kModeColor) {   // X/Y/Z planes (and the colour lanes' Y): the popping lane is not the ray's, so the CMF row comes from the pool (<= 8 KB, cache-resident)
        const WlEntryDev e = P.wl_pool[wl];
        cx = e.cmf_x, cy = e.cmf_y, cz = e.cmf_z;
      }
      uint64_t cm = 0ull;
      if constexpr (MODE == kModeColor) cm = static_cast<uint64_t>(cache.qm->lo[i]) | (static_cast<uint64_t>(cache.qm->hi[i]) << 32);
      land_exit<MODE, MONO, SMALLC>(P, cache, nullptr, cm, Q.x[i], Q.y[i], Q.z[i], Q.w[i], cx, cy, cz, wl, sums, pr);
    }
    qn -= min(na, qn);
  }
  sums.qn = qn;
}



template <int MODE, bool MONO, bool SMALLC>
HD void emit_gate(const DispatchParams& P, const AccCtx<MONO, SMALLC>& cache, const FilterDev* filter, const ColorDev* color, uint64_t carried, Stream& gate, const float* R, bool live,
                  float lx, float ly, float lz, float w, float cmf_x, float cmf_y, float cmf_z, uint32_t wl_idx, uint32_t root, uint32_t seq,
                  const PathView& pv, uint32_t uni_len, RaySums& sums, Probe& pr) {
  // `live`: this lane has an outgoing candidate.  Kernels with an exit queue call this with every lane of the interaction loop
  // (the push is a wave-wide step); the others branch around it here.
  const bool queued = ModeTraits<MODE>::kFast && cache.q != nullptr;
  if (!queued && !live) return;
  // crystal → world (trace_backend.hpp:71-89 invariant: everything leaving the crystal is world-space)
  float wx = HALO_FMA(R[2], lz, HALO_FMA(R[1], ly, R[0] * lx));
  float wy = HALO_FMA(R[5], lz, HALO_FMA(R[4], ly, R[3] * lx));
  float wz = HALO_FMA(R[8], lz, HALO_FMA(R[7], ly, R[6] * lx));
  // physical filter first: a failing exit terminates — neither emitted nor continued (simulator.cpp:689,725-728)
  if (ModeTraits<MODE>::kTables && filter != nullptr) {
    if (!filter_check(*filter, pv, wx, wy, wz, P.crystal_id)) return;
  }
  uint64_t cmask = carried;
  if (ModeTraits<MODE>::kTables && color != nullptr) cmask = color_bits(*color, carried, pv, wx, wy, wz, P.crystal_id);
  if constexpr (ModeTraits<MODE>::kFastPath) {
    // every lane of this interaction has recorded uni_len faces (a stray lane emits where it strays, before its path would differ)
    FastTablesC& F = *cache.fast;
    const FastHdr fh = load_fast_hdr(F);
    const bool ok = fast_filter(F, cache.fast_ee, fh, static_cast<uint32_t>(fh.len_mode >> (2u * uni_len)) & 3u, uni_len, pv.reg, wx, wy, wz, P.crystal_id);
    if (MODE == kModeColor) cmask = fast_color_bits(F, cache.fast_ee, carried, uni_len, pv.reg, wx, wy, wz, P.crystal_id);
    live = live && ok;
    if (!queued && !live) return;
  }
  // prob gate (CollectData simulator.cpp:719): one draw per outgoing candidate; u in [0,1) so prob<=0 never
  // passes and prob>=1 always does — the draw is skipped there without changing any outcome.
  bool pass = false;
  if (!(ModeTraits<MODE>::kFast && cache.nogate) && live && P.prob > 0.0f) pass = (P.prob >= 1.0f) ? true : (uniform(gate) < P.prob);
  if (pass) {
    if (!(ModeTraits<MODE>::kFast && cache.last) && !P.final_layer) {  // "continue" with no next layer is dropped (simulator.cpp:719-722)
      // wave64 ballot compaction: one atomic per wave per emit site, lanes take consecutive slots
      const uint64_t mask = __ballot(1);
      const uint32_t lane = __lane_id();
      const uint32_t leader = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(mask))) - 1u;
      uint32_t base = 0u;
      const uint32_t shard = blockIdx.x & (kContShards - 1);
      if (lane == leader) base = atomicAdd(&P.cont_cnt[shard * kContCntStride], static_cast<uint32_t>(__popcll(mask)));
      base = __shfl(base, static_cast<int>(leader));
      const uint32_t off = base + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
      if (off < P.cont_out_cap) {
        const uint32_t slot = shard * P.cont_out_cap + off;
        const uint32_t st = P.cont_out_stride;
        P.cont_out[slot] = wx;
        P.cont_out[st + slot] = wy;
        P.cont_out[2u * st + slot] = wz;
        P.cont_out[3u * st + slot] = w;
        reinterpret_cast<uint32_t*>(P.cont_out)[4u * st + slot] = wl_idx;
        if (MODE == kModeColor || (ModeTraits<MODE>::kTables && color != nullptr)) {  // the mask rides with the continuation (cu:922,1129)
          reinterpret_cast<uint32_t*>(P.cont_out)[5u * st + slot] = static_cast<uint32_t>(cmask);
          reinterpret_cast<uint32_t*>(P.cont_out)[6u * st + slot] = static_cast<uint32_t>(cmask >> 32);
        }
      }
    }
    if (!queued) return;
  }
  PROBE_MARK(pr, kPhEmitGate);
  if (ModeTraits<MODE>::kFast && cache.none) return;   // (never reached with an exit in hand: prob >= 1 passed it on above)
  if (queued) {
    if (P.prob >= 1.0f) return;   // dispatch-uniform: every candidate of this layer continues, nothing goes to the image
    const bool out = live && !pass;
    sums.exit_w += out ? w : 0.0f;
    sums.exit_n += out ? 1u : 0u;   // (counted per lane: one popcount of the ballot per emit, a scalar add, measured 2.7 % SLOWER at configs[1])
    // (taking the cull's view axis from the LDS copy of the projection as well — HALO_PROJ_LDS — measured the same: 1.668 vs 1.669 ms)
    const bool want = out && exit_may_land(P.proj, wx, wy, wz, cache.lens, cache.vis);
    const uint64_t m = __ballot(want);
    ExitQueue& Q = *cache.q;
    if (want) {
      const uint32_t i = sums.qn + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
      Q.x[i] = wx;
      Q.y[i] = wy;
      Q.z[i] = wz;
      Q.w[i] = w;
      Q.wl[i] = static_cast<uint8_t>(wl_idx);
      if constexpr (MODE == kModeColor) {
        cache.qm->lo[i] = static_cast<uint32_t>(cmask);
        cache.qm->hi[i] = static_cast<uint32_t>(cmask >> 32);
      }
    }
    sums.qn += static_cast<uint32_t>(__popcll(m));
    if (sums.qn >= 64u) drain_exits<MODE, MONO, SMALLC>(P, cache, sums, false, pr);
    return;
  }
  const int primary = land_exit<MODE, MONO, SMALLC>(P, cache, color, cmask, wx, wy, wz, w, cmf_x, cmf_y, cmf_z, wl_idx, sums, pr);
  sums.exit_w += w;
  sums.exit_n++;
  if (MODE == kModeCapture) {
    uint32_t slot = atomicAdd(&P.counters[kCntExit], 1u);
    if (slot < P.exit_cap) {
      HaloExitRecord rec;
      rec.dir[0] = wx;
      rec.dir[1] = wy;
      rec.dir[2] = wz;
      rec.weight = w;
      rec.root = root;
      rec.seq = static_cast<uint16_t>(seq);
      rec.layer = static_cast<uint8_t>(P.layer);
      rec.path_len = static_cast<uint8_t>(pv.len < HALO_PATH_CAP ? pv.len : HALO_PATH_CAP);
      for (int k = 0; k < HALO_PATH_CAP; k++) rec.path[k] = (static_cast<uint32_t>(k) < pv.len) ? path_at(pv, static_cast<uint32_t>(k)) : 0;
      rec.pixel = primary;
      rec.crystal_id = static_cast<uint16_t>(P.crystal_id);
      rec.wl_idx = static_cast<uint16_t>(wl_idx);
      rec.color_mask = cmask;
      P.exits[slot] = rec;
    }
  }
}

// The arithmetic of the entry picks is written out (explicit fma chains, products that feed a running sum rounded on their own): the three
// variants below must pick the same triangle for the same uniform whatever else the instantiation around them compiles, and left to the
// backend the choice between fma(a, b, c * d) and fma(c, d, a * b) follows the surrounding code (round 4: once the regular-prism kernels
// stopped compiling the other two variants, one ray in 3 million entered through the neighbouring triangle).
HD float dot3_fma(const float* d, float x, float y, float z) { return HALO_FMA(d[2], z, HALO_FMA(d[1], y, d[0] * x)); }
HD float tri_point(float u, float v, float a, float b, float c) { return HALO_FMA(u, b - a, HALO_FMA(v, c - a, a)); }

// Projected-area categorical entry pick + uniform point (InitRay_p_fid simulator.cpp:133-192 in its device form
// gen_root_kernel cu:1556-1597).  Two passes over the fan table instead of a 64-float private array.
template <typename ShapePtr>
HD int sample_entry(Stream& s, ShapePtr sh, int tri_cnt, const float* d, float* p) {
  const float u_cat = uniform(s);
  if (tri_cnt == 0) {
    p[0] = p[1] = p[2] = 0.0f;
    return -1;
  }
  float total = 0.0f;
  for (int t = 0; t < tri_cnt; ++t) {
    const float4 na = *reinterpret_cast<const float4*>(sh->tri_na[t]);
    const float dot = dot3_fma(d, na.x, na.y, na.z);
    total += fmaxf(mul_rn(-dot, na.w), 0.0f);
  }
  int tri = 0;
  if (total > 0.0f) {
    const float target = u_cat * total;
    float cum = 0.0f;
    tri = tri_cnt - 1;
    for (int t = 0; t < tri_cnt; ++t) {
      const float4 na = *reinterpret_cast<const float4*>(sh->tri_na[t]);
      const float dot = dot3_fma(d, na.x, na.y, na.z);
      cum += fmaxf(mul_rn(-dot, na.w), 0.0f);
      if (cum > target) {
        tri = t;
        break;
      }
    }
  }
  float u = uniform(s);
  float v = uniform(s);
  if (u + v > 1.0f) {
    u = 1.0f - u;
    v = 1.0f - v;
  }
  const float* vt = sh->tri_v[tri];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float a = vt[k], b = vt[3 + k], c = vt[6 + k];
    p[k] = tri_point(u, v, a, b, c);
  }
  return static_cast<int>(sh->tri_face[tri]);
}

// The same categorical pick, face by face.  All fan triangles of a face share its plane, so their weights are
// max(-d.n_f, 0) * area_t: one dot product per face decides how much of the target falls into it, and the triangle inside the
// face is found on the areas alone.  Same uniform, same cumulative order (faces in table order, their triangles in table
// order) as the flat walk; the two differ only by the rounding of the partial sums (and of n_f against the per-triangle
// normals), i.e. in which of two adjacent triangles a ray within ~1e-7 of a boundary lands.
template <typename ShapePtr>
HD int sample_entry_by_face(Stream& s, ShapePtr sh, const FaceIndex& fi, int face_cnt, int tri_cnt, const float* d, float* p) {
  const float u_cat = uniform(s);
  if (tri_cnt == 0) {
    p[0] = p[1] = p[2] = 0.0f;
    return -1;
  }
  float total = 0.0f;
  for (int f = 0; f < face_cnt; ++f) {
    const float4 pl = *reinterpret_cast<const float4*>(sh->face[f]);
    total += mul_rn(fmaxf(-dot3_fma(d, pl.x, pl.y, pl.z), 0.0f), fi.area[f]);
  }
  int tri = 0;
  if (total > 0.0f) {
    const float target = u_cat * total;
    float cum = 0.0f;
    tri = tri_cnt - 1;
    for (int f = 0; f < face_cnt; ++f) {
      const float4 pl = *reinterpret_cast<const float4*>(sh->face[f]);
      const float c = fmaxf(-dot3_fma(d, pl.x, pl.y, pl.z), 0.0f);
      const float w = mul_rn(c, fi.area[f]);
      if (cum + w > target) {   // c > 0 here
        const float r = (target - cum) * fast_rcp(c);
        const int t0 = fi.tri0[f], tn = fi.tric[f];
        float acc = 0.0f;
        tri = t0 + tn - 1;
        for (int k = 0; k < tn; ++k) {
          acc += sh->tri_na[t0 + k][3];
          if (acc > r) {
            tri = t0 + k;
            break;
          }
        }
        break;
      }
      cum += w;
    }
  }
  float u = uniform(s);
  float v = uniform(s);
  if (u + v > 1.0f) {
    u = 1.0f - u;
    v = 1.0f - v;
  }
  const float* vt = sh->tri_v[tri];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float a = vt[k], b = vt[3 + k], c = vt[6 + k];
    p[k] = tri_point(u, v, a, b, c);
  }
  return static_cast<int>(sh->tri_face[tri]);
}

// The same pick for a FULL prism (EntryFastDev, halo_device.h): one dot product per slab, the four products stay in registers,
// and the seven candidate weights (face 0 or 1, then sides 2, 3, 4, then their opposites 5, 6, 7 — face order) are walked without
// a branch.  Weights, partial sums and the uniform are those of sample_entry_by_face: the unlit face of a slab contributes an
// exact zero there.
// EF: EntryFastDev (one shape per dispatch, host-built) or SlotFast (a sampled prism's, rebuilt by its half-wave every pass — below)
template <typename ShapePtr, typename EF>
HD int sample_entry_prism(Stream& s, ShapePtr sh, const EF& ef, int tri_cnt, const float* d, float* p) {
  const float u_cat = uniform(s);
  float dn[4], ap[4], am[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float4 g = *reinterpret_cast<const float4*>(sh->slab[k]);
    dn[k] = dot3_fma(d, g.x, g.y, g.z);
    const float2v a = *reinterpret_cast<const float2v*>(ef.slab_area[k]);
    ap[k] = a.x;
    am[k] = a.y;
  }
  // candidate k: 0 = basal slab (face 0 or 1), 1..3 = sides 2..4 (lit when d.n < 0), 4..6 = their opposites 5..7
  float wgt[7], cs[7];
  cs[0] = fabsf(dn[0]);
  wgt[0] = mul_rn(cs[0], dn[0] < 0.0f ? ap[0] : am[0]);
#pragma unroll
  for (int k = 1; k < 4; k++) {
    cs[k] = fmaxf(-dn[k], 0.0f);
    wgt[k] = mul_rn(cs[k], ap[k]);
    cs[3 + k] = fmaxf(dn[k], 0.0f);
    wgt[3 + k] = mul_rn(cs[3 + k], am[k]);
  }
  float total = wgt[0];
#pragma unroll
  for (int k = 1; k < 7; k++) total += wgt[k];
  const float target = u_cat * total;
  int face = 7;
  float rem = 0.0f, c = 1.0f, cum = 0.0f;
  bool found = false;
#pragma unroll
  for (int k = 0; k < 7; k++) {
    const float nc = cum + wgt[k];
    const bool hit = !found && (nc > target);
    const int f = (k == 0) ? (dn[0] < 0.0f ? 0 : 1) : (k + 1);
    face = hit ? f : face;
    rem = hit ? (target - cum) : rem;
    c = hit ? cs[k] : c;
    found = found || hit;
    cum = nc;
  }
  int tri;
  if (!(total > 0.0f)) {   // no lit face: the flat walk's tri = 0
    tri = 0;
    face = 0;
  } else {
    const uint32_t t0n = ef.tri0n[face];
    const int t0 = static_cast<int>(t0n & 0xFFu), tn = static_cast<int>(t0n >> 8);
    tri = t0 + tn - 1;
    if (found) {
      const float r = rem * fast_rcp(c);
      const float4 ar = *reinterpret_cast<const float4*>(ef.tri_area[face]);
      const float a0 = ar.x, a1 = a0 + ar.y, a2 = a1 + ar.z;   // zero-padded past tn: the partial sum stops growing
      tri = (a2 > r) ? t0 + 2 : tri;
      tri = (a1 > r) ? t0 + 1 : tri;
      tri = (a0 > r) ? t0 : tri;
      tri = min(tri, t0 + tn - 1);
    } else {
      tri = tri_cnt - 1;   // the walk ran off the end (cannot happen for finite weights): last triangle, like the flat walk
    }
  }
  float u = uniform(s);
  float v = uniform(s);
  if (u + v > 1.0f) {
    u = 1.0f - u;
    v = 1.0f - v;
  }
  if constexpr (std::is_same<EF, EntryFastDev>::value) {
    const float4* vt = reinterpret_cast<const float4*>(ef.tri_v[tri]);
    const float4 q0 = vt[0], q1 = vt[1], q2 = vt[2];   // a = q0.xyz, b = (q0.w, q1.x, q1.y), c = (q1.z, q1.w, q2.x)
    p[0] = tri_point(u, v, q0.x, q0.w, q1.z);
    p[1] = tri_point(u, v, q0.y, q1.x, q1.w);
    p[2] = tri_point(u, v, q0.z, q1.y, q2.x);
  } else {   // the slot's own 36-byte corner rows
    const float* vt = sh->tri_v[tri];
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = tri_point(u, v, vt[k], vt[3 + k], vt[6 + k]);
  }
  return face;
}

#ifndef HALO_POOL_HEXN
#define HALO_POOL_HEXN 1   // sampled full prisms with the regular prism's normals search their next face with literal normals (A/B knob)
#endif
// The entry pick's view of a sampled FULL prism, per half-wave: EntryFastDev's first three tables (BuildEntryFast, halo_host.cpp), rebuilt from
// the slot's fan table at the top of every pass by the half-wave itself — ~100 wave instructions against the ~300 the walk over all 20 fan
// triangles (two passes, an LDS row each) costs every ray more than the slab-wise pick.  ok = 0 (a prism that lost a face to its neighbours'
// distances, a fan the pick cannot index): the half-wave's rays take the walk over triangles as before.
struct SlotFast {
  float tri_area[kEntryFastFaces][4];
  uint32_t tri0n[kEntryFastFaces];
  float slab_area[4][2];
  float d_plus[4], d_minus[4];   // the slabs' plane constants, for the literal-normal next-face search (hexn)
  uint32_t ok;     // a full eight-face prism whose fan the slab-wise entry pick can index
  uint32_t hexn;   // ... whose slab normals are the regular prism's, bit for bit (a sampled prism varies its face DISTANCES): the next-face search
                   // then has the normals as literals like the one-shape regular-prism kernels, with a plane constant per face
  uint32_t pad[2];
};
static_assert(sizeof(SlotFast) % 16 == 0, "rows are read as float4 / float2");
// all 32 lanes of the half-wave call this together, after the slot's rows are in place
HD void build_slot_fast(SlotFast* sf, const ShapePrism* sh, uint32_t l32) {
  const int tc = sh->tri_cnt;
  bool ok = sh->face_cnt == kEntryFastFaces && sh->slab_cnt == 4 && sh->single_cnt == 0 && tc >= 8 && tc <= kEntryFastTris;
#pragma unroll
  for (int k = 0; k < 4; k++) {   // slabs (0,1), (2,5), (3,6), (4,7): what the pick's candidate order assumes
    const int ip = __float_as_int(sh->slab[k][5]), im = __float_as_int(sh->slab[k][6]);
    ok = ok && ip == (k == 0 ? 0 : k + 1) && im == (k == 0 ? 1 : k + 4);
  }
  const uint32_t tf = (ok && static_cast<int>(l32) < tc) ? sh->tri_face[l32] : 0xFFu;   // lane t holds triangle t's face
  const uint32_t half_shift = threadIdx.x & 32u;
  uint32_t next = 0u, my_t0 = 0u, my_tn = 0u;
#pragma unroll
  for (uint32_t f = 0; f < static_cast<uint32_t>(kEntryFastFaces); f++) {   // triangles grouped face by face, in face order, 1..4 per face
    const uint32_t m = static_cast<uint32_t>(__ballot(tf == f) >> half_shift);
    const uint32_t tn = static_cast<uint32_t>(__popc(m));
    ok = ok && tn >= 1u && tn <= 4u && m == (((1u << tn) - 1u) << next);
    if (l32 == f) {
      my_t0 = next;
      my_tn = tn;
    }
    next += tn;
  }
  ok = ok && next == static_cast<uint32_t>(tc);
  float area = 0.0f;
  if (ok && l32 < static_cast<uint32_t>(kEntryFastFaces)) {
    float a[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
      a[k] = k < my_tn ? sh->tri_na[my_t0 + k][3] : 0.0f;
      area += a[k];   // (the host's partial sums in the same order; the padding adds exact zeros)
    }
    *reinterpret_cast<float4*>(sf->tri_area[l32]) = make_float4(a[0], a[1], a[2], a[3]);
    sf->tri0n[l32] = my_t0 | (my_tn << 8);
  }
  // slab k's areas: {its +n face, its -n face}
  const uint32_t kk = l32 & 3u;
  const int base = static_cast<int>(threadIdx.x & 32u);
  const float ap = __shfl(area, base | static_cast<int>(kk == 0u ? 0u : kk + 1u)), am = __shfl(area, base | static_cast<int>(kk == 0u ? 1u : kk + 4u));
  if (ok && l32 < 4u) *reinterpret_cast<float2v*>(sf->slab_area[l32]) = float2v{ap, am};
  bool hexn = ok;
  if (ok && l32 < 4u) {   // lane k: slab k's normal against the literal the search uses, its two plane constants to the table
    constexpr float kS60 = 0.86602540378443864676f;
    const float4 g = *reinterpret_cast<const float4*>(sh->slab[l32]);
    const float wx = l32 == 1u ? 1.0f : l32 == 2u ? 0.5f : l32 == 3u ? -0.5f : 0.0f, wy = l32 >= 2u ? kS60 : 0.0f, wz = l32 == 0u ? 1.0f : 0.0f;
    hexn = g.x == wx && g.y == wy && g.z == wz;
    sf->d_plus[l32] = g.w;
    sf->d_minus[l32] = sh->slab[l32][4];
  }
  hexn = (static_cast<uint32_t>(__ballot(!hexn) >> half_shift) & 0xFu) == 0u && ok && HALO_POOL_HEXN != 0;
  if (l32 == 0u) {
    sf->ok = ok ? 1u : 0u;
    sf->hexn = hexn ? 1u : 0u;
  }
}

// The next pass's pool record, a pass ahead (prism pools under the hit log, round 5).  A pass of a shape-pool kernel used to BEGIN with the staging
// copy of its record: global loads, s_waitcnt vmcnt(0), LDS writes.  On gfx950 loads and stores share vmcnt and complete in order, so that wait
// was also a wait for every hit-log store the wave had issued up to the last instruction of the pass before — 19 % of the waves' resident time
// (tools/phase_probe.py stoch).  Now the record of pass k+1 is requested at the TOP of pass k, when the only stores in flight are a pass old, rides
// in twelve registers through the generation phase (orientation, sun, entry pick: no vector-memory wait of the compiler's in there) and lands in
// LDS right before the interaction loop issues this pass's first store: the rows only the entry pick reads (fan corners, normals and areas, 66
// of the record's 85 sixteen-byte rows) go straight into the half-wave's slot — this pass is done with them — and the 19 rows the interaction
// loop still reads (header, face and slab rows, face numbers) wait in a 304-byte mirror that the next pass copies over, LDS to LDS.  The loads
// are ordinary loads (the compiler waits for them where land() uses them — and before any copy of the registers it might decide to make;
// hand-written asm loads with an explicit s_waitcnt measured the same and would not have that guarantee).  What makes it work is the ORDER of a
// pass: nothing between request() and land() waits on a vector-memory load — the pool entry comes from LDS, loads of other ray sources are
// waited for where they are issued (HALO_ARRIVED).
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr uint32_t kPrismRows = sizeof(ShapePrism) / 16u;                                    // 85
constexpr uint32_t kPrismHotLo = (16u + sizeof(ShapePrism::face) + sizeof(ShapePrism::slab)) / 16u;   // rows [0, 17): header, face, slab
constexpr uint32_t kPrismHotHi = offsetof(ShapePrism, face_number) / 16u;                      // rows [83, 85): face_number, single (the row shares its first bytes with tri_face's tail, dead by then)
constexpr uint32_t kPrismHotRows = kPrismHotLo + (kPrismRows - kPrismHotHi);                  // 19
static_assert(offsetof(ShapePrism, tri_v) == kPrismHotLo * 16u && kPrismRows == 85u && kPrismHotHi == 83u && kPrismRows <= 96u, "ShapePrism row map");
HD bool prism_row_hot(uint32_t row) { return row < kPrismHotLo || row >= kPrismHotHi; }
HD uint32_t prism_hot_index(uint32_t row) { return row < kPrismHotLo ? row : row - kPrismHotHi + kPrismHotLo; }
struct NextShape {
  const f4v* src;   // the next pass's pool record (nullptr: this was the half-wave's last pass)
  f4v* slot;        // the half-wave's LDS slot, as rows
  f4v* mirror;      // kPrismHotRows rows
  uint32_t l32;
  f4v r0, r1, r2;   // rows l32, l32 + 32, l32 + 64
  HD void request() {
    r0 = r1 = r2 = f4v{0.0f, 0.0f, 0.0f, 0.0f};
    if (src != nullptr) {
      const f4v* a = src + l32;
      r0 = a[0];
      r1 = a[32];
      if (l32 + 64u < kPrismRows) r2 = a[64];
    }
  }
  HD void put(uint32_t row, const f4v& v) {
    if (prism_row_hot(row)) mirror[prism_hot_index(row)] = v;
    else slot[row] = v;
  }
  HD void land() {
    if (src != nullptr) {
      put(l32, r0);
      slot[l32 + 32u] = r1;   // rows 32..63: fan corners
      if (l32 + 64u < kPrismRows) put(l32 + 64u, r2);
    }
  }
  HD void late() {}   // (nothing of a prism's record waits in registers)
  static constexpr uint32_t kMirrorRows = kPrismHotRows;
  // the mirror's rows go to their places in the slot: the first thing a pass does (the interaction loop that read the old ones is a pass behind)
  HD static void take_mirror(f4v* slot, const f4v* mirror, uint32_t l32) {
    if (l32 < kPrismHotRows) slot[l32 < kPrismHotLo ? l32 : l32 - kPrismHotLo + kPrismHotHi] = mirror[l32];
  }
};
// The same for a general pool shape: record ShapeDev (4 KB), slot ShapeSlot48 (96 rows = three per lane; the corner rows stay in the record).
// Slot row -> record bytes: rows 0..40 (header, face, slab) lie where they lie in the record; row 41 is the slot's pointer to the record's corner
// rows; 42..89 = tri_na[0..47]; 90..92 = tri_face[0..47]; 93..95 = face_number | single | pad (contiguous in the record too).  The interaction
// loop reads rows 0..40 and 93..95.  Rows 0..31 — each lane's first — stay in four registers through the loop and go to the slot BEHIND it
// (late()); rows 32..40 and 93..95 wait in a 192-byte mirror; everything else lands before the loop like a prism's rows.
constexpr uint32_t kSlot48Rows = sizeof(ShapeSlot48) / 16u, kSlot48PtrRow = offsetof(ShapeSlot48, tri_v) / 16u, kSlot48NaRow = offsetof(ShapeSlot48, tri_na) / 16u,
                   kSlot48FaceRow = offsetof(ShapeSlot48, tri_face) / 16u, kSlot48NumRow = offsetof(ShapeSlot48, face_number) / 16u;
constexpr uint32_t kSlot48MirrorRows = (kSlot48PtrRow - 32u) + (kSlot48Rows - kSlot48NumRow);   // 9 + 3
static_assert(kSlot48Rows == 96u && kSlot48PtrRow == 41u && kSlot48NaRow == 42u && kSlot48FaceRow == 90u && kSlot48NumRow == 93u, "ShapeSlot48 row map");
static_assert(offsetof(ShapeDev, tri_v) == kSlot48PtrRow * 16u && offsetof(ShapeDev, tri_na) % 16u == 0 && offsetof(ShapeDev, tri_face) % 16u == 0 &&
              offsetof(ShapeDev, face_number) % 16u == 0 && offsetof(ShapeDev, single) == offsetof(ShapeDev, face_number) + kMaxFaces &&
              offsetof(ShapeSlot48, single) == offsetof(ShapeSlot48, face_number) + kMaxFaces, "ShapeDev rows the slot takes");
HD uint32_t slot48_record_offset(uint32_t row) {   // (row != kSlot48PtrRow)
  return row < kSlot48PtrRow ? row * 16u
         : row < kSlot48FaceRow ? static_cast<uint32_t>(offsetof(ShapeDev, tri_na)) + (row - kSlot48NaRow) * 16u
         : row < kSlot48NumRow ? static_cast<uint32_t>(offsetof(ShapeDev, tri_face)) + (row - kSlot48FaceRow) * 16u
                               : static_cast<uint32_t>(offsetof(ShapeDev, face_number)) + (row - kSlot48NumRow) * 16u;
}
struct NextShape48 {
  const f4v* src;   // the next pass's pool record (nullptr: this was the half-wave's last pass)
  f4v* slot;
  f4v* mirror;      // kSlot48MirrorRows rows
  uint32_t l32;
  f4v r0, r1, r2;   // slot rows l32, l32 + 32, l32 + 64
  static constexpr uint32_t kMirrorRows = kSlot48MirrorRows;
  HD void request() {
    r0 = r1 = r2 = f4v{0.0f, 0.0f, 0.0f, 0.0f};
    if (src != nullptr) {
      const char* base = reinterpret_cast<const char*>(src);
      const char* a0 = base + l32 * 16u;
      const char* a1 = base + slot48_record_offset(l32 + 32u == kSlot48PtrRow ? 0u : l32 + 32u);   // (the pointer row's lane loads row 0 again and drops it)
      const char* a2 = base + slot48_record_offset(l32 + 64u);
      r0 = *reinterpret_cast<const f4v*>(a0);
      r1 = *reinterpret_cast<const f4v*>(a1);
      r2 = *reinterpret_cast<const f4v*>(a2);
    }
  }
  HD void land() {   // before the interaction loop
    if (src != nullptr) {
      if (l32 == 0u) {   // the header: counts clamped to the slot's capacity, as stage_shape does
        int32_t* h = reinterpret_cast<int32_t*>(&r0);
        h[0] = min(h[0], kMaxFaces);
        h[1] = min(h[1], 48);
        h[2] = min(h[2], kMaxSlabs);
        h[3] = min(h[3], kMaxFaces);
      }
      const uint32_t row1 = l32 + 32u, row2 = l32 + 64u;
      if (row1 < kSlot48PtrRow) mirror[row1 - 32u] = r1;
      else if (row1 == kSlot48PtrRow) {
        const uint64_t pv = reinterpret_cast<uint64_t>(reinterpret_cast<const char*>(src) + offsetof(ShapeDev, tri_v));
        reinterpret_cast<uint64_t*>(slot + kSlot48PtrRow)[0] = pv;
      } else slot[row1] = r1;
      if (row2 >= kSlot48NumRow) mirror[(kSlot48PtrRow - 32u) + (row2 - kSlot48NumRow)] = r2;
      else slot[row2] = r2;
    }
  }
  HD void late() {   // behind the interaction loop: rows 0..31
    if (src != nullptr) {
      asm volatile("" : "+v"(r0));
      slot[l32] = r0;
    }
  }
  HD static void take_mirror(f4v* slot, const f4v* mirror, uint32_t l32) {
    if (l32 < kSlot48MirrorRows) slot[l32 < kSlot48PtrRow - 32u ? 32u + l32 : kSlot48NumRow + (l32 - (kSlot48PtrRow - 32u))] = mirror[l32];
  }
};

// "This load has arrived": an empty asm that uses the value, right behind the load.  A load the compiler may still count as pending when a pass
// ends (the pass's early exits skip the code that would have waited for it) makes it open EVERY pass of the ray loop with s_waitcnt vmcnt(0) —
// on gfx950, where loads and stores share that counter, a wait for every hit-log and continuation STORE the wave has in flight, although at run
// time no load is pending there.  With the wait pinned inside the branch that issued the load the top of the loop has nothing to wait for.
// Used by the kernels that fetch their next pool record ahead (`pinned` = their NextShape is there), which must not wait before it has landed:
// configs[4]'s trace kernel 2.40 -> 2.34 ms.  NOT by the others: measured on the one-shape kernels, the loop without that wait is SLOWER
// (configs[1] 1.77 -> 1.85 ms per launch, configs[2] 9.15 -> 9.57) — the wait per pass meters the waves' stores; without it they queue up behind
// the memory pipeline inside the interaction loop, where a stall costs more (tools/phase_probe.py cfg1 puts 38 % of the resident time in it, yet
// the kernel is VALU-issue-bound: the other waves use the slots).
#define HALO_ARRIVED(pinned, x) do { if (pinned) asm volatile("" : "+v"(x)); } while (0)

struct Wl0 {   // entry 0 of the wavelength pool and 1 / n, loaded once per kernel
  WlEntryDev e;
  float inv_n;
};

template <int MODE, bool MONO, bool SMALLC, bool HEX, typename ShapePtr, typename NextT = NextShape>
HD void trace_one(const DispatchParams& P, LdsTables<MONO, SMALLC>& T, const AccCtx<MONO, SMALLC>& acc, const FilterDev* filter, const ColorDev* color, ShapePtr sh,
                  const Wl0& wl0, uint32_t tid, RaySums& sums, Probe& pr, NextT* next = nullptr, const WlEntryDev* wl_lds = nullptr,
                  const SlotFast* slot_fast = nullptr) {
  const bool pinned = next != nullptr;   // (a compile-time constant after inlining: see HALO_ARRIVED)
  if (next != nullptr) next->request();
  uint64_t carried = 0ull;  // raypath-colour mask inherited from the previous scattering layers
  float R[9], d[3], p[3], w;
  int face;
  uint32_t wl_idx = 0u;
  const int face_cnt = sh->face_cnt;
#if HALO_RELOAD & 2
  const DispatchParams G = reload_params();   // root generation reads its share of the dispatch record here, once per pass (see reload_params)
#else
  const DispatchParams& G = P;
#endif
  Stream gate = make_stream(G.gate_seed, G.gate_lo, G.gate_hi, tid);

  const uint32_t source = G.source;
  if (source == kSrcGen) {
    Stream s = make_stream(G.gen_seed, G.gen_lo, G.gen_hi, tid);
    // per-ray wavelength in its own seed domain (BuildWlStream pcg_shared.h:213-219)
    Stream wls = s;
    wls.seed ^= kNonceWl;
    if (G.wl_pool_size > 1u) {  // a one-entry pool needs no draw: floor(u * 1) is 0 for every u in [0, 1) (own stream, nothing to keep aligned)
      wl_idx = static_cast<uint32_t>(uniform(wls) * static_cast<float>(G.wl_pool_size));
      if (wl_idx >= G.wl_pool_size) wl_idx = G.wl_pool_size - 1u;
    }
    float lon, lat, roll;
    PROBE_MARK(pr, kPhStream);
    sample_lat_lon_roll(s, G, T.lut, lon, lat, roll);
    PROBE_MARK(pr, kPhOrient);
    build_crystal_rotation(lon, lat, roll, R);
    PROBE_MARK(pr, kPhRotation);
    // sun cone (sample_sph_cap pcg_shared.h:514-529; trig of the fixed sun angles is host-evaluated)
    float u = uniform(s);
    float x = add_rn(u, mul_rn(1.0f - u, G.c_cap));  // separately rounded: see the note on r below
    // x is within 1e-5 of 1: `1 - x*x` cancels catastrophically, so keep the product separately rounded like the
    // reference's host evaluation (a contracted fma here moves r by up to ~2e-4 for rays near the cone axis)
    float r = fast_sqrt(fmaxf(add_rn(1.0f, -mul_rn(x, x)), 0.0f));
    float phi = uniform(s) * 2.0f * kPiF;
    float sp, cp;
    sincos_small(phi, &sp, &cp);
    float y = cp * r, z = sp * r;
    float dwx = HALO_FMA(-(G.c_lon * G.s_lat), z, HALO_FMA(-G.s_lon, y, (G.c_lon * G.c_lat) * x));
    float dwy = HALO_FMA(-(G.s_lon * G.s_lat), z, HALO_FMA(G.c_lon, y, (G.s_lon * G.c_lat) * x));
    float dwz = HALO_FMA(G.c_lat, z, G.s_lat * x);
    apply_inverse(R, dwx, dwy, dwz, d);
    PROBE_MARK(pr, kPhSun);
    if constexpr (HEX) face = sample_entry_prism(s, sh, T.efast, sh->tri_cnt, d, p);   // a regular prism always comes with its EntryFastDev (halo_backend.cpp)
    else if (slot_fast != nullptr) face = slot_fast->ok ? sample_entry_prism(s, sh, *slot_fast, sh->tri_cnt, d, p) : sample_entry(s, sh, sh->tri_cnt, d, p);
    else if (G.entry_fast != nullptr) face = sample_entry_prism(s, sh, T.efast, sh->tri_cnt, d, p);
    else face = T.fidx.ok ? sample_entry_by_face(s, sh, T.fidx, face_cnt, sh->tri_cnt, d, p) : sample_entry(s, sh, sh->tri_cnt, d, p);
    if (next != nullptr) w = 0.0f;   // (kernels that fetch their next record ahead read the pool entry once, below, from LDS when they can)
    else w = (G.wl_pool_size == 1u) ? wl0.e.spd_weight : G.wl_pool[wl_idx].spd_weight;
    PROBE_MARK(pr, kPhEntry);
  } else if (source == kSrcTransit) {
    Stream s = make_stream(G.transit_seed, G.transit_lo, G.transit_hi, tid);
    const uint32_t pos = G.ci_start + tid;
    // Recombine's shuffle, applied as a gather at read time.  The reference permutes single rays (shuffle_cont_kernel
    // cu:1633-1657, out[tid] = in[feistel(tid)]); a per-ray random gather costs five scattered 4-byte reads per ray (one
    // 64 B sector each: 76 GB for 237 M rays).  Here the same Feistel bijection permutes CHUNKS of 2^shuffle_chunk_log2
    // (default 32) consecutive pool entries, so the five plane reads of a half-wave stay one 128 B line each.  A chunk holds
    // rays of 32 different roots (lanes of one ballot-compacted append), so nothing the shuffle is there to break up —
    // position ranges assigned to crystal entries, the K-shape clock — sees correlated neighbours.  The tail past the last
    // full chunk keeps its place.  Option "shuffle_chunk" = 1 gives the reference's per-ray permutation (A/B in the tests).
    uint32_t logical = pos;
    if (G.shuffle) {
      const uint32_t cl = G.shuffle_chunk_log2;
      const uint32_t chunks = G.cont_in_n >> cl;
      if (pos < (chunks << cl)) logical = (feistel_bijection(pos >> cl, chunks, G.shuffle_seed) << cl) + (pos & ((1u << cl) - 1u));
    }
    uint32_t sh_i = 0u;  // largest shard with seg[shard] <= logical (empty shards repeat their neighbour's start)
#pragma unroll
    for (uint32_t step = kContShards / 2; step >= 1u; step >>= 1)
      if (T.seg[sh_i + step] <= logical) sh_i += step;
    const uint32_t src = sh_i * G.cont_in_region + (logical - T.seg[sh_i]);
    const uint32_t st = G.cont_in_stride;
    float dwx = G.cont_in[src], dwy = G.cont_in[st + src], dwz = G.cont_in[2u * st + src];
    w = G.cont_in[3u * st + src];
    wl_idx = reinterpret_cast<const uint32_t*>(G.cont_in)[4u * st + src];
    if (MODE == kModeColor || (ModeTraits<MODE>::kTables && color != nullptr)) {
      uint32_t c_lo = reinterpret_cast<const uint32_t*>(G.cont_in)[5u * st + src], c_hi = reinterpret_cast<const uint32_t*>(G.cont_in)[6u * st + src];
      HALO_ARRIVED(pinned, c_lo);
      HALO_ARRIVED(pinned, c_hi);
      carried = static_cast<uint64_t>(c_lo) | (static_cast<uint64_t>(c_hi) << 32);
    }
    HALO_ARRIVED(pinned, dwx);
    HALO_ARRIVED(pinned, dwy);
    HALO_ARRIVED(pinned, dwz);
    HALO_ARRIVED(pinned, w);
    HALO_ARRIVED(pinned, wl_idx);
    float lon, lat, roll;
    PROBE_MARK(pr, kPhStream);
    sample_lat_lon_roll(s, G, T.lut, lon, lat, roll);
    PROBE_MARK(pr, kPhOrient);
    build_crystal_rotation(lon, lat, roll, R);
    PROBE_MARK(pr, kPhRotation);
    apply_inverse(R, dwx, dwy, dwz, d);
    PROBE_MARK(pr, kPhSun);
    if constexpr (HEX) face = sample_entry_prism(s, sh, T.efast, sh->tri_cnt, d, p);   // a regular prism always comes with its EntryFastDev (halo_backend.cpp)
    else if (slot_fast != nullptr) face = slot_fast->ok ? sample_entry_prism(s, sh, *slot_fast, sh->tri_cnt, d, p) : sample_entry(s, sh, sh->tri_cnt, d, p);
    else if (G.entry_fast != nullptr) face = sample_entry_prism(s, sh, T.efast, sh->tri_cnt, d, p);
    else face = T.fidx.ok ? sample_entry_by_face(s, sh, T.fidx, face_cnt, sh->tri_cnt, d, p) : sample_entry(s, sh, sh->tri_cnt, d, p);
    PROBE_MARK(pr, kPhEntry);
  } else {  // kSrcHost: crystal-local golden rays, identity rotation (cpu_trace_backend.cpp:121-144)
    R[0] = R[4] = R[8] = 1.0f;
    R[1] = R[2] = R[3] = R[5] = R[6] = R[7] = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      d[k] = G.host_d[static_cast<size_t>(tid) * 3 + k];
      p[k] = G.host_p[static_cast<size_t>(tid) * 3 + k];
    }
    w = G.host_w[tid];
    face = static_cast<int>(G.host_tf[tid]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      HALO_ARRIVED(pinned, d[k]);
      HALO_ARRIVED(pinned, p[k]);
    }
    HALO_ARRIVED(pinned, w);
    HALO_ARRIVED(pinned, face);
  }
  if (face < 0 || face >= face_cnt) {   // empty crystal / invalid entry face: contributes nothing (its share of the next record's copy still lands)
    if (next != nullptr) {
      next->land();
      next->late();
    }
    return;
  }

  // the wavelength pool (<= 8 KB) is read where it lies: a couple of L1-resident reads per ray — staging it cost every workgroup
  // 8 KB of LDS.  The one entry of a discrete session comes in registers (Wl0): a load here would also make the wave wait for
  // every store it has in flight (hit-log records, continuation rays) once per pass
  WlEntryDev wle = wl0.e;   // a one-entry pool (discrete wavelength): read once per kernel, not per pass
  float inv_n = wl0.inv_n;
  if (next != nullptr) {
    // A pass of these kernels must not WAIT for a load before its next record has landed (NextShape): the pool of an illuminant session (<= 64
    // entries) is staged in LDS; a larger one is read where it lies, and the wait for it stays inside this branch (the empty asm is a use).
    if (P.wl_pool_size != 1u) {
      if (wl_lds != nullptr) {
        wle = wl_lds[wl_idx];
      } else {
        wle = P.wl_pool[wl_idx];
        asm volatile("" : "+v"(wle.n_idx), "+v"(wle.spd_weight), "+v"(wle.cmf_x), "+v"(wle.cmf_y), "+v"(wle.cmf_z));
      }
      inv_n = 1.0f / wle.n_idx;
    }
    if (source == kSrcGen) w = wle.spd_weight;
  } else
  if (P.wl_pool_size != 1u) {
    wle = P.wl_pool[wl_idx];
    inv_n = 1.0f / wle.n_idx;  // once per ray, IEEE like the reference
  }
#if HALO_N_VGPR   // the refractive index (dispatch-uniform for a discrete wavelength: the compiler keeps it in an SGPR) pinned into a VGPR: the Fresnel
                  // split's ~10 operations per interaction that take it then have VGPR operands only (an SGPR operand makes a VALU instruction 1.56x
                  // dearer on this part) — configs[1] 1.690 -> 1.671 ms per launch, configs[2] and [4] -1 %; same values (round 6)
  float n_idx_pin = wle.n_idx;
  asm volatile("" : "+v"(n_idx_pin));
  const float n_idx = n_idx_pin;
#else
  const float n_idx = wle.n_idx;
#endif
  const float cmf_x = wle.cmf_x, cmf_y = wle.cmf_y, cmf_z = wle.cmf_z;

  uint8_t path[ModeTraits<MODE>::kTables ? kFilterPathCap : 1];
  PathView pv = {path, 0u, {0ull, 0ull}};
  if constexpr (MODE != kModePlain) {
    if constexpr (HEX) pv.reg.lo = static_cast<uint32_t>(face) + 1u;   // regular prism: face numbers 1..8 in face order (BuildEntryFast checks)
    else pv.reg.lo = sh->face_number[face];
    pv.len = 1u;
  }

  bool done = false;
  const bool slot_hexn = slot_fast != nullptr && slot_fast->hexn != 0u;
  float hex_d_basal = 0.0f, hex_d_side = 0.0f;
  if constexpr (HEX) {
    hex_d_basal = T.efast.hex_d_basal;
    hex_d_side = T.efast.hex_d_side;
  }
  const bool queued = ModeTraits<MODE>::kFast && acc.q != nullptr;
  if (queued) sums.qn = acc.q->n;   // parked there between passes (ExitQueue)
  // Fresnel split for relative index rr (HitSurface optics.cpp:18-53): reflected / refracted directions and weights
  struct Split {
    float rl[3], rf[3], w_refl, w_refr;
    bool tir;
  };
  auto fresnel = [&](float cos_t, float rr, float rr2, float one_m_rr2, const float4& fn) {
    Split o;
#if HALO_FRESNEL == 0
    const float dd = HALO_FMA(one_m_rr2, fast_rcp(cos_t * cos_t), rr2);
#else
    const float dd = add_rn(fresnel_div(one_m_rr2, cos_t * cos_t), rr2);   // quotient rounded, then the sum: the reference's order (optics.cpp:30)
#endif
    o.tir = dd <= 0.0f;
    const float sq = fresnel_sqrt(fmaxf(dd, 0.0f));
    float Rs = fresnel_div(rr - sq, rr + sq);
    Rs *= Rs;
    float Rp = fresnel_div(HALO_FMA(-rr, sq, 1.0f), HALO_FMA(rr, sq, 1.0f));
    Rp *= Rp;
    o.w_refl = ((Rs + Rp) * 0.5f) * w;
    o.w_refr = w - o.w_refl;
    const float k_refl = 2.0f * cos_t;
    const float k_refr = (rr - sq) * cos_t;
    o.rl[0] = HALO_FMA(-k_refl, fn.x, d[0]), o.rl[1] = HALO_FMA(-k_refl, fn.y, d[1]), o.rl[2] = HALO_FMA(-k_refl, fn.z, d[2]);
    o.rf[0] = HALO_FMA(-k_refr, fn.x, rr * d[0]), o.rf[1] = HALO_FMA(-k_refr, fn.y, rr * d[1]), o.rf[2] = HALO_FMA(-k_refr, fn.z, rr * d[2]);
    return o;
  };
#if HALO_N_VGPR
  float n2_pin = n_idx * n_idx, one_m_n2_pin = 1.0f - n_idx * n_idx;
  asm volatile("" : "+v"(n2_pin), "+v"(one_m_n2_pin));
  const float n2 = n2_pin, one_m_n2 = one_m_n2_pin;
#else
  const float n2 = n_idx * n_idx, one_m_n2 = 1.0f - n_idx * n_idx;
#endif
  if (next != nullptr) next->land();   // the next pass's pool record goes to LDS: behind every load wait of this pass, in front of its first store
  // Legacy-CPU next-face strategy (option rehit_strategy = 0; generic kernels only — see the test on the outgoing child below): a child that
  // "re-hits" is parked here with its path and walked after the ray's main path, by re-entering the loop at its interaction index.
  constexpr bool kLegacyCapable = !ModeTraits<MODE>::kFast;
  constexpr int kParkCap = kLegacyCapable ? 2 : 1;
  struct Parked {
    float d[3], p[3], w;
    int face;
    uint32_t i;
    PathView pv;
  };
  Parked parked[kParkCap];
  uint8_t parked_path[kParkCap][ModeTraits<MODE>::kTables ? kFilterPathCap : 1];
  int n_parked = 0;
  uint32_t i_start = 0u;
walk_path:
  for (uint32_t i = i_start; i < P.max_hits; ++i) {
    // --- Fresnel split at `face` ---
    const float4 fn = *reinterpret_cast<const float4*>(sh->face[face]);
    const float cos_t = dot3_fma(d, fn.x, fn.y, fn.z);
    // On a convex body exactly one child stays inside: the refracted one when entering (cos < 0), the reflected one otherwise; the other
    // child leaves through `face` and is the outgoing candidate.  A ray ENTERS at most once, at its first interaction: every later face
    // was picked by the search below with n.d > eps, i.e. it is hit from inside.  So only the first turn of the loop (a wave-uniform
    // branch) carries the selects between the two cases; the others know the refracted child leaves, the reflected one stays and the
    // relative index is n — eight v_cndmask (1.6x the cost of an FMA on this part, tools/valu_rate_bench) and the rr arithmetic less.
    float ex[3], in[3], ex_w, in_w;
    uint32_t ex_seq, in_seq;
    bool has_exit;
    if (i == 0u) {
      const bool entering = cos_t < 0.0f;
      const float rr = cos_t > 0.0f ? n_idx : inv_n;
      const Split sp = fresnel(cos_t, rr, rr * rr, 1.0f - rr * rr, fn);
#pragma unroll
      for (int a = 0; a < 3; a++) {
        ex[a] = entering ? sp.rl[a] : sp.rf[a];
        in[a] = entering ? sp.rf[a] : sp.rl[a];
      }
      ex_w = entering ? sp.w_refl : sp.w_refr;
      in_w = entering ? sp.w_refr : sp.w_refl;
      ex_seq = entering ? 0u : 1u;
      in_seq = entering ? 1u : 0u;
      has_exit = entering || !sp.tir;
    } else {
      const Split sp = fresnel(cos_t, n_idx, n2, one_m_n2, fn);
#pragma unroll
      for (int a = 0; a < 3; a++) {
        ex[a] = sp.rf[a];
        in[a] = sp.rl[a];
      }
      ex_w = sp.w_refr;
      in_w = sp.w_refl;
      ex_seq = 2u * i + 1u;
      in_seq = 2u * i;
      has_exit = !sp.tir;
    }
    PROBE_MARK(pr, kPhFresnel);
    if constexpr (kLegacyCapable) {
      // The reference excludes the face a ray stands on in two ways (traversal_shared.h:23-29).  Its CUDA backend — and this engine by default —
      // emits the child that leaves through `face` where it is.  Its CPU path (SURVEY's stated ground truth) propagates that child like any
      // other ray: PropagateSlab over ALL faces, the source face included, accepted above +eps when the nearest face is the source and above
      // -eps otherwise (optics.cpp:116-155).  On a convex body with the entry point on its face's plane nothing is ever accepted and the two
      // agree; where a fan's corners were moved off the plane by the vertex merge (or a point sits within rounding of an edge) the CPU path
      // lets the child "re-hit" and go on as a segment.  Option rehit_strategy = 0 does the same here: the child is parked and walked after
      // the main path (the inner child's search is the same under both strategies: its source face has n.d < 0 and is never a candidate).
      if (P.rehit_legacy != 0u && !done && has_exit) {
        float t_far = 1e30f;
        int far = -1;
        for (int fi = 0; fi < face_cnt; fi++) {
          const float4 g = *reinterpret_cast<const float4*>(sh->face[fi]);
          const float den = dot3_fma(ex, g.x, g.y, g.z);
          if (den > kSlabEps) {
            const float t = -(dot3_fma(p, g.x, g.y, g.z) + g.w) / den;   // lm_traversal::SlabFaceT
            if (t < t_far) {
              t_far = t;
              far = fi;
            }
          }
        }
        const bool rehit = far >= 0 && t_far > ((far != face) ? -kSlabEps : kSlabEps);
        if (rehit) has_exit = false;   // it did not leave: the CPU path carries it on as a segment (which a spent hit budget then never traces)
        if (rehit && i + 1u < P.max_hits && n_parked < kParkCap) {
          Parked& k = parked[n_parked];
          for (int a = 0; a < 3; a++) {
            k.d[a] = ex[a];
            k.p[a] = HALO_FMA(t_far, ex[a], p[a]);
          }
          k.w = ex_w;
          k.face = far;
          k.i = i + 1u;
          k.pv = pv;
          if constexpr (MODE != kModePlain) {   // FillRayOtherInfo (simulator.cpp:645): the segment's path takes the face it lands on
            const uint8_t fnum = sh->face_number[far];
            if constexpr (ModeTraits<MODE>::kTables) {
              for (uint32_t b = 0u; b < kFilterPathCap; b++) parked_path[n_parked][b] = path[b];
              if (k.pv.len >= 16u && k.pv.len < kFilterPathCap) parked_path[n_parked][k.pv.len] = fnum;
            }
            if (k.pv.len < 16u) {
              k.pv.reg = pk_shl8(k.pv.reg);
              k.pv.reg.lo |= fnum;
            }
            k.pv.len++;
          }
          n_parked++;
        }
      }
    }
    // every lane that entered the loop is still here at every emit (the exit queue's push is a wave-wide step): `live` says who has a candidate
    emit_gate<MODE, MONO, SMALLC>(P, acc, filter, color, carried, gate, R, !done && has_exit, ex[0], ex[1], ex[2], ex_w, cmf_x, cmf_y, cmf_z, wl_idx, P.ci_start + tid, ex_seq, pv,
                                  i + 1u, sums, pr);
    PROBE_MARK(pr, kPhEmitGate);
    if (i + 1u == P.max_hits) break;
    if (!queued && done) break;
    const uint32_t inward_seq = in_seq;
    d[0] = in[0];
    d[1] = in[1];
    d[2] = in[2];
    w = in_w;
    // --- next face on the convex body (PropagateSlab optics.cpp:64-158) ---
    // min over candidate faces of t = num/den (den > eps > 0) without dividing: num_i/den_i < num_b/den_b
    // <=> num_i*den_b < num_b*den_i.  One reciprocal at the end.
    float num_b = 1e30f, den_b = 1.0f;
    int hit = -1;
    bool none_ahead;
    if constexpr (HEX) {
      // Regular hexagonal prism: slab k has normal (0,0,1), (1,0,0), (1/2, s60, 0), (-1/2, s60, 0) — the builder's exact table
      // values, here literals — and face ids (0,1), (2,5), (3,6), (4,7); both faces of a slab have the same plane constant.  The
      // generic search below with its table reads gone and the zero / unit components folded (x * 1 + 0 == x exactly): same
      // candidates, same order, same comparisons.
      // Round 5: no select on the travel direction.  The face ahead in slab k is +n when n.d > 0 and -n otherwise, and its numerator is
      // -(n.p + d_k) resp. (n.p - d_k) = -((-n.p) + d_k): both are (-d_k) - t with t = n.p carrying n.d's sign — one three-input bit
      // operation and a subtraction where there were a compare, an add, a subtract and a select — and the face id is decoded once after
      // the loop from the winner's code (slab index | sign bit of its n.d) instead of being selected per slab.  Same candidates, same
      // order, the same products and comparisons: the values are bit for bit the old ones (negation is exact).
      constexpr float kS60 = 0.86602540378443864676f;
      const float ndb = -hex_d_basal, nds = -hex_d_side;
      const float hd = d[0] * 0.5f, hp = p[0] * 0.5f;
      uint32_t cb = 0xFFFFFFFFu;   // winner's code; all ones = none
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // (n.d, n.p) of slab k.  The two oblique slabs are four scalar fmas on two shared half-products (x * -0.5 == -(x * 0.5) exactly).  Until
        // the end of round 6 these were packed pairs (v_pk_mul / v_pk_fma on {d, p}): a packed fp32 instruction costs 1.6 plain ones on this part
        // AND its operands must first be moved into register pairs — 4 packed + 4 moves against 6 plain instructions per interaction, same values
        // (configs[1] 1.661 -> 1.626 ms per launch).
        const float nd = (k == 0) ? d[2] : (k == 1) ? d[0] : HALO_FMA(d[1], kS60, k == 2 ? hd : -hd);
        const float np = (k == 0) ? p[2] : (k == 1) ? p[0] : HALO_FMA(p[1], kS60, k == 2 ? hp : -hp);
        const uint32_t sgn = __float_as_uint(nd) & 0x80000000u;
        const float t = __uint_as_float(__float_as_uint(np) ^ sgn);
        const float num = ((k == 0) ? ndb : nds) - t;
        const float den = fabsf(nd);
        const bool better = (den > kSlabEps) && (num * den_b < num_b * den);
        num_b = better ? num : num_b;
        den_b = better ? den : den_b;
        // code = sign of n.d | id of the +n face << 2 | (id of the -n face - id of the +n face): faces (0,1), (2,5), (3,6), (4,7)
        const uint32_t kCode = (k == 0) ? ((0u << 2) | 1u) : ((static_cast<uint32_t>(k + 1) << 2) | 3u);
        cb = better ? (sgn | kCode) : cb;
      }
      none_ahead = cb == 0xFFFFFFFFu;
      hit = static_cast<int>(((cb >> 2) & 7u) + (cb >> 31) * (cb & 3u));   // (meaningless when none_ahead: the lane strays and never reads it)
    } else if (slot_fast != nullptr && slot_hexn) {
      // A sampled FULL prism with the regular prism's normals (SlotFast::hexn): the search above with a plane constant per FACE — the one of the
      // face ahead is picked by the sign of n.d (one select more per slab than the regular prism's, whose two faces share theirs).  Same
      // candidates, order, products and comparisons as the table-driven search below: x * 1 + 0 and 0 * z + acc are exact.
      constexpr float kS60 = 0.86602540378443864676f;
      const float4 dp4 = *reinterpret_cast<const float4*>(slot_fast->d_plus), dm4 = *reinterpret_cast<const float4*>(slot_fast->d_minus);
      const float hd = d[0] * 0.5f, hp = p[0] * 0.5f;
      uint32_t cb = 0xFFFFFFFFu;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float nd = (k == 0) ? d[2] : (k == 1) ? d[0] : HALO_FMA(d[1], kS60, k == 2 ? hd : -hd);   // (scalar, like the regular prism's search above)
        const float np = (k == 0) ? p[2] : (k == 1) ? p[0] : HALO_FMA(p[1], kS60, k == 2 ? hp : -hp);
        const float dpk = (k == 0) ? dp4.x : (k == 1) ? dp4.y : (k == 2) ? dp4.z : dp4.w, dmk = (k == 0) ? dm4.x : (k == 1) ? dm4.y : (k == 2) ? dm4.z : dm4.w;
        const bool pos = nd > 0.0f;
        const float den = fabsf(nd);
        const float num = pos ? -(np + dpk) : (np - dmk);   // (spelled like the table-driven search: the same roundings)
        const bool better = (den > kSlabEps) && (num * den_b < num_b * den);
        num_b = better ? num : num_b;
        den_b = better ? den : den_b;
        const uint32_t kCode = (k == 0) ? ((0u << 2) | 1u) : ((static_cast<uint32_t>(k + 1) << 2) | 3u);
        cb = better ? ((pos ? 0u : 0x80000000u) | kCode) : cb;
      }
      none_ahead = cb == 0xFFFFFFFFu;
      hit = static_cast<int>(((cb >> 2) & 7u) + (cb >> 31) * (cb & 3u));
    } else {
    // (n.d, n.p) as one packed pair per plane: v_pk_mul/v_pk_fma, each element an ordinary fma chain (the pairs are assembled once and serve
    // every plane of the body: 5-20 planes here, where the two literal searches above have two)
    const float2v X = {d[0], p[0]}, Y = {d[1], p[1]}, Z = {d[2], p[2]};
    const int slab_cnt = sh->slab_cnt, single_cnt = sh->single_cnt;
    for (int k = 0; k < slab_cnt; ++k) {
      // opposite faces +n / -n: den(-n) = -den(+n) and n.p flips sign, so only the face the ray travels towards can be ahead
      const float4 g = *reinterpret_cast<const float4*>(sh->slab[k]);
      const float4 e = *reinterpret_cast<const float4*>(sh->slab[k] + 4);
      const float2v r = pk_dot(X, Y, Z, g.x, g.y, g.z);
      const bool pos = r.x > 0.0f;
      const float den = fabsf(r.x);
      const float num = pos ? -(r.y + g.w) : (r.y - e.x);
      const int fi = pos ? __float_as_int(e.y) : __float_as_int(e.z);
      // the face the ray stands on is never ahead: the child that stays inside has n.d < 0 there (it was built from the
      // face normal a few lines up), so the `den > eps` gate already excludes it — no index compare needed
      const bool better = (den > kSlabEps) && (num * den_b < num_b * den);
      num_b = better ? num : num_b;
      den_b = better ? den : den_b;
      hit = better ? fi : hit;
    }
    for (int k = 0; k < single_cnt; ++k) {
      const int fi = sh->single[k];
      const float4 g = *reinterpret_cast<const float4*>(sh->face[fi]);
      const float2v r = pk_dot(X, Y, Z, g.x, g.y, g.z);
      const float den = r.x;
      const float num = -(r.y + g.w);
      const bool better = (den > kSlabEps) && (num * den_b < num_b * den);
      num_b = better ? num : num_b;
      den_b = better ? den : den_b;
      hit = better ? fi : hit;
    }
    }
    if constexpr (!HEX) {
      if (!(slot_fast != nullptr && slot_hexn)) none_ahead = hit < 0;
    }
    const float t_best = num_b * fast_rcp(den_b);
    // No face ahead (numerical edge): legacy treats the child as an outgoing candidate (simulator.cpp:678).  Rare — a wave holds such a lane
    // once in thousands of passes — so it does not ride through the loop's emit site as a special case (selects on every operand of every
    // emit): the wave branches here when it has one, and the lane's exit goes out at once through the plain emit (filter, gate,
    // projection and accumulation at the site, no queue), with the path it has recorded so far.
    const bool stray = !done && (none_ahead || t_best <= -kSlabEps);
    if (__ballot(stray) != 0ull) {
      AccCtx<MONO, SMALLC> direct = acc;
      direct.q = nullptr;
      emit_gate<MODE, MONO, SMALLC>(P, direct, filter, color, carried, gate, R, stray, d[0], d[1], d[2], w, cmf_x, cmf_y, cmf_z, wl_idx, P.ci_start + tid, inward_seq, pv, i + 1u,
                                    sums, pr);
      done = done || stray;
    }
    if (!done) {
      p[0] = HALO_FMA(t_best, d[0], p[0]);
      p[1] = HALO_FMA(t_best, d[1], p[1]);
      p[2] = HALO_FMA(t_best, d[2], p[2]);
      face = hit;
      if constexpr (ModeTraits<MODE>::kFastPath) {   // (max_hits <= 16: the register holds every path; its length is the loop counter)
        uint32_t fn;
        if constexpr (HEX) fn = static_cast<uint32_t>(face) + 1u;   // regular prism: numbers 1..8 in face order (BuildEntryFast checks)
        else fn = sh->face_number[face];
        pv.reg = pk_shl8(pv.reg);
        pv.reg.lo |= fn;
      } else if constexpr (MODE != kModePlain && !HEX) {
        const uint8_t fn = sh->face_number[face];
        if (pv.len < 16u) {
          pv.reg = pk_shl8(pv.reg);
          pv.reg.lo |= fn;
        } else if (pv.len < kFilterPathCap) {
          path[pv.len] = fn;
        }
        pv.len++;
      }
    }
    PROBE_MARK(pr, kPhSlab);
  }
  if constexpr (kLegacyCapable) {
    if (n_parked > 0) {   // the parked children of this ray, last in first out: each re-enters the loop at its own interaction index
      const Parked& k = parked[--n_parked];
      for (int a = 0; a < 3; a++) {
        d[a] = k.d[a];
        p[a] = k.p[a];
      }
      w = k.w;
      face = k.face;
      pv.len = k.pv.len;
      pv.reg = k.pv.reg;
      if constexpr (ModeTraits<MODE>::kTables)
        for (uint32_t b = 0u; b < kFilterPathCap; b++) path[b] = parked_path[n_parked][b];
      done = false;
      i_start = k.i;
      goto walk_path;
    }
  }
  if (next != nullptr) next->late();
  if (queued) {   // park the count: the lanes here agree on it, the first of them writes
    const uint64_t m = __ballot(1);
    if (__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u)) == 0u) acc.q->n = sums.qn;
  }
}

// Bin the staged hits by image tile and append them to the tiles' lists (all kBlock threads call this together).
// Three sweeps over the staged hits: count per tile; reserve each tile's segment with ONE returning global atomic; place
// (the position inside the segment comes from a second LDS counter, so nothing has to stay in registers across barriers).
HD void bin_flush(const DispatchParams& P, HitBuffer& hb) {
  const uint32_t n = min(hb.n, static_cast<uint32_t>(kHitBuf));
  const uint32_t tmask = P.bin_tiles - 1u;   // one level: list = low slot bits (the column hash balances the lists); two levels: a contiguous slot range
  for (uint32_t t = threadIdx.x; t < P.bin_tiles; t += kBlock) hb.cur[t] = 0u;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += kBlock) atomicAdd(&hb.cur[(hb.h[i].x >> P.bin_shift) & tmask], 1u);
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < P.bin_tiles; t += kBlock) {
    const uint32_t c = hb.cur[t];
    hb.cur[t] = c ? atomicAdd(&P.bin_cnt[t * kBinCntStride], c) : 0u;   // count -> start of the reserved segment
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += kBlock) {
    const uint2 h = hb.h[i];
    const uint32_t tile = (h.x >> P.bin_shift) & tmask;
    const uint32_t idx = atomicAdd(&hb.cur[tile], 1u);
    if (idx < P.bin_cap) reinterpret_cast<uint2*>(P.bin_list)[static_cast<size_t>(tile) * P.bin_cap + idx] = h;
    else overflow_add(P, h.x, __uint_as_float(h.y));   // list full (copy 0)
  }
  __syncthreads();
  if (threadIdx.x == 0) hb.n = 0u;
  __syncthreads();
}

// Flush every `every` passes of the ray loop; the interval follows the fill level so that a flush finds the buffer about
// half full (a pass that overruns the buffer falls back to direct atomics, so a late flush costs speed, not hits).  Every
// thread of the workgroup reads the same fill count after the barrier, so the new interval is workgroup-uniform.
HD uint32_t bin_flush_adaptive(const DispatchParams& P, HitBuffer& hb, uint32_t every, uint32_t& since) {
  __syncthreads();
  const uint32_t n = hb.n;
  since = 0u;
  bin_flush(P, hb);
  if (n > static_cast<uint32_t>(kHitBuf) * 5u / 8u) return every > 1u ? every - 1u : 1u;
  if (n < static_cast<uint32_t>(kHitBuf) * 3u / 8u) return every < 64u ? every + 1u : every;
  return every;
}

HD float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return v;
}

// waves per SIMD the register allocator must leave room for: 5 for the production kernels (96 VGPRs, measured 4.5 % faster
// than 4); the filter / capture kernels carry the path register, the predicate tables' addressing and the colour mask on
// top, and are built for HALO_MIN_WAVES_FILTER (measured below)
#ifndef HALO_MIN_WAVES
#define HALO_MIN_WAVES 5
#endif
#ifndef HALO_FILTER_SIX
#define HALO_FILTER_SIX 1   // the logging filter kernels of one regular prism take the small cache and six waves too (round 4: +8 .. +14 %)
#endif
#ifndef HALO_POOL_DB48
#define HALO_POOL_DB48 1   // ... general pool shapes (pyramids) too (NextShape48)
#endif
#ifndef HALO_POOL_DB
#define HALO_POOL_DB 1   // prism pools under the hit log: the next pass's record requested a pass ahead (NextShape)
#endif
#ifndef HALO_POOL_WAVES
#define HALO_POOL_WAVES 4   // the logging shape-pool kernels
#endif
#ifndef HALO_LOG_WAVES
#define HALO_LOG_WAVES 6   // the LOGGING plain scalar-plane kernels of one regular prism (see min_waves)
#endif
#ifndef HALO_MIN_WAVES_FILTER
#define HALO_MIN_WAVES_FILTER 3
#endif
// The plain scalar-plane kernels of one REGULAR prism: 81 VGPRs and 31 KB of LDS run five waves per SIMD (round 4: +7 % over four); the
// logging ones of them halve the pixel cache (a miss is an 8-byte log record there, not a memory-side atomic: 1024 slots measure the same
// as 2048 at four and five waves) for 26 KB and SIX waves: configs[1] 20.24 -> 19.60 ms per step.
template <int MODE, int GEOM, bool MONO, int ACC>
constexpr bool small_cache_hex() { return (MODE == kModePlain || (HALO_FILTER_SIX && MODE == kModeFilter)) && GEOM == kGeomOneHex && MONO && (ACC == kAccLog || ACC == kAccLogFinal); }
template <int MODE, int GEOM, bool MONO, int ACC>
constexpr int min_waves() {
  if (!ModeTraits<MODE>::kFast) return HALO_MIN_WAVES_FILTER;
  if ((ACC != kAccDirect && ACC != kAccNone) || ((GEOM == kGeomOne || GEOM == kGeomOneHex) && ACC != kAccNone)) {
    if (small_cache_hex<MODE, GEOM, MONO, ACC>()) return HALO_LOG_WAVES;
    if (MODE == kModePlain && (GEOM == kGeomPoolPrism || GEOM == kGeomPool) && (ACC == kAccLog || ACC == kAccLogFinal)) return HALO_POOL_WAVES;
    return (MODE == kModePlain && GEOM == kGeomOneHex && MONO) ? 5 : 4;
  }
  return MODE == kModePlain ? HALO_MIN_WAVES : 4;
}
// LENS / VIS >= 0, NOGATE: instantiated for that lens, that visible range and prob <= 0 — the projection's dispatch over 11 lens
// types (uniform branches, and the SGPRs their parameters hold), the visibility tests and the gate stream fold away: configs[1]
// 2.96 -> 2.73 (lens) -> 2.60 ms per launch.  Done for the last-layer one-shape scalar kernels and the lenses of the shipped examples.
template <int MODE, int GEOM, bool MONO, int ACC, int LENS = -1, int VIS = -1, bool NOGATE = false>   // ACC: kAccDirect, kAccBin (staged + binned hit lists), kAccLog (per-workgroup hit log), ...
__global__ void __launch_bounds__(kBlock, (min_waves<MODE, GEOM, MONO, ACC>())) halo_trace_kernel(const DispatchParams P) {
  constexpr bool BIN = ACC == kAccBin, LOG = ACC == kAccLog || ACC == kAccLogFinal, NONE = ACC == kAccNone, LAST = ACC == kAccLogFinal;
  static_assert(!LOG || ModeTraits<MODE>::kFast, "the hit log is a production-mode route");
  static_assert(!NONE || (ModeTraits<MODE>::kFast && MONO), "kAccNone: production mode; nothing accumulates, so one (scalar) flavour serves every session");
  static_assert(MODE == kModePlain || MODE == kModeFilter || (LENS < 0 && VIS < 0 && !NOGATE), "lens / visible-range / closed-gate specialisations exist for the plain and the filter kernels");
  static_assert(!BIN || MONO, "binned accumulation is a one-plane mode");
  Probe pr;
#ifdef HALO_PROBE
  for (int k = 0; k < 16; k++) pr.acc[k] = 0u;
  probe_start(pr);
  const uint64_t t_begin = pr.t0;
#endif
  // prism pools under the hit log fetch the next pass's record a pass ahead (NextShape): 304 more bytes per half-wave
  constexpr bool POOLDB = HALO_POOL_DB && (GEOM == kGeomPoolPrism || (HALO_POOL_DB48 && GEOM == kGeomPool)) && LOG;
  typedef typename std::conditional<GEOM == kGeomPoolPrism, NextShape, NextShape48>::type NextT;
  constexpr bool SMALLC = (BIN && GEOM != kGeomOne && GEOM != kGeomOneHex) || small_cache_hex<MODE, GEOM, MONO, ACC>();
  __shared__ __attribute__((aligned(16))) LdsTables<MONO, SMALLC> T;
  __shared__ __attribute__((aligned(16))) HitSlot<BIN> s_hits;
  constexpr bool QUEUE = ModeTraits<MODE>::kFast && ACC != kAccBin && ACC != kAccNone && (GEOM == kGeomOne || GEOM == kGeomOneHex);
  __shared__ __attribute__((aligned(16))) ExitQueues<QUEUE> s_queue;
  __shared__ __attribute__((aligned(16))) ExitQueueMasks<QUEUE && MODE == kModeColor> s_queue_mask;
  __shared__ uint32_t s_fast_ee[ModeTraits<MODE>::kFastPath ? kFastEeLds * 32u : 1u];
  AccCtx<MONO, SMALLC> acc;
  acc.q = nullptr;
  acc.qm = nullptr;
  acc.fast = nullptr;
  acc.fast_ee = nullptr;
  if constexpr (QUEUE) {
    acc.q = &s_queue.q[threadIdx.x >> 6];
    if ((threadIdx.x & 63u) == 0u) acc.q->n = 0u;
    if constexpr (MODE == kModeColor) acc.qm = &s_queue_mask.q[threadIdx.x >> 6];
  }
  if constexpr (ModeTraits<MODE>::kFastPath) {
    acc.fast = (const FastTables __attribute__((address_space(4)))*)(P.fast);
    acc.fast_ee = s_fast_ee;
    for (uint32_t i = threadIdx.x; i < kFastEeLds * 32u; i += kBlock) s_fast_ee[i] = P.fast->ee[i >> 5][i & 31u];
  }
  acc.none = NONE;
  acc.last = LAST;
  acc.lens = LENS;
  acc.vis = VIS;
  acc.nogate = NOGATE;
  acc.cache = &T.cache;
  acc.hits = nullptr;
  acc.log_n = nullptr;
  acc.proj = nullptr;
#if HALO_PROJ_LDS
  __shared__ __attribute__((aligned(16))) ProjLds s_proj;
  // Measured per kernel family (same box, alternating builds): the last-layer plain kernels gain (configs[1] 1.720 -> 1.688 ms per launch, SGPR
  // spills 49 -> 20); the filter-mode queue kernels LOSE (ms_multi_crystal_complex_filter's first layer 2.46 -> 2.64 ms) and so do the pool kernels,
  // which project at the emit site (configs[4] 2.07 -> 2.16 ms, 4d 0.129 -> 0.149): it stays with the kernels it pays for (2 / 3 widen it, A/B only).
  if constexpr ((QUEUE && MODE == kModePlain && LAST) || (HALO_PROJ_LDS == 2 && QUEUE) || (HALO_PROJ_LDS == 3 && ModeTraits<MODE>::kFast)) {   // (visible to the waves behind the barrier that ends the prologue)
    if (threadIdx.x < sizeof(ProjLds) / 4u) reinterpret_cast<uint32_t*>(&s_proj)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&P.proj)[threadIdx.x];   // proj and proj_pre
    acc.proj = &s_proj;
  }
#endif
  __shared__ uint32_t s_log_n;
  if constexpr (LOG) {
    acc.log_n = &s_log_n;
    if (threadIdx.x == 0) s_log_n = 0u;
  }
  if constexpr (BIN) {
    acc.hits = &s_hits.b;
    if (threadIdx.x == 0) s_hits.b.n = 0u;
  }
  __shared__ __attribute__((aligned(16))) FilterSlot<ModeTraits<MODE>::kTables> s_filter;
  constexpr bool POOL = GEOM != kGeomOne && GEOM != kGeomOneHex;
  typedef typename PoolSlotType<GEOM>::type PoolSlot;
  typedef typename PoolSlotType<GEOM>::rec PoolRec;
  __shared__ __attribute__((aligned(16))) PoolSlots<POOL, PoolSlot> s_pool;       // stochastic: one shape per half-wave
  __shared__ __attribute__((aligned(16))) SlotFast s_slot_fast[(POOLDB && GEOM == kGeomPoolPrism) ? kBlock / 32 : 1];
  constexpr uint32_t kWlLds = 64u;   // the reference's default illuminant pool (BASELINE configs[4]: 31)
  __shared__ __attribute__((aligned(16))) WlEntryDev s_wl[POOLDB ? kWlLds : 1u];
  __shared__ __attribute__((aligned(16))) f4v s_pool_mirror[POOLDB ? (kBlock / 32) * NextT::kMirrorRows : 1];   // ... and the rows of its next one that cannot land in the slot yet (NextShape)
  constexpr bool HEXK = GEOM == kGeomOneHex;
  typedef typename std::conditional<HEXK, ShapeHead, ShapeDev>::type OneShape;   // a regular prism's kernels stage the 464-byte prefix they read (halo_device.h ShapeHead)
  __shared__ __attribute__((aligned(16))) PoolSlots<!POOL, OneShape, 1> s_shape;  // deterministic: the dispatch's one shape
  __shared__ __attribute__((aligned(16))) ColorSlot<ModeTraits<MODE>::kTables> s_color;
  const ColorDev* color = nullptr;
  if (ModeTraits<MODE>::kTables && P.color != nullptr) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(P.color);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&s_color);
    for (uint32_t i = threadIdx.x; i < sizeof(ColorDev) / 4u; i += kBlock) dst[i] = src[i];
    color = reinterpret_cast<const ColorDev*>(&s_color);
  }
  const FilterDev* filter = nullptr;
  if (ModeTraits<MODE>::kTables && P.filter != nullptr) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(P.filter);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&s_filter);
    for (uint32_t i = threadIdx.x; i < sizeof(FilterDev) / 4u; i += kBlock) dst[i] = src[i];
    filter = reinterpret_cast<const FilterDev*>(&s_filter);
  }
  if (!NONE && (P.aggregate == 1u || P.aggregate == 3u)) {
    for (int i = threadIdx.x; i < CacheGeom<MONO, SMALLC>::kN; i += kBlock) T.cache.tag[i] = 0u;
    for (int i = threadIdx.x; i < CacheGeom<MONO, SMALLC>::kN * (MONO ? 1 : 3); i += kBlock) T.cache.val[i] = 0.0f;
  }
  // ---- stage the dispatch-constant tables into LDS ----
  if (P.lat_path == kLatLut)
    for (int i = threadIdx.x; i < 3 * kLutNodes; i += kBlock) T.lut[i] = P.lut[i];
  if constexpr (POOLDB) {
    static_assert(sizeof(WlEntryDev) == 32u, "copied as two float4");
    if (P.wl_pool_size <= kWlLds)
      for (uint32_t i = threadIdx.x; i < 2u * P.wl_pool_size; i += kBlock) reinterpret_cast<f4v*>(s_wl)[i] = reinterpret_cast<const f4v*>(P.wl_pool)[i];
  }
  if (P.source == kSrcTransit)
    for (int i = threadIdx.x; i <= kContShards; i += kBlock) T.seg[i] = P.cont_in_seg[i];
  if (threadIdx.x == 0) T.fidx.ok = 0u;
  if (!POOL && P.entry_fast != nullptr) {
    const float4* src = reinterpret_cast<const float4*>(P.entry_fast);
    float4* dst = reinterpret_cast<float4*>(&T.efast);
    for (uint32_t i = threadIdx.x; i < sizeof(EntryFastDev) / 16u; i += kBlock) dst[i] = src[i];
  }
  if constexpr (!POOL) {
    const float4* src = reinterpret_cast<const float4*>(P.shapes);
    float4* dst = reinterpret_cast<float4*>(&s_shape.s[0]);
    for (uint32_t i = threadIdx.x; i < sizeof(OneShape) / 16u; i += kBlock) dst[i] = src[i];
  }
  if constexpr (!POOL && !HEXK) {   // (a regular prism's entry pick goes by EntryFastDev alone: no per-face view needed)
    __syncthreads();
    // per-face view of the fan table (FaceIndex): thread f sums face f's triangles; the grouping is verified, not assumed
    const ShapeDev& S0 = s_shape.s[0];
    const int fc = S0.face_cnt, tc = S0.tri_cnt;
    if (static_cast<int>(threadIdx.x) < fc) {
      const int f = static_cast<int>(threadIdx.x);
      float a = 0.0f;
      int first = tc, cnt = 0;
      for (int t = 0; t < tc; ++t)
        if (S0.tri_face[t] == f) {
          a += S0.tri_na[t][3];
          first = min(first, t);
          cnt++;
        }
      T.fidx.area[f] = a;
      T.fidx.tri0[f] = static_cast<uint8_t>(first);
      T.fidx.tric[f] = static_cast<uint8_t>(cnt);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      bool ok = fc > 0 && fc <= kMaxFaces && tc > 0;
      int next = 0;
      for (int f = 0; f < fc && ok; ++f) {
        ok = T.fidx.tric[f] > 0 && T.fidx.tri0[f] == next;
        for (int k = 0; k < T.fidx.tric[f] && ok; ++k) ok = S0.tri_face[next + k] == f;
        next += T.fidx.tric[f];
      }
      T.fidx.ok = (ok && next == tc) ? 1u : 0u;
    }
  }
  __syncthreads();

  RaySums sums = {0.0f, 0.0f, 0u, 0u, 0u};
  Wl0 wl0;
  wl0.e = P.wl_pool[0];
  wl0.inv_n = 1.0f / wl0.e.n_idx;
  PROBE_MARK(pr, kPhPrologue);
  const uint32_t stride = gridDim.x * kBlock;
  uint32_t flush_every = 1u, since_flush = 0u;  // binned mode: passes between workgroup-wide flushes (adaptive, uniform)
  bool staged = false;
  if constexpr (POOL) {
   if ((P.geom_clock & 31u) == 0u) {
    staged = true;
    // stochastic geometry: geom_clock consecutive rays share one sampled shape (simulator.cpp:1244-1275).  With the
    // clock a multiple of 32 every half-wave traces ONE shape per pass: its 32 lanes copy the rows that shape uses
    // (header, face_cnt plane rows, tri_cnt fan rows — 1.3 KB for a prism) from the pool into the half-wave's LDS slot
    // with coalesced 16-byte loads, and the interaction loop then reads them as LDS broadcasts exactly like the
    // deterministic path.  LDS operations of one wave retire in order, so no barrier is needed around the copy.
    const uint32_t l32 = threadIdx.x & 31u;
    if constexpr (POOLDB) {
      constexpr bool PRISM = GEOM == kGeomPoolPrism;
      static_assert(!PRISM || (std::is_same<PoolRec, ShapePrism>::value && std::is_same<PoolSlot, ShapePrism>::value), "a prism's slot is its record, copied whole");
      static_assert(PRISM || (std::is_same<PoolRec, ShapeDev>::value && std::is_same<PoolSlot, ShapeSlot48>::value), "NextShape48 maps ShapeDev onto ShapeSlot48");
      PoolSlot* const slot = &s_pool.s[threadIdx.x >> 5];
      f4v* const mirror = &s_pool_mirror[(threadIdx.x >> 5) * NextT::kMirrorRows];
      {
        const uint32_t first0 = blockIdx.x * kBlock + (threadIdx.x & ~31u);
        if (first0 < P.n_rays) stage_shape(slot, reinterpret_cast<const PoolRec*>(P.shapes) + first0 / P.geom_clock, l32);   // the first pass's record: the old way
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
      const WlEntryDev* const wl_lds = P.wl_pool_size <= kWlLds ? s_wl : nullptr;
      bool mirrored = false;   // (half-wave-uniform)
      for (uint32_t base = blockIdx.x * kBlock; base < P.n_rays; base += stride) {
        const uint32_t tid = base + threadIdx.x;
        const uint32_t first = base + (threadIdx.x & ~31u);
        if (mirrored) NextT::take_mirror(reinterpret_cast<f4v*>(slot), mirror, l32);
        asm volatile("" : : : "memory");
        __builtin_amdgcn_wave_barrier();
        SlotFast* sfast = nullptr;
        if constexpr (PRISM) {
          sfast = &s_slot_fast[threadIdx.x >> 5];
          if (first < P.n_rays) {
            if (P.pool_entry_fast != 0u) build_slot_fast(sfast, slot, l32);
            else if (l32 == 0u) sfast->ok = sfast->hexn = 0u;
          }
        }
        // (LDS operations of one wave retire in order: all the copies need is that the compiler keeps them in order — NOT a workgroup-scope
        // fence, which on gfx950 is a wait for every store in flight)
        asm volatile("" : : : "memory");
        __builtin_amdgcn_wave_barrier();
        // (rays of one launch number < 2^28 + a stride: none of these sums wraps)
        const uint32_t next_first = first + stride;
        NextT nx;
        nx.src = next_first < P.n_rays ? reinterpret_cast<const f4v*>(reinterpret_cast<const PoolRec*>(P.shapes) + next_first / P.geom_clock) : nullptr;
        nx.slot = reinterpret_cast<f4v*>(slot);
        nx.mirror = mirror;
        nx.l32 = l32;
        mirrored = nx.src != nullptr;
        PROBE_MARK(pr, kPhStage);
        // (a lane without a ray — the launch's last rays — has no next record either: next_first > tid >= n_rays)
        if (tid < P.n_rays) trace_one<MODE, MONO, SMALLC, false>(P, T, acc, filter, color, static_cast<const PoolSlot*>(slot), wl0, tid, sums, pr, &nx, wl_lds, sfast);
        asm volatile("" : : : "memory");
        __builtin_amdgcn_wave_barrier();
        PROBE_MARK(pr, kPhSlab);
      }
    } else {
    PoolSlot* slot = &s_pool.s[threadIdx.x >> 5];
    for (uint32_t base = blockIdx.x * kBlock; base < P.n_rays; base += stride) {
      const uint32_t tid = base + threadIdx.x;
      const uint32_t first = base + (threadIdx.x & ~31u);
      PROBE_MARK(pr, kPhSlab);   // (loop bookkeeping since the last ray goes with the previous phase)
      if (first < P.n_rays) stage_shape(slot, reinterpret_cast<const PoolRec*>(P.shapes) + first / P.geom_clock, l32);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      PROBE_MARK(pr, kPhStage);
      if (tid < P.n_rays) trace_one<MODE, MONO, SMALLC, GEOM == kGeomOneHex>(P, T, acc, filter, color, static_cast<const PoolSlot*>(slot), wl0, tid, sums, pr);
      __builtin_amdgcn_wave_barrier();
      PROBE_MARK(pr, kPhSlab);
      if constexpr (BIN) {
        if (++since_flush >= flush_every) flush_every = bin_flush_adaptive(P, s_hits.b, flush_every, since_flush);
      }
      PROBE_MARK(pr, kPhFlush);
    }
    }
   }
  }
  if (!staged) {
    for (uint32_t base = blockIdx.x * kBlock; base < P.n_rays; base += stride) {   // workgroup-uniform trip count
      const uint32_t tid = base + threadIdx.x;
      if (tid < P.n_rays) {
        if constexpr (POOL) {  // shape clock not a multiple of 32: lanes of a half-wave may differ, read the pool through L1/L2
          const PoolRec* sh = reinterpret_cast<const PoolRec*>(P.shapes) + (tid / P.geom_clock);
          trace_one<MODE, MONO, SMALLC, GEOM == kGeomOneHex>(P, T, acc, filter, color, sh, wl0, tid, sums, pr);
        } else {
          const OneShape* sh = &s_shape.s[0];  // LDS: ds_read_b128 broadcasts
          trace_one<MODE, MONO, SMALLC, GEOM == kGeomOneHex>(P, T, acc, filter, color, sh, wl0, tid, sums, pr);
        }
      }
      if constexpr (BIN) {
        if (++since_flush >= flush_every) flush_every = bin_flush_adaptive(P, s_hits.b, flush_every, since_flush);
      }
    }
  }
  if constexpr (QUEUE) {   // what the last passes left on the wave's exit queue
#ifdef HALO_PROBE
    probe_start(pr);
#endif
    sums.qn = acc.q->n;
    drain_exits<MODE, MONO, SMALLC>(P, acc, sums, true, pr);
    PROBE_MARK(pr, kPhFinalDrain);
  }
  if constexpr (BIN) {
    __syncthreads();
    bin_flush(P, s_hits.b);
  }
  // ---- flush the workgroup's pixel cache: one global atomic per claimed slot and channel ----
#ifdef HALO_PROBE
  probe_start(pr);
#endif
  if (!NONE && (P.aggregate == 1u || P.aggregate == 3u)) {
    __syncthreads();
    for (int i = threadIdx.x; i < CacheGeom<MONO, SMALLC>::kN; i += kBlock) {
      const uint32_t key = T.cache.tag[i];
      if (key == 0u) continue;
      // (a plane index rides above bit 23 only in sessions with a plane per pool entry, which stop at 2^23 pixels: halo_begin)
      const uint32_t pl = (MONO && P.mono_by_wl) ? (key - 1u) >> 23 : 0u, pix = (MONO && P.mono_by_wl) ? ((key - 1u) & 0x7FFFFFu) : key - 1u;
      if (MONO) {
        const float v = T.cache.val[i];
        if (v == 0.0f) continue;
        if (LOG) log_hit<false>(P, &s_log_n, log_slot(P, pl, pix), v);
        else atomic_add_f32(mono_slot(P, pl, pix), v);
      } else if (LOG) {   // a cached pixel leaves as three records whose weights ARE X, Y, Z (codes pool size + channel)
        for (uint32_t c = 0; c < 3u; ++c) {
          const float v = T.cache.val[i * 3 + c];
          if (v != 0.0f) log_hit<true>(P, &s_log_n, log_slot_xyz(P, P.wl_pool_size + c, pix), v);
        }
      } else {
        atomic_add_f32(mono_slot(P, 0u, pix), T.cache.val[i * 3 + 0]);
        atomic_add_f32(mono_slot(P, 1u, pix), T.cache.val[i * 3 + 1]);
        atomic_add_f32(mono_slot(P, 2u, pix), T.cache.val[i * 3 + 2]);
      }
    }
  }
  if constexpr (LOG) {   // the region's fill count, for the split pass
    __syncthreads();
    if (threadIdx.x == 0) P.bin_cnt[blockIdx.x] = min(s_log_n, P.bin_cap);
  }
  // ---- the scalar tallies: per-wave reduction, the four waves' sums joined in LDS, then ONE set of fp64 atomics per workgroup onto
  // one of kTallyLines cache lines.  (Round 5: four atomics per WAVE onto one line were 4 x 1024 same-line atomics for a 256-workgroup
  // launch — they serialise memory-side at ~12 ns each, 37 us behind a kernel whose waves were gone after 18: the fixed cost of every
  // small session, tools/launch_ramp_bench.hip + tools/phase_probe.py tiny.)  The tallies are CUMULATIVE: nobody zeroes them, the host
  // reads differences (halo_backend.cpp pull_tally).
  __shared__ float s_tally[kBlock / 64][4];
  {
    const float landed = wave_sum(sums.landed);
    const float exit_w = wave_sum(sums.exit_w);
    const float exit_n = wave_sum(static_cast<float>(sums.exit_n));
    const float pix_n = wave_sum(static_cast<float>(sums.pix_n));
    if ((threadIdx.x & 63) == 0) {
      float* t = s_tally[threadIdx.x >> 6];
      t[kSumLanded] = landed, t[kSumExitW] = exit_w, t[kSumExitN] = exit_n, t[kSumPixN] = pix_n;
    }
    __syncthreads();
    if (threadIdx.x < 4u) {   // lane k sums tally k over the waves (exit and pixel counts: < 2^24 per wave and launch pass, exact in fp32)
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) v += static_cast<double>(s_tally[w][threadIdx.x]);
      if (v != 0.0) atomicAdd(&P.tally[(blockIdx.x & (kTallyLines - 1u)) * kTallyStride + threadIdx.x], v);
    }
  }
#ifdef HALO_PROBE
  PROBE_MARK(pr, kPhKernelFixed);
  pr.acc[kPhTotal] = static_cast<uint32_t>(pr.t0 - t_begin);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 16; k++) atomicAdd(&g_halo_probe[k], static_cast<unsigned long long>(pr.acc[k]));
#endif
}

// host-callable launcher pieces: each halo_trace_m<MODE>.hip translation unit instantiates the kernels of one MODE
// (the instantiations of the three MODEs compile in parallel that way; halo_backend.cpp is plain C++ and never sees <<<>>>)
// The specialised instantiations of the last-layer kernels: lens, visible range and "prob <= 0" (the usual last layer: nothing
// passes the gate) as template constants.  Lenses of the reference's shipped examples (config_example.json: linear, dual fisheye
// equal area; the BASELINE configurations: fisheye equal area; bench_config_stoch.json: rectangular) x visible upper / full;
// anything else runs the generic kernel.
// (NOGATE = true: the last layer with prob <= 0.  NOGATE = false, round 6: the logging kernels of the layers BEFORE the last — their gate is open, the lens
//  and the visible range are as constant as in the last layer, and the generic-lens form of them carried every lens's code: 8723 instructions, 120
//  spilled SGPRs, 10 spilled VGPRs and scratch against 4612 / 45 / 0 / none for the dual-fisheye instantiation, tools/isa_by_line.py.)
template <int MODE, int GEOM, bool MONO, int ACC, int LENS, bool NOGATE>
static void launch_vis(const DispatchParams& P, dim3 grid, dim3 block, hipStream_t stream) {
  if (P.proj.visible_range == HALO_VISIBLE_UPPER) hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, MONO, ACC, LENS, HALO_VISIBLE_UPPER, NOGATE>), grid, block, 0, stream, P);
  else if (P.proj.visible_range == HALO_VISIBLE_FULL) hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, MONO, ACC, LENS, HALO_VISIBLE_FULL, NOGATE>), grid, block, 0, stream, P);
  else hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, MONO, ACC>), grid, block, 0, stream, P);
}
template <int MODE, int GEOM, bool MONO, int ACC, bool NOGATE = true>
static void launch_lens(const DispatchParams& P, dim3 grid, dim3 block, hipStream_t stream) {
  if (NOGATE && P.prob > 0.0f) {
    hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, MONO, ACC>), grid, block, 0, stream, P);
    return;
  }
  switch (P.proj.proj_type) {
    case HALO_LENS_LINEAR: launch_vis<MODE, GEOM, MONO, ACC, HALO_LENS_LINEAR, NOGATE>(P, grid, block, stream); break;
    case HALO_LENS_FISHEYE_EQUAL_AREA: launch_vis<MODE, GEOM, MONO, ACC, HALO_LENS_FISHEYE_EQUAL_AREA, NOGATE>(P, grid, block, stream); break;
    case HALO_LENS_DUAL_FISHEYE_EQUAL_AREA: launch_vis<MODE, GEOM, MONO, ACC, HALO_LENS_DUAL_FISHEYE_EQUAL_AREA, NOGATE>(P, grid, block, stream); break;
    case HALO_LENS_RECTANGULAR: launch_vis<MODE, GEOM, MONO, ACC, HALO_LENS_RECTANGULAR, NOGATE>(P, grid, block, stream); break;
    default: hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, MONO, ACC>), grid, block, 0, stream, P); break;
  }
}

template <int MODE, int GEOM>
static void launch_mono(const DispatchParams& P, dim3 grid, dim3 block, hipStream_t stream, bool mono) {
  if constexpr (ModeTraits<MODE>::kFast && (GEOM == kGeomOne || GEOM == kGeomOneHex)) {
    if (P.no_land != 0u) {   // every exit of this layer continues: no cache, no queue, no accumulation code
      hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, true, kAccNone>), grid, block, 0, stream, P);
      return;
    }
  }
  if constexpr (ModeTraits<MODE>::kFast) {   // the hit log exists for the production-shaped kernels: scalar planes, or X/Y/Z planes of an illuminant session
    if (P.bin_log != 0u) {
      if constexpr (GEOM == kGeomOne || GEOM == kGeomOneHex) {
        if (mono && P.final_layer != 0u) {   // the last layer's one-shape scalar kernels carry no continuation-append code
          if constexpr (MODE == kModePlain || MODE == kModeFilter) launch_lens<MODE, GEOM, true, kAccLogFinal>(P, grid, block, stream);
          else hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, true, kAccLogFinal>), grid, block, 0, stream, P);
          return;
        }
      }
      // (round 6, end: the reference's own stochastic benchmark — bench_config_stoch.json: rectangular lens, full sky, closed gate — as constants for
      //  the shape-pool kernels of both plane layouts.  The pool kernels project at the emit site, and the generic form of them holds every lens's
      //  code inside the interaction loop: 14 283 instructions, 157 spilled SGPRs, 3 spilled VGPRs and scratch against 9950 / 73 / 0 / none.)
      if constexpr (MODE == kModePlain && (GEOM == kGeomPool || GEOM == kGeomPoolPrism)) {
        if (P.prob <= 0.0f && P.proj.visible_range == HALO_VISIBLE_FULL && P.proj.proj_type == HALO_LENS_RECTANGULAR) {
          if (mono) hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, true, kAccLog, HALO_LENS_RECTANGULAR, HALO_VISIBLE_FULL, true>), grid, block, 0, stream, P);
          else hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, false, kAccLog, HALO_LENS_RECTANGULAR, HALO_VISIBLE_FULL, true>), grid, block, 0, stream, P);
          return;
        }
      }
      if (mono) {
        if constexpr ((MODE == kModePlain || MODE == kModeFilter) && (GEOM == kGeomOne || GEOM == kGeomOneHex)) launch_lens<MODE, GEOM, true, kAccLog, false>(P, grid, block, stream);
        else hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, true, kAccLog>), grid, block, 0, stream, P);
      } else if constexpr (GEOM == kGeomPool || GEOM == kGeomPoolPrism) {
        // illuminant sessions over sampled crystals: a full-sky render with a closed gate (bench_config_stoch.json's shape) knows both at
        // compile time (configs[4] 5.52 -> 5.43 ms per step; its lens as a constant gains nothing more: 4.31 vs 4.32 ms per launch)
        if constexpr (MODE == kModePlain) {
          if (P.prob <= 0.0f && P.proj.visible_range == HALO_VISIBLE_FULL) {
            hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, false, kAccLog, -1, HALO_VISIBLE_FULL, true>), grid, block, 0, stream, P);
            return;
          }
        }
        hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, false, kAccLog>), grid, block, 0, stream, P);
      } else hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, false, kAccLog>), grid, block, 0, stream, P);
      return;
    }
  }
  if constexpr (GEOM == kGeomOneHex) {   // ... and the regular-prism instantiation for the production mode's direct and logged routes
    if (mono) hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, true, kAccDirect>), grid, block, 0, stream, P);
    else hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, false, kAccDirect>), grid, block, 0, stream, P);
  } else {
    if (mono && P.bin_list != nullptr) hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, true, kAccBin>), grid, block, 0, stream, P);
    else if (mono) hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, true, kAccDirect>), grid, block, 0, stream, P);
    else hipLaunchKernelGGL((halo_trace_kernel<MODE, GEOM, false, kAccDirect>), grid, block, 0, stream, P);
  }
}

template <int MODE>
static hipError_t launch_mode(const DispatchParams& P, int blocks, hipStream_t stream, int geom, bool mono) {
  dim3 grid(blocks), block(kBlock);
  if constexpr (ModeTraits<MODE>::kFast) {
    if (geom == kGeomOneHex && (P.bin_list == nullptr || P.bin_log != 0u || P.no_land != 0u)) {
      launch_mono<MODE, kGeomOneHex>(P, grid, block, stream, mono);
      return hipGetLastError();
    }
  }
  if (geom == kGeomOneHex) geom = kGeomOne;
  if (geom == kGeomPoolPrism) launch_mono<MODE, kGeomPoolPrism>(P, grid, block, stream, mono);
  else if (geom == kGeomPool) launch_mono<MODE, kGeomPool>(P, grid, block, stream, mono);
  else launch_mono<MODE, kGeomOne>(P, grid, block, stream, mono);
  return hipGetLastError();
}

}  // namespace halo
