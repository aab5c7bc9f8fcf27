/*
 * halo_oracle.c — CPU restatement (plain C99) of the Lumice per-ray trace path.
 *
 * TEST INFRASTRUCTURE ONLY (see halo_oracle.h).  Build: oracle/Makefile → oracle/liboracle.so with
 * -O2 -ffp-contract=off so float expressions evaluate exactly as the reference's host build does.
 * All citations are file:line under /root/reference.
 */
#include "halo_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cie_tables_oracle.inc"

/* math.hpp:21-31 */
#define HO_PI_F 3.14159265359f
#define HO_PI_2F (HO_PI_F / 2.0f)
#define HO_FLOAT_EPS 1e-5f
#define HO_DEG2RAD (HO_PI_F / 180.0f)
#define HO_SQRT3_F 1.73205080757f
/* lm_shims.h:84-85 (host branch) */
#define LM_PI_F 3.14159265358979323846f
#define LM_PI_2F 1.5707963267948966f

static const double kPiD = 3.14159265358979323846;
static float clampf(float x, float a, float b) { return fminf(fmaxf(x, a), b); } /* std::clamp on finite input */

/* ======================================================================================== */
/* src/core/shared/pcg_shared.h                                                              */
/* ======================================================================================== */

/* pcg_shared.h:193-197 */
uint32_t ho_pcg_hash(uint32_t x) {
  x = x * 747796405u + 2891336453u;
  x = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
  return (x >> 22u) ^ x;
}

/* pcg_shared.h:199-201 */
float ho_u01_from_hash(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

/* pcg_shared.h:257-261 */
uint32_t ho_pcg_advance_hi(uint32_t base_lo, uint32_t base_hi, uint32_t tid) {
  uint32_t lo = base_lo + tid;
  uint32_t carry = (lo < base_lo) ? 1u : 0u;
  return base_hi + carry;
}

/* pcg_shared.h:263-268 */
uint32_t ho_pcg_seed_with_high(uint32_t seed, uint32_t hi) {
  if (hi == 0u) return seed;
  return seed ^ ho_pcg_hash(hi);
}

/* pcg_shared.h:270-274 */
float ho_pcg_uniform(HoStream* s) {
  uint32_t h = ho_pcg_hash(s->seed ^ ho_pcg_hash(s->global_idx * 1000003u + s->slot));
  s->slot++;
  return ho_u01_from_hash(h);
}

/* pcg_shared.h:277-281 */
float ho_pcg_gaussian(HoStream* s) {
  float u1 = fmaxf(ho_pcg_uniform(s), 1e-7f);
  float u2 = ho_pcg_uniform(s);
  return sqrtf(-2.0f * logf(u1)) * cosf(2.0f * LM_PI_F * u2);
}

/* pcg_shared.h:290-308 */
float ho_pcg_get_dist(HoStream* s, uint32_t dtype, float mean, float std_val) {
  if (dtype == HALO_DIST_NONE) return mean;
  if (dtype == HALO_DIST_UNIFORM) return (ho_pcg_uniform(s) - 0.5f) * std_val + mean;
  if (dtype == HALO_DIST_GAUSS || dtype == HALO_DIST_GAUSS_LEGACY) return ho_pcg_gaussian(s) * std_val + mean;
  if (dtype == HALO_DIST_ZIGZAG) return fabsf(std_val * sinf(ho_pcg_uniform(s) * 2.0f * LM_PI_F) + mean);
  {
    float u = ho_pcg_uniform(s);
    float sgn = (u < 0.5f) ? -1.0f : 1.0f;
    float arg = fmaxf(1.0f - 2.0f * fabsf(u - 0.5f), 1e-30f);
    return mean - std_val * sgn * logf(arg);
  }
}

/* pcg_shared.h:311-322 */
void ho_normalize_latitude(float phi, float* phi_out, int* flip) {
  float theta = LM_PI_2F - phi;
  theta = fmodf(theta, 2.0f * LM_PI_F);
  if (theta < 0.0f) theta += 2.0f * LM_PI_F;
  *flip = theta > LM_PI_F;
  if (*flip) theta = 2.0f * LM_PI_F - theta;
  *phi_out = LM_PI_2F - theta;
}

/* pcg_shared.h:345-363 */
float ho_invert_lat_lut(float xi, const float* theta_nodes, const float* cdf_nodes, uint32_t n_nodes) {
  xi = clampf(xi, cdf_nodes[0], cdf_nodes[n_nodes - 1u]);
  uint32_t lo = 0u, hi = n_nodes - 1u;
  while (hi - lo > 1u) {
    uint32_t mid = (lo + hi) >> 1u;
    if (cdf_nodes[mid] <= xi) lo = mid; else hi = mid;
  }
  float c0 = cdf_nodes[lo];
  float c1 = cdf_nodes[lo + 1u];
  float denom = c1 - c0;
  float w = denom > 0.0f ? (xi - c0) / denom : 0.0f;
  return theta_nodes[lo] + w * (theta_nodes[lo + 1u] - theta_nodes[lo]);
}

/* pcg_shared.h:370-378 */
uint32_t ho_lat_lut_bin(float theta, const float* theta_nodes, uint32_t n_nodes) {
  float span = theta_nodes[n_nodes - 1u] - theta_nodes[0];
  float t = span > 0.0f ? (theta - theta_nodes[0]) / span : 0.0f;
  int idx = (int)(t * (float)(n_nodes - 1u));
  idx = idx < 0 ? 0 : idx;
  int last = (int)n_nodes - 2;
  idx = idx > last ? last : idx;
  return (uint32_t)idx;
}

enum { HO_LAT_FULL_SPHERE = 0, HO_LAT_NO_RANDOM = 1, HO_LAT_GAUSS_LEGACY = 3, HO_LAT_LUT = 6 }; /* pcg_shared.h:56-59 */

/* pcg_shared.h:392-437 */
void ho_sample_lat_lon_roll(HoStream* s, const HoGenParams* gp, const float* lut_theta, const float* lut_cdf,
                            const float* lut_flip, float* out_lon, float* out_lat, float* out_roll) {
  float phi = 0.0f;
  int flip = 0;
  float lon = 0.0f;
  if (gp->lat_path == HO_LAT_FULL_SPHERE) {
    float u = ho_pcg_uniform(s) * 2.0f - 1.0f;
    u = clampf(u, -1.0f, 1.0f);
    phi = asinf(u);
    lon = ho_pcg_uniform(s) * 2.0f * LM_PI_F;
  } else if (gp->lat_path == HO_LAT_NO_RANDOM) {
    phi = gp->lat_mean_rad;
  } else if (gp->lat_path == HO_LAT_GAUSS_LEGACY) {
    float raw = ho_pcg_get_dist(s, HALO_DIST_GAUSS_LEGACY, gp->lat_mean_rad, gp->lat_std_rad);
    ho_normalize_latitude(raw, &phi, &flip);
  } else if (gp->lat_path == HO_LAT_LUT) {
    float xi = ho_pcg_uniform(s);
    float colatitude = ho_invert_lat_lut(xi, lut_theta, lut_cdf, gp->lat_lut_n);
    phi = LM_PI_2F - colatitude;
    uint32_t bin = ho_lat_lut_bin(colatitude, lut_theta, gp->lat_lut_n);
    flip = ho_pcg_uniform(s) < lut_flip[bin];
  }
  if (gp->lat_path != HO_LAT_FULL_SPHERE) lon = ho_pcg_get_dist(s, gp->az_type, gp->az_mean_rad, gp->az_std_rad);
  float roll = ho_pcg_get_dist(s, gp->roll_type, gp->roll_mean_rad, gp->roll_std_rad);
  if (flip) {
    lon += LM_PI_F;
    roll += LM_PI_F;
  }
  *out_lon = lon;
  *out_lat = phi;
  *out_roll = roll;
}

/* pcg_shared.h:441-454 (== Rotation::FillMat geo3d.cpp:100-113) */
static void axis_angle_rotation_9(const float* ax, float theta, float* out) {
  float c = cosf(theta);
  float s = sinf(theta);
  float cc = 1.0f - c;
  out[0] = ax[0] * ax[0] * cc + c;
  out[1] = ax[0] * ax[1] * cc - ax[2] * s;
  out[2] = ax[0] * ax[2] * cc + ax[1] * s;
  out[3] = ax[0] * ax[1] * cc + ax[2] * s;
  out[4] = ax[1] * ax[1] * cc + c;
  out[5] = ax[1] * ax[2] * cc - ax[0] * s;
  out[6] = ax[0] * ax[2] * cc - ax[1] * s;
  out[7] = ax[1] * ax[2] * cc + ax[0] * s;
  out[8] = ax[2] * ax[2] * cc + c;
}

/* pcg_shared.h:456-467: m <- r * m */
static void chain_left_mul_9(float* m, const float* r) {
  float t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      t[i * 3 + j] = r[i * 3 + 0] * m[0 * 3 + j] + r[i * 3 + 1] * m[1 * 3 + j] + r[i * 3 + 2] * m[2 * 3 + j];
  memcpy(m, t, sizeof(t));
}

/* pcg_shared.h:473-483; simulator.cpp:224-231 */
void ho_build_crystal_rotation_9(float lon, float lat, float roll, float* mat9) {
  float ey[3] = {0.0f, 1.0f, 0.0f};
  float ez[3] = {0.0f, 0.0f, 1.0f};
  float middle[9], outer[9];
  axis_angle_rotation_9(ez, roll, mat9);
  axis_angle_rotation_9(ey, lat - LM_PI_2F, middle);
  chain_left_mul_9(mat9, middle);
  axis_angle_rotation_9(ez, lon - LM_PI_F, outer);
  chain_left_mul_9(mat9, outer);
}

/* pcg_shared.h:487-491 */
void ho_apply_inverse_mat9(const float* m, const float* d, float* o) {
  o[0] = m[0] * d[0] + m[3] * d[1] + m[6] * d[2];
  o[1] = m[1] * d[0] + m[4] * d[1] + m[7] * d[2];
  o[2] = m[2] * d[0] + m[5] * d[1] + m[8] * d[2];
}

/* Rotation::Apply geo3d.cpp:63-72; cuda_trace_backend.cu:879-881 */
void ho_apply_mat9(const float* m, const float* v, float* o) {
  for (int i = 0; i < 3; i++) o[i] = m[i * 3 + 0] * v[0] + m[i * 3 + 1] * v[1] + m[i * 3 + 2] * v[2];
}

/* pcg_shared.h:496-509 */
void ho_sample_triangle(HoStream* s, const float* vtx9, float* out_p) {
  float u = ho_pcg_uniform(s);
  float v = ho_pcg_uniform(s);
  if (u + v > 1.0f) {
    u = 1.0f - u;
    v = 1.0f - v;
  }
  for (int k = 0; k < 3; k++) {
    float a = vtx9[k], b = vtx9[3 + k], c = vtx9[6 + k];
    out_p[k] = u * (b - a) + v * (c - a) + a;
  }
}

/* pcg_shared.h:514-529 */
void ho_sample_sph_cap(HoStream* s, float lon, float lat, float half_angle, float* out_d) {
  float c_cap = cosf(half_angle);
  float u = ho_pcg_uniform(s);
  float x = u + (1.0f - u) * c_cap;
  float r = sqrtf(fmaxf(1.0f - x * x, 0.0f));
  float phi = ho_pcg_uniform(s) * 2.0f * LM_PI_F;
  float y = cosf(phi) * r;
  float z = sinf(phi) * r;
  float c_lon = cosf(lon), s_lon = sinf(lon), c_lat = cosf(lat), s_lat = sinf(lat);
  out_d[0] = c_lon * c_lat * x - s_lon * y - c_lon * s_lat * z;
  out_d[1] = s_lon * c_lat * x + c_lon * y - s_lon * s_lat * z;
  out_d[2] = s_lat * x + c_lat * z;
}

/* pcg_shared.h:550-603 */
uint32_t ho_feistel_bijection(uint32_t i, uint32_t n, uint32_t seed) {
  if (n <= 1u) return i;
  if (n == 2u) return i ^ 1u;
  uint32_t bits = 0u;
  while (bits < 30u && (1u << bits) < n) bits++;
  if ((bits & 1u) != 0u) bits++;
  uint32_t half_bits = bits >> 1u;
  uint32_t hm = (1u << half_bits) - 1u;
  const uint32_t round_const[4] = {0x9E3779B9u, 0x85EBCA6Bu, 0xC2B2AE35u, 0x27D4EB2Fu};
  uint32_t cur = i;
  for (uint32_t guard = 0u; guard < 64u; guard++) {
    uint32_t L = (cur >> half_bits) & hm;
    uint32_t R = cur & hm;
    for (uint32_t k = 0u; k < 4u; k++) {
      uint32_t f = ho_pcg_hash(seed ^ R ^ round_const[k]) & hm;
      uint32_t new_R = L ^ f;
      L = R;
      R = new_R;
    }
    uint32_t out = (L << half_bits) | R;
    if (out < n) return out;
    cur = out;
  }
  return cur % n;
}

/* pcg_shared.h:607-624 */
uint32_t ho_categorical_sample(const float* weights, uint32_t n, float u_in) {
  float total = 0.0f;
  for (uint32_t i = 0u; i < n; i++) total += fmaxf(weights[i], 0.0f);
  if (total <= 0.0f) return 0u;
  float target = u_in * total;
  float cumsum = 0.0f;
  for (uint32_t i = 0u; i < n; i++) {
    cumsum += fmaxf(weights[i], 0.0f);
    if (cumsum > target) return i;
  }
  return n - 1u;
}

/* ======================================================================================== */
/* optics                                                                                    */
/* ======================================================================================== */

/* optics_shared.h:17-24 */
float ho_reflect_ratio(float delta, float rr) {
  float d_sqrt = sqrtf(delta);
  float Rs = (rr - d_sqrt) / (rr + d_sqrt);
  Rs *= Rs;
  float Rp = (1.0f - rr * d_sqrt) / (1.0f + rr * d_sqrt);
  Rp *= Rp;
  return (Rs + Rp) * 0.5f;
}

/* traversal_shared.h:61-71 */
float ho_slab_face_t(const float d[3], const float p[3], const float n[3], float fd) {
  float denom = d[0] * n[0] + d[1] * n[1] + d[2] * n[2];
  if (denom <= 1e-5f) return 1.0e30f;
  return -(p[0] * n[0] + p[1] * n[1] + p[2] * n[2] + fd) / denom;
}

/* optics.cpp:180-197, optics.hpp:14-31 (kCoefAvr) */
double ho_ice_refractive_index(double wave_length) {
  static const float kCoefAvr[4] = {0.701777f, 1.091144f, 0.884400f, 0.796950f};
  if (wave_length < 350.0f || wave_length > 900.0f) return 1.0f;
  wave_length /= 1e3;
  double n = 1.0;
  n += kCoefAvr[0] / (1 - kCoefAvr[2] * 1e-2f / wave_length / wave_length);
  n += kCoefAvr[1] / (1 - kCoefAvr[3] * 1e2f / wave_length / wave_length);
  return sqrt(n);
}

/* ======================================================================================== */
/* projection                                                                                */
/* ======================================================================================== */

typedef struct { float x, y; int valid; } ProjXY;

/* projection_shared.h:42-45 */
static ProjXY fisheye_equal_area_fwd(float dx, float dy, float dz, float r_scale) {
  float k = r_scale / sqrtf(1.0f + clampf(dz, -1.0f + 1e-6f, 1.0f));
  ProjXY r = {k * dx, k * dy, 1};
  return r;
}
/* projection_shared.h:48-56 */
static ProjXY fisheye_equidistant_fwd(float dx, float dy, float dz, float r_scale) {
  float rho = sqrtf(dx * dx + dy * dy);
  ProjXY r = {0.0f, 0.0f, 1};
  if (rho < 1e-10f) return r;
  float theta = acosf(clampf(dz, -1.0f, 1.0f));
  float scale = r_scale * theta / (LM_PI_2F * rho);
  r.x = scale * dx;
  r.y = scale * dy;
  return r;
}
/* projection_shared.h:59-67 */
static ProjXY fisheye_stereographic_fwd(float dx, float dy, float dz, float r_scale) {
  float rho = sqrtf(dx * dx + dy * dy);
  ProjXY r = {0.0f, 0.0f, 1};
  if (rho < 1e-10f) return r;
  float theta = acosf(clampf(dz, -1.0f, 1.0f));
  float scale = r_scale * tanf(theta / 2.0f) / rho;
  r.x = scale * dx;
  r.y = scale * dy;
  return r;
}
/* projection_shared.h:70-75 */
static ProjXY fisheye_orthographic_fwd(float dx, float dy, float dz, float r_scale) {
  ProjXY r = {0.0f, 0.0f, 0};
  if (dz < 0.0f) return r;
  r.x = r_scale * dx;
  r.y = r_scale * dy;
  r.valid = 1;
  return r;
}
/* projection_shared.h:78-82 */
static ProjXY rectangular_fwd(float dx, float dy, float dz) {
  ProjXY r = {atan2f(dy, dx), asinf(clampf(dz, -1.0f, 1.0f)), 1};
  return r;
}
/* projection_shared.h:86-91 */
static ProjXY linear_fwd(float dx, float dy, float dz) {
  ProjXY r = {0.0f, 0.0f, 0};
  if (dz <= 0.0f) return r;
  r.x = dx / dz;
  r.y = dy / dz;
  r.valid = 1;
  return r;
}
/* projection_shared.h:161-166 */
static void apply_rot_transpose(const float* rot, float a, float b, float c, float* o0, float* o1, float* o2) {
  *o0 = rot[0] * a + rot[3] * b + rot[6] * c;
  *o1 = rot[1] * a + rot[4] * b + rot[7] * c;
  *o2 = rot[2] * a + rot[5] * b + rot[8] * c;
}
/* projection_shared.h:170-184 */
static void dual_fisheye_to_pixel_xy(float xn, float yn, int is_upper, int width, int height, float* fx, float* fy) {
  int half_w = width / 2;
  int short_res = half_w < height ? half_w : height;
  float r = (float)short_res / 2.0f;
  float cy = (float)height / 2.0f;
  if (is_upper) {
    float cx = (float)width / 2.0f - r;
    *fx = -yn * r + cx;
    *fy = xn * r + cy;
  } else {
    float cx = (float)width / 2.0f + r;
    *fx = yn * r + cx;
    *fy = xn * r + cy;
  }
}

static ProjXY dual_fwd(int t, float sx, float sy, float z, float r_scale) {
  if (t == HALO_LENS_DUAL_FISHEYE_EQUAL_AREA) return fisheye_equal_area_fwd(sx, sy, z, r_scale);
  if (t == HALO_LENS_DUAL_FISHEYE_EQUIDISTANT) return fisheye_equidistant_fwd(sx, sy, z, r_scale);
  if (t == HALO_LENS_DUAL_FISHEYE_STEREOGRAPHIC) return fisheye_stereographic_fwd(sx, sy, z, r_scale);
  return fisheye_orthographic_fwd(sx, sy, z, r_scale);
}

/* projection_shared.h:196-375 */
HoProjResult ho_project_exit_to_pixel(const HoProjParams* p, float wx, float wy, float wz) {
  HoProjResult r;
  memset(&r, 0, sizeof(r));
  int t = p->proj_type;
  if (t == HALO_LENS_LINEAR || t == HALO_LENS_FISHEYE_EQUAL_AREA || t == HALO_LENS_FISHEYE_EQUIDISTANT ||
      t == HALO_LENS_FISHEYE_STEREOGRAPHIC || t == HALO_LENS_FISHEYE_ORTHOGRAPHIC) {
    if ((p->visible_range == HALO_VISIBLE_UPPER && wz > 0.0f) || (p->visible_range == HALO_VISIBLE_LOWER && wz < 0.0f))
      return r;
    float cx, cy, cz;
    apply_rot_transpose(p->rot, -wx, -wy, -wz, &cx, &cy, &cz);
    ProjXY xy = {0.0f, 0.0f, 0};
    if (t == HALO_LENS_LINEAR) {
      xy = linear_fwd(cx, cy, cz);
    } else {
      if (cz <= 0.0f) return r;
      if (t == HALO_LENS_FISHEYE_EQUAL_AREA) xy = fisheye_equal_area_fwd(cx, cy, cz, 1.0f);
      else if (t == HALO_LENS_FISHEYE_EQUIDISTANT) xy = fisheye_equidistant_fwd(cx, cy, cz, 1.0f);
      else if (t == HALO_LENS_FISHEYE_STEREOGRAPHIC) xy = fisheye_stereographic_fwd(cx, cy, cz, 1.0f);
      else xy = fisheye_orthographic_fwd(cx, cy, cz, 1.0f);
    }
    if (!xy.valid) return r;
    xy.x = -xy.x;
    r.hits[0].px = (int)floorf(xy.x * p->scale + (float)p->img_w / 2.0f + 0.5f + (float)p->lens_shift_x);
    r.hits[0].py = (int)floorf(xy.y * p->scale + (float)p->img_h / 2.0f + 0.5f + (float)p->lens_shift_y);
    r.hits[0].bump_landed = 1;
    r.count = 1;
    return r;
  }
  if (t == HALO_LENS_RECTANGULAR) {
    ProjXY proj = rectangular_fwd(-wx, -wy, -wz);
    float lon = proj.x - p->az0;
    while (lon < -LM_PI_F) lon += 2.0f * LM_PI_F;
    while (lon > LM_PI_F) lon -= 2.0f * LM_PI_F;
    int raw_x = (int)floorf(lon * p->scale + (float)p->img_w / 2.0f + 0.5f);
    r.hits[0].px = ((raw_x % p->img_w) + p->img_w) % p->img_w;
    r.hits[0].py = (int)floorf(-proj.y * p->scale + (float)p->img_h / 2.0f + 0.5f);
    r.hits[0].bump_landed = 1;
    r.count = 1;
    return r;
  }
  if (t == HALO_LENS_DUAL_FISHEYE_EQUAL_AREA || t == HALO_LENS_DUAL_FISHEYE_EQUIDISTANT ||
      t == HALO_LENS_DUAL_FISHEYE_STEREOGRAPHIC || t == HALO_LENS_DUAL_FISHEYE_ORTHOGRAPHIC) {
    float sx = -wx, sy = -wy, sz = -wz;
    int is_upper = (sz >= 0.0f);
    float z_hemi = is_upper ? sz : -sz;
    ProjXY xy = dual_fwd(t, sx, sy, z_hemi, p->r_scale);
    float fx, fy;
    dual_fisheye_to_pixel_xy(xy.x, xy.y, is_upper, p->img_w, p->img_h, &fx, &fy);
    r.hits[0].px = (int)floorf(fx + 0.5f);
    r.hits[0].py = (int)floorf(fy + 0.5f);
    r.hits[0].bump_landed = 1;
    r.count = 1;
    if (p->max_abs_dz > 0.0f && fabsf(sz) < p->max_abs_dz) {
      float z_opp = -z_hemi;
      ProjXY xy2 = dual_fwd(t, sx, sy, z_opp, p->r_scale);
      float fx2, fy2;
      dual_fisheye_to_pixel_xy(xy2.x, xy2.y, !is_upper, p->img_w, p->img_h, &fx2, &fy2);
      r.hits[1].px = (int)floorf(fx2 + 0.5f);
      r.hits[1].py = (int)floorf(fy2 + 0.5f);
      r.hits[1].bump_landed = 0;
      r.count = 2;
    }
    return r;
  }
  if (t == HALO_LENS_GLOBE) {
    const float kGlobeCameraD = 4.0f;
    float cx, cy, cz;
    apply_rot_transpose(p->rot, -wx, -wy, -wz, &cx, &cy, &cz);
    if (cz >= -1.0f / kGlobeCameraD) return r;
    float denom = kGlobeCameraD + cz;
    r.hits[0].px = (int)floorf(-cx / denom * p->scale + (float)p->img_w / 2.0f + 0.5f + (float)p->lens_shift_x);
    r.hits[0].py = (int)floorf(cy / denom * p->scale + (float)p->img_h / 2.0f + 0.5f + (float)p->lens_shift_y);
    r.hits[0].bump_landed = 1;
    r.count = 1;
    return r;
  }
  return r;
}

/* scatter_accum.hpp:18-27 MakeCameraRotation + lens_proj_build.hpp:22-137 */
void ho_build_proj_params(const HaloRender* cfg, HoProjParams* p) {
  memset(p, 0, sizeof(*p));
  p->proj_type = cfg->lens_type;
  p->img_w = cfg->width;
  p->img_h = cfg->height;
  p->visible_range = cfg->visible;
  p->lens_shift_x = cfg->lens_shift[0];
  p->lens_shift_y = cfg->lens_shift[1];
  p->r_scale = 1.0f;
  p->max_abs_dz = 0.0f;
  /* rot = identity .Chain(Rz(-90+ro)) .Chain(Ry(90-el)) .Chain(Rz(az)); Chain left-multiplies (geo3d.cpp:32-46) */
  float ax_z[3] = {0, 0, 1}, ax_y[3] = {0, 1, 0};
  float rot[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tmp[9];
  axis_angle_rotation_9(ax_z, (-90.0f + cfg->view_ro) * HO_DEG2RAD, tmp);
  chain_left_mul_9(rot, tmp);
  axis_angle_rotation_9(ax_y, (90.0f - cfg->view_el) * HO_DEG2RAD, tmp);
  chain_left_mul_9(rot, tmp);
  axis_angle_rotation_9(ax_z, cfg->view_az * HO_DEG2RAD, tmp);
  chain_left_mul_9(rot, tmp);
  memcpy(p->rot, rot, sizeof(rot));

  float short_pix = (float)(cfg->width < cfg->height ? cfg->width : cfg->height); /* scatter_accum.hpp:54 */
  float fov_rad = cfg->fov * HO_DEG2RAD;
  p->scale = 1.0f;
  p->az0 = 0.0f;
  switch (cfg->lens_type) { /* lens_proj_build.hpp:22-67 */
    case HALO_LENS_LINEAR: p->scale = short_pix / 2.0f / tanf(fov_rad / 2.0f); break;
    case HALO_LENS_FISHEYE_EQUAL_AREA: p->scale = short_pix / 2.0f / sqrtf(2.0f) / sinf(fov_rad / 4.0f); break;
    case HALO_LENS_FISHEYE_EQUIDISTANT: p->scale = short_pix * HO_PI_2F / fov_rad; break;
    case HALO_LENS_FISHEYE_STEREOGRAPHIC: p->scale = short_pix / 2.0f / tanf(fov_rad / 4.0f); break;
    case HALO_LENS_FISHEYE_ORTHOGRAPHIC: p->scale = short_pix / 2.0f / sinf(fov_rad / 2.0f); break;
    case HALO_LENS_RECTANGULAR: {
      int half_w = cfg->width / 2;
      int short_res = half_w < cfg->height ? half_w : cfg->height;
      p->scale = (float)short_res / HO_PI_F;
      float z[3] = {0, 0, 1}, o[3];
      ho_apply_mat9(rot, z, o);
      p->az0 = atan2f(o[1], o[0]);
      break;
    }
    case HALO_LENS_GLOBE: p->scale = short_pix / 2.0f / tanf(fov_rad / 2.0f); break;
    default: break;
  }
  if (cfg->overlap > 0) { /* lens_proj_build.hpp:103-131; projection.cpp:192-204 */
    if (cfg->lens_type == HALO_LENS_DUAL_FISHEYE_EQUAL_AREA) {
      p->max_abs_dz = cfg->overlap;
      p->r_scale = 1.0f / sqrtf(1.0f + cfg->overlap);
    } else if (cfg->lens_type == HALO_LENS_DUAL_FISHEYE_EQUIDISTANT) {
      p->max_abs_dz = cfg->overlap;
      p->r_scale = HO_PI_2F / (HO_PI_2F + asinf(cfg->overlap));
    } else if (cfg->lens_type == HALO_LENS_DUAL_FISHEYE_STEREOGRAPHIC) {
      p->max_abs_dz = cfg->overlap;
      p->r_scale = 1.0f / tanf((HO_PI_2F + asinf(cfg->overlap)) / 2.0f);
    }
  }
}

/* ======================================================================================== */
/* latitude LUT — src/core/lat_lut.cpp:24-204                                               */
/* ======================================================================================== */
#define HO_LUT_FINE 4096
#define HO_LUT_QUAD (1 << 16)

static double proposal_lat_from_u(int type, double mean_rad, double scale_rad, double u) { /* lat_lut.cpp:30-45 */
  switch (type) {
    case HALO_DIST_UNIFORM: return (u - 0.5) * scale_rad + mean_rad;
    case HALO_DIST_ZIGZAG: return fabs(scale_rad * sin(u * 2.0 * kPiD) + mean_rad);
    case HALO_DIST_LAPLACIAN: {
      double sgn = (u < 0.5) ? -1.0 : 1.0;
      double arg = fmax(1.0 - 2.0 * fabs(u - 0.5), 1e-30);
      return mean_rad - scale_rad * sgn * log(arg);
    }
    default: return mean_rad;
  }
}

static double lerp_cum(const double* cum, double theta) { /* lat_lut.cpp:49-60 */
  double x = theta / (kPiD / HO_LUT_FINE);
  int i = (int)x;
  if (i < 0) return cum[0];
  if (i >= HO_LUT_FINE) return cum[HO_LUT_FINE];
  double f = x - i;
  return cum[i] * (1.0 - f) + cum[i + 1] * f;
}

static void degenerate_lut(double colat, float* theta, float* cdf, float* flip) { /* lat_lut.cpp:63-72 */
  float c = (float)fmin(fmax(colat, 0.0), kPiD);
  for (uint32_t i = 0; i < HALO_LUT_NODES; ++i) {
    theta[i] = c;
    cdf[i] = (float)i / (float)(HALO_LUT_NODES - 1);
    flip[i] = 0.0f;
  }
}

typedef struct { double* mass; double* flip_mass; double dtheta; } LutAcc;
static void lut_accumulate(LutAcc* a, double lat_rad, double weight) { /* lat_lut.cpp:88-104 */
  float phi_out = 0.0f;
  int flip = 0;
  ho_normalize_latitude((float)lat_rad, &phi_out, &flip);
  double theta_z = kPiD / 2.0 - (double)phi_out;
  double w = weight * sin(theta_z);
  if (w <= 0.0) return;
  int bin = (int)(theta_z / a->dtheta);
  if (bin < 0) bin = 0;
  if (bin > HO_LUT_FINE - 1) bin = HO_LUT_FINE - 1;
  a->mass[bin] += w;
  if (flip) a->flip_mass[bin] += w;
}

void ho_build_lat_lut(const HaloDist* lat, float* theta, float* cdf, float* flipp) {
  const double kDeg2Rad = kPiD / 180.0;
  double mean_rad = (double)lat->center * kDeg2Rad;
  double scale_rad = (double)lat->spread * kDeg2Rad;
  int type = lat->type;
  double dtheta = kPiD / HO_LUT_FINE;
  double* mass = (double*)calloc(HO_LUT_FINE, sizeof(double));
  double* flip_mass = (double*)calloc(HO_LUT_FINE, sizeof(double));
  double* cum_mass = (double*)calloc(HO_LUT_FINE + 1, sizeof(double));
  double* cum_flip = (double*)calloc(HO_LUT_FINE + 1, sizeof(double));
  LutAcc acc = {mass, flip_mass, dtheta};
  if (type == HALO_DIST_GAUSS) { /* lat_lut.cpp:106-116 */
    double lo = mean_rad - 12.0 * scale_rad;
    double hi = mean_rad + 12.0 * scale_rad;
    double dL = (hi - lo) / HO_LUT_QUAD;
    double inv2s2 = (scale_rad > 0.0) ? 1.0 / (2.0 * scale_rad * scale_rad) : 0.0;
    for (int i = 0; i < HO_LUT_QUAD; ++i) {
      double L = lo + (i + 0.5) * dL;
      double d = L - mean_rad;
      lut_accumulate(&acc, L, exp(-d * d * inv2s2) * dL);
    }
  } else { /* lat_lut.cpp:117-125 */
    double dU = 1.0 / HO_LUT_QUAD;
    for (int i = 0; i < HO_LUT_QUAD; ++i) {
      double u = (i + 0.5) * dU;
      lut_accumulate(&acc, proposal_lat_from_u(type, mean_rad, scale_rad, u), dU);
    }
  }
  for (int i = 0; i < HO_LUT_FINE; ++i) {
    cum_mass[i + 1] = cum_mass[i] + mass[i];
    cum_flip[i + 1] = cum_flip[i] + flip_mass[i];
  }
  double total = cum_mass[HO_LUT_FINE];
  if (!(total > 0.0)) { /* lat_lut.cpp:135-141 */
    float phi_out = 0.0f;
    int flip = 0;
    ho_normalize_latitude((float)mean_rad, &phi_out, &flip);
    degenerate_lut(kPiD / 2.0 - (double)phi_out, theta, cdf, flipp);
    goto done;
  }
  {
    double theta_lo = 0.0, theta_hi = kPiD; /* lat_lut.cpp:144-160 */
    for (int i = 0; i <= HO_LUT_FINE; ++i)
      if (cum_mass[i] / total >= 1e-7) { theta_lo = i * dtheta; break; }
    for (int i = HO_LUT_FINE; i >= 0; --i)
      if (cum_mass[i] / total <= 1.0 - 1e-7) { theta_hi = i * dtheta; break; }
    if (!(theta_hi > theta_lo)) {
      degenerate_lut(0.5 * (theta_lo + theta_hi), theta, cdf, flipp);
      goto done;
    }
    double span = theta_hi - theta_lo; /* lat_lut.cpp:163-185 */
    for (uint32_t n = 0; n < HALO_LUT_NODES; ++n) {
      double t = theta_lo + span * n / (HALO_LUT_NODES - 1);
      theta[n] = (float)t;
      cdf[n] = (float)(lerp_cum(cum_mass, t) / total);
    }
    for (uint32_t n = 1; n < HALO_LUT_NODES; ++n)
      if (cdf[n] <= cdf[n - 1]) cdf[n] = nextafterf(cdf[n - 1], INFINITY);
    for (uint32_t n = 0; n + 1 < HALO_LUT_NODES; ++n) {
      double t0 = theta[n], t1 = theta[n + 1];
      double m = lerp_cum(cum_mass, t1) - lerp_cum(cum_mass, t0);
      double fm = lerp_cum(cum_flip, t1) - lerp_cum(cum_flip, t0);
      flipp[n] = (m > 0.0) ? (float)fmin(fmax(fm / m, 0.0), 1.0) : 0.0f;
    }
    flipp[HALO_LUT_NODES - 1] = flipp[HALO_LUT_NODES - 2];
  }
done:
  free(mass);
  free(flip_mass);
  free(cum_mass);
  free(cum_flip);
}

static int float_equal(float a, float b) { return fabsf(a - b) < HO_FLOAT_EPS; } /* math.cpp FloatEqual */

/* math.cpp:555-560 IsFullSphereUniform + lat_path_selection.hpp:62-75 */
uint32_t ho_select_lat_path(const HaloAxis* a) {
  int full = a->azimuth.type == HALO_DIST_UNIFORM && float_equal(a->azimuth.center, 0.0f) &&
             float_equal(a->azimuth.spread, 360.0f) && a->latitude.type == HALO_DIST_UNIFORM &&
             float_equal(a->latitude.center, 90.0f) && float_equal(a->latitude.spread, 360.0f);
  if (full) return HO_LAT_FULL_SPHERE;
  if (a->latitude.type == HALO_DIST_NONE) return HO_LAT_NO_RANDOM;
  if (a->latitude.type == HALO_DIST_GAUSS_LEGACY) return HO_LAT_GAUSS_LEGACY;
  return HO_LAT_LUT;
}

/* ======================================================================================== */
/* prism geometry — geo3d_closedform.cpp:14-302,1318-1407; crystal.cpp:77-186,304-347;      */
/* simulator.cpp:61-129                                                                      */
/* ======================================================================================== */
/* geo3d_closedform.hpp:48-52 */
static const double kHexFaceCos[6] = {1.0, 0.5, -0.5, -1.0, -0.5, 0.5};
static const double kHexFaceSin[6] = {0.0, 0.86602540378443864676, 0.86602540378443864676, 0.0, -0.86602540378443864676,
                                      -0.86602540378443864676};

static int solve2x2(double a00, double a01, double a10, double a11, double b0, double b1, double* x, double* y) {
  double det = a00 * a11 - a01 * a10; /* geo3d_closedform.cpp:27-35 */
  if (det == 0.0) return 0;
  *x = (b0 * a11 - b1 * a01) / det;
  *y = (a00 * b1 - a10 * b0) / det;
  return 1;
}

typedef struct {
  int corner_cnt;
  double cx[12], cy[12];
  int side_present[6];
  int bounded;
} HexXs;

/* geo3d_closedform.cpp:124-302 SolveHexCrossSection */
static void solve_hex_cross_section(const double r[6], HexXs* out) {
  memset(out, 0, sizeof(*out));
  double scale = 0.0;
  for (int i = 0; i < 6; i++) scale = fmax(scale, fabs(r[i]));
  double tol = 5.0 * (double)HO_FLOAT_EPS * scale; /* GapToleranceForScale :91-93 */
  double px_[12], py_[12];
  int n = 0;
  for (int i = 0; i < 6; i++) {
    for (int j = i + 1; j < 6; j++) {
      if (j == i + 3) continue;
      double px = 0, py = 0;
      solve2x2(kHexFaceCos[i], kHexFaceSin[i], kHexFaceCos[j], kHexFaceSin[j], r[i], r[j], &px, &py);
      int feasible = 1;
      for (int m = 0; m < 6; m++) {
        if (m == i || m == j) continue;
        if (kHexFaceCos[m] * px + kHexFaceSin[m] * py > r[m] + tol) { feasible = 0; break; }
      }
      if (!feasible) continue;
      int dup = 0;
      for (int v = 0; v < n; v++) {
        double dx = px_[v] - px, dy = py_[v] - py;
        if (sqrt(dx * dx + dy * dy) <= tol) { dup = 1; break; }
      }
      if (dup) continue;
      if (n < 12) { px_[n] = px; py_[n] = py; n++; }
    }
  }
  for (int i = 0; i < 6; i++) {
    int on = 0;
    for (int v = 0; v < n; v++)
      if (fabs(kHexFaceCos[i] * px_[v] + kHexFaceSin[i] * py_[v] - r[i]) <= tol) on++;
    out->side_present[i] = (on >= 2);
  }
  int present_idx[6], p_n = 0;
  for (int i = 0; i < 6; i++) if (out->side_present[i]) present_idx[p_n++] = i;
  int has_opp = 0;
  for (int k = 0; k < p_n; k++)
    if (abs(present_idx[k] - present_idx[(k + 1) % p_n]) == 3) { has_opp = 1; break; }
  out->bounded = (p_n >= 3) && !has_opp;
  if (!out->bounded) return;
  for (int k = 0; k < p_n; k++) {
    int i = present_idx[k], j = present_idx[(k + 1) % p_n];
    double px = 0, py = 0;
    solve2x2(kHexFaceCos[i], kHexFaceSin[i], kHexFaceCos[j], kHexFaceSin[j], r[i], r[j], &px, &py);
    out->cx[k] = px;
    out->cy[k] = py;
  }
  out->corner_cnt = p_n;
}

typedef struct { /* CrystalGeom subset — crystal.hpp:78 */
  int face_cnt;
  float plane_coef[HALO_MAX_FACES * 4];
  float face_normal[HALO_MAX_FACES * 3];
  int face_number[HALO_MAX_FACES];
  int face_present[HALO_MAX_FACES];
  int face_vtx_cnt[HALO_MAX_FACES];
  float face_vtx[HALO_MAX_FACES * HALO_MAX_FACE_VTX * 3];
} HoCfGeom;

/* ComputeClosedFormPrism geo3d_closedform.cpp:1318-1407 + AdaptClosedFormPrismToCrystalGeom crystal.cpp:109-186.
 * Returns corner count (validity gate IsValidClosedFormPrism crystal.cpp:77-79: h > eps && corner_cnt >= 3). */
static int prism_cf_geom(float h, const float dist[6], HoCfGeom* g, float* ring_x, float* ring_y) {
  memset(g, 0, sizeof(*g));
  g->face_cnt = 8;
  for (int i = 0; i < 8; i++) g->face_number[i] = i + 1;
  g->face_normal[2] = 1.0f;
  g->face_normal[5] = -1.0f;
  for (int i = 0; i < 6; i++) {
    g->face_normal[(2 + i) * 3 + 0] = (float)kHexFaceCos[i];
    g->face_normal[(2 + i) * 3 + 1] = (float)kHexFaceSin[i];
  }
  float h_half = 0.5f * h;
  g->plane_coef[2] = 1.0f;
  g->plane_coef[3] = -h_half;
  g->plane_coef[6] = -1.0f;
  g->plane_coef[7] = -h_half;
  double k_r = HO_SQRT3_F / 4.0;
  double k_d = HO_SQRT3_F / 8.0;
  for (int i = 0; i < 6; i++) {
    g->plane_coef[(2 + i) * 4 + 0] = 0.5f * (float)kHexFaceCos[i];
    g->plane_coef[(2 + i) * 4 + 1] = 0.5f * (float)kHexFaceSin[i];
    g->plane_coef[(2 + i) * 4 + 3] = -(float)(k_d * (double)dist[i]);
  }
  if (h <= HO_FLOAT_EPS) return 0;
  double r_side[6];
  for (int i = 0; i < 6; i++) r_side[i] = k_r * (double)dist[i];
  HexXs xs;
  solve_hex_cross_section(r_side, &xs);
  g->face_present[0] = xs.bounded;
  g->face_present[1] = xs.bounded;
  for (int i = 0; i < 6; i++) g->face_present[2 + i] = xs.side_present[i];
  int n = xs.corner_cnt;
  float cx[12], cy[12];
  for (int c = 0; c < n; c++) {
    cx[c] = (float)xs.cx[c];
    cy[c] = (float)xs.cy[c];
    if (ring_x) { ring_x[c] = cx[c]; ring_y[c] = cy[c]; }
  }
  if (n < 3) return n;
  float z_top = 0.5f * h, z_bot = -0.5f * h;
  if (g->face_present[0]) {
    g->face_vtx_cnt[0] = n;
    float* base = g->face_vtx;
    for (int k = 0; k < n; k++) { base[k * 3] = cx[k]; base[k * 3 + 1] = cy[k]; base[k * 3 + 2] = z_top; }
  }
  if (g->face_present[1]) {
    g->face_vtx_cnt[1] = n;
    float* base = g->face_vtx + 1 * HALO_MAX_FACE_VTX * 3;
    for (int k = 0; k < n; k++) { int rev = n - 1 - k; base[k * 3] = cx[rev]; base[k * 3 + 1] = cy[rev]; base[k * 3 + 2] = z_bot; }
  }
  int present_idx[6], p_n = 0;
  for (int i = 0; i < 6; i++) if (g->face_present[2 + i]) present_idx[p_n++] = i;
  for (int k = 0; k < p_n; k++) {
    int slot = 2 + present_idx[k];
    int c_prev = (k - 1 + p_n) % p_n, c_curr = k;
    float* base = g->face_vtx + slot * HALO_MAX_FACE_VTX * 3;
    g->face_vtx_cnt[slot] = 4;
    base[0] = cx[c_prev]; base[1] = cy[c_prev]; base[2] = z_bot;
    base[3] = cx[c_curr]; base[4] = cy[c_curr]; base[5] = z_bot;
    base[6] = cx[c_curr]; base[7] = cy[c_curr]; base[8] = z_top;
    base[9] = cx[c_prev]; base[10] = cy[c_prev]; base[11] = z_top;
  }
  return n;
}

/* Crystal::PopulateFromCfGeom crystal.cpp:304-347 + detail::BuildEntrySubTris simulator.cpp:90-129 */
static void cf_geom_to_tables(const HoCfGeom* g, HaloGeomTables* out) {
  memset(out, 0, sizeof(*out));
  int p = 0;
  for (int slot = 0; slot < g->face_cnt; slot++) {
    if (!g->face_present[slot]) continue;
    out->face_n[p * 3 + 0] = g->face_normal[slot * 3 + 0];
    out->face_n[p * 3 + 1] = g->face_normal[slot * 3 + 1];
    out->face_n[p * 3 + 2] = g->face_normal[slot * 3 + 2];
    const float* coef = g->plane_coef + slot * 4;
    float norm = sqrtf(coef[0] * coef[0] + coef[1] * coef[1] + coef[2] * coef[2]);
    out->face_d[p] = (norm > HO_FLOAT_EPS) ? (coef[3] / norm) : 0.0f;
    out->face_number[p] = g->face_number[slot];
    p++;
  }
  out->face_cnt = p;
  int t = 0, present_id = 0;
  for (int slot = 0; slot < g->face_cnt; slot++) {
    if (!g->face_present[slot]) continue;
    int face_id = present_id++;
    int vtx_cnt = g->face_vtx_cnt[slot];
    if (vtx_cnt < 3) continue;
    const float* base = g->face_vtx + (size_t)slot * HALO_MAX_FACE_VTX * 3;
    for (int k = 1; k + 1 < vtx_cnt && t < HALO_MAX_TRIS; k++) {
      float* v = out->tri_v + t * 9;
      memcpy(v + 0, base + 0, 3 * sizeof(float));
      memcpy(v + 3, base + k * 3, 3 * sizeof(float));
      memcpy(v + 6, base + (k + 1) * 3, 3 * sizeof(float));
      float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]};
      float e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
      float nn[3]; /* Cross3 math.cpp:36-40 */
      nn[0] = -e2[1] * e1[2] + e1[1] * e2[2];
      nn[1] = e2[0] * e1[2] - e1[0] * e2[2];
      nn[2] = -e2[0] * e1[1] + e1[0] * e2[1];
      float raw_len = sqrtf(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
      out->tri_area[t] = raw_len / 2.0f;
      if (raw_len > 0.0f) { nn[0] /= raw_len; nn[1] /= raw_len; nn[2] /= raw_len; }
      else { nn[0] = nn[1] = nn[2] = 0.0f; }
      memcpy(out->tri_n + t * 3, nn, sizeof(nn));
      out->tri_face[t] = face_id;
      t++;
    }
  }
  out->tri_cnt = t;
}

void ho_prism_geometry(float h, const float dist[6], HaloGeomTables* out) {
  HoCfGeom g;
  int n = prism_cf_geom(h, dist, &g, NULL, NULL);
  if (!(h > HO_FLOAT_EPS && n >= 3)) { /* MakePrismClosedForm crystal.cpp:349-368 → Crystal() */
    memset(out, 0, sizeof(*out));
    return;
  }
  cf_geom_to_tables(&g, out);
}

int ho_prism_corner_ring(float h, const float dist[6], float* cx, float* cy, int* face_present8) {
  HoCfGeom g;
  int n = prism_cf_geom(h, dist, &g, cx, cy);
  for (int i = 0; i < 8; i++) face_present8[i] = g.face_present[i];
  return n;
}

/* ---------------------------------------------------------------------------------------- */
/* pyramid family: Crystal::CreatePyramid crystal.cpp:379-426 → ComputeClosedFormPyramid      */
/* geo3d_closedform.cpp:743-1460.  Same plane set and validity gates as the reference         */
/* (FillHexCrystalCoef geo3d.cpp:346-512: basal, 6 prism, 6 upper-cone, 6 lower-cone planes,   */
/* cone slope a = (sqrt3/4)/tan(wedge), legal wedge 0.1..89.9 deg, basal cut at fraction h1/h3  */
/* of the way from the shoulder to the natural apex).  The solid itself is built as the generic */
/* half-space intersection the reference's mesh builder used (plane triples in double, feasible */
/* vertices, per-plane grouping, CCW ordering: math.cpp:938-1004,1173-1216) rather than by the   */
/* closed-form event walk; on well-conditioned inputs both yield the same polytope, which the    */
/* tests pin against the reference's topology goldens (vertex count + present-face mask).        */
/* ---------------------------------------------------------------------------------------- */
typedef struct { double a, b, c, d; } HoPlane;

static int solve3_planes(const HoPlane* p, const HoPlane* q, const HoPlane* r, double out[3]) {
  /* Cramer on unit-normal planes; singular when the normals are (nearly) coplanar */
  double det = p->a * (q->b * r->c - q->c * r->b) - p->b * (q->a * r->c - q->c * r->a) + p->c * (q->a * r->b - q->b * r->a);
  if (fabs(det) < 1e-9) return 0;
  double dx = -p->d, dy = -q->d, dz = -r->d;
  out[0] = (dx * (q->b * r->c - q->c * r->b) - p->b * (dy * r->c - q->c * dz) + p->c * (dy * r->b - q->b * dz)) / det;
  out[1] = (p->a * (dy * r->c - q->c * dz) - dx * (q->a * r->c - q->c * r->a) + p->c * (q->a * dz - dy * r->a)) / det;
  out[2] = (p->a * (q->b * dz - dy * r->b) - p->b * (q->a * dz - dy * r->a) + dx * (q->a * r->b - q->b * r->a)) / det;
  return 1;
}

/* max (sign=+1) or min (sign=-1) z over the feasible 3-plane concurrences of one cone's six planes: the natural apex */
static int cone_apex_z(const HoPlane cone[6], double tol, int sign, double* z_out) {
  int found = 0;
  double best = 0.0, x[3];
  for (int i = 0; i < 6; i++)
    for (int j = i + 1; j < 6; j++)
      for (int k = j + 1; k < 6; k++) {
        if (!solve3_planes(&cone[i], &cone[j], &cone[k], x)) continue;
        int ok = 1;
        for (int m = 0; m < 6 && ok; m++)
          if (cone[m].a * x[0] + cone[m].b * x[1] + cone[m].c * x[2] + cone[m].d > tol) ok = 0;
        if (!ok) continue;
        if (!found || sign * x[2] > sign * best) best = x[2];
        found = 1;
      }
  *z_out = best;
  return found;
}

typedef struct { double ang; int idx; } HoAngIdx;
static int cmp_ang(const void* a, const void* b) {
  double x = ((const HoAngIdx*)a)->ang, y = ((const HoAngIdx*)b)->ang;
  return (x > y) - (x < y);
}

int ho_pyramid_topology(float wedge_u, float wedge_l, float h1, float h2, float h3, const float dist[6], HoCfGeom* g,
                        int* vtx_cnt_out) {
  memset(g, 0, sizeof(*g));
  if (vtx_cnt_out) *vtx_cnt_out = 0;
  g->face_cnt = 20;
  static const int kFaceNumber[20] = {1, 2, 3, 4, 5, 6, 7, 8, 13, 14, 15, 16, 17, 18, 23, 24, 25, 26, 27, 28};
  for (int i = 0; i < 20; i++) g->face_number[i] = kFaceNumber[i];
  const int has_upper = h1 > HO_FLOAT_EPS && wedge_u >= 0.1f && wedge_u <= 89.9f;
  const int has_lower = h3 > HO_FLOAT_EPS && wedge_l >= 0.1f && wedge_l <= 89.9f;
  if (!has_upper && !has_lower && h2 < HO_FLOAT_EPS) return 0; /* zero-volume guard geo3d.cpp:395-399 */
  const double k8 = (double)HO_SQRT3_F / 8.0, h2_2 = 0.5 * (double)h2;
  const double a1 = has_upper ? ((double)(HO_SQRT3_F / 4.0f)) / tan((double)wedge_u * (double)HO_DEG2RAD) : -1.0;
  const double a2 = has_lower ? ((double)(HO_SQRT3_F / 4.0f)) / tan((double)wedge_l * (double)HO_DEG2RAD) : -1.0;
  HoPlane raw[20];
  int active[20] = {0};
  for (int i = 0; i < 6; i++) {
    raw[2 + i] = (HoPlane){0.5 * kHexFaceCos[i], 0.5 * kHexFaceSin[i], 0.0, -k8 * (double)dist[i]};
    active[2 + i] = 1;
    if (has_upper) {
      raw[8 + i] = (HoPlane){0.5 * a1 * kHexFaceCos[i], 0.5 * a1 * kHexFaceSin[i], k8, -k8 * (h2_2 + a1 * (double)dist[i])};
      active[8 + i] = 1;
    }
    if (has_lower) {
      raw[14 + i] = (HoPlane){0.5 * a2 * kHexFaceCos[i], 0.5 * a2 * kHexFaceSin[i], -k8, -k8 * (h2_2 + a2 * (double)dist[i])};
      active[14 + i] = 1;
    }
  }
  /* unit-normal copies for distance-valued tolerances */
  HoPlane unit[20];
  double scale = fabs(h2_2);
  for (int s = 2; s < 20; s++) {
    if (!active[s]) continue;
    double len = sqrt(raw[s].a * raw[s].a + raw[s].b * raw[s].b + raw[s].c * raw[s].c);
    unit[s] = (HoPlane){raw[s].a / len, raw[s].b / len, raw[s].c / len, raw[s].d / len};
    scale = fmax(scale, fabs(unit[s].d));
  }
  const double tol = 5.0 * (double)HO_FLOAT_EPS * fmax(scale, 1e-3);
  double z_top = h2_2, z_bot = -h2_2;
  if (has_upper) {
    double z_apex;
    if (!cone_apex_z(unit + 8, tol, +1, &z_apex)) return 0; /* empty feasible region geo3d.cpp:486-494 */
    z_top = h2_2 + (double)h1 * (z_apex - h2_2);
  }
  if (has_lower) {
    double z_apex;
    if (!cone_apex_z(unit + 14, tol, -1, &z_apex)) return 0;
    z_bot = -h2_2 + (double)h3 * (z_apex + h2_2);
  }
  raw[0] = unit[0] = (HoPlane){0.0, 0.0, 1.0, -z_top};
  raw[1] = unit[1] = (HoPlane){0.0, 0.0, -1.0, z_bot};
  active[0] = active[1] = 1;
  /* The duplicate radius is LATERAL (geo3d_closedform.cpp:77-96 GapToleranceForScale, :393 LateralMergeTol, :702-710 InsertOrFindVertex,
   * :927 kApexMergeTol): 5 * kFloatEps x the scale of the crystal's m = 0 cross section, (sqrt3/4) max_i |dist_i| — the ruler the reference
   * applies at the apex and the upper bound of the rulers it applies at every other inset.  (Until round 5 the radius here was twice
   * 5 * kFloatEps x the largest plane constant; a steep wedge puts a cone plane's |d| far above the crystal's width, and the radius then
   * swallowed slivers the reference resolves — test_closed_form_pyramid.cpp:1579-1663, t10298: 11 faces where the reference demands >= 12.) */
  double max_dist0 = 0.0;
  for (int i = 0; i < 6; i++) max_dist0 = fmax(max_dist0, fabs((double)dist[i]));
  const double merge = 5.0 * (double)HO_FLOAT_EPS * (0.25 * 1.7320508075688772935) * fmax(max_dist0, 1e-3);
  /* vertices: feasible, de-duplicated concurrences of plane triples */
  /* Incidence: the planes that pass through a kept vertex exactly (|distance| <= 1e-9 of the crystal's size; the concurrence is computed
   * in double) — its own three and whatever else meets there — united over the candidates the duplicate filter folds into it.  (A face used
   * to claim every kept vertex within 2 tol of its plane; a corner 6e-5 off a neighbouring plane then joined that face too and tilted its
   * fan off the plane — tables that are no polytope, which the reference's closed-form builder never emits, optics.cpp:100-104.  The 488
   * topology goldens hold either way.) */
  double vx[96][3];
  unsigned vmask[96];
  const double tight = 1e-9 * fmax(scale, 1e-3);
  int nv = 0;
  for (int i = 0; i < 20; i++) {
    if (!active[i]) continue;
    for (int j = i + 1; j < 20; j++) {
      if (!active[j]) continue;
      for (int k = j + 1; k < 20; k++) {
        if (!active[k]) continue;
        double x[3];
        if (!solve3_planes(&unit[i], &unit[j], &unit[k], x)) continue;
        int ok = 1;
        unsigned mk = (1u << i) | (1u << j) | (1u << k);
        for (int m = 0; m < 20; m++) {
          if (!active[m]) continue;
          double ev = unit[m].a * x[0] + unit[m].b * x[1] + unit[m].c * x[2] + unit[m].d;
          if (ev > tight) ok = 0;
          if (fabs(ev) <= tight) mk |= 1u << m;
        }
        if (!ok) continue;
        int dup = 0;
        for (int v = 0; v < nv && !dup; v++) {
          double dx = vx[v][0] - x[0], dy = vx[v][1] - x[1], dz = vx[v][2] - x[2];
          if (sqrt(dx * dx + dy * dy + dz * dz) <= merge) {
            dup = 1;
            vmask[v] |= mk;
          }
        }
        if (dup || nv >= 96) continue;
        vmask[nv] = mk;
        memcpy(vx[nv++], x, sizeof(x));
      }
    }
  }
  if (vtx_cnt_out) *vtx_cnt_out = nv;
  /* faces: vertices lying on each plane, ordered CCW seen from outside */
  int present_cnt = 0;
  for (int s = 0; s < 20; s++) {
    if (!active[s]) continue;
    int on[HALO_MAX_FACE_VTX], n_on = 0;
    for (int v = 0; v < nv; v++)
      if (((vmask[v] >> s) & 1u) && n_on < HALO_MAX_FACE_VTX) on[n_on++] = v;
    g->plane_coef[s * 4 + 0] = (float)raw[s].a;
    g->plane_coef[s * 4 + 1] = (float)raw[s].b;
    g->plane_coef[s * 4 + 2] = (float)raw[s].c;
    g->plane_coef[s * 4 + 3] = (float)raw[s].d;
    g->face_normal[s * 3 + 0] = (float)unit[s].a;
    g->face_normal[s * 3 + 1] = (float)unit[s].b;
    g->face_normal[s * 3 + 2] = (float)unit[s].c;
    if (n_on < 3) continue;
    double c[3] = {0, 0, 0};
    for (int k = 0; k < n_on; k++) for (int a = 0; a < 3; a++) c[a] += vx[on[k]][a] / n_on;
    double n[3] = {unit[s].a, unit[s].b, unit[s].c};
    double e1[3] = {vx[on[0]][0] - c[0], vx[on[0]][1] - c[1], vx[on[0]][2] - c[2]};
    double l1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    if (l1 <= tight) continue;   /* a face whose vertices all coincide; a sliver a few 1e-5 across is a face (t10298) */
    for (int a = 0; a < 3; a++) e1[a] /= l1;
    double e2[3] = {n[1] * e1[2] - n[2] * e1[1], n[2] * e1[0] - n[0] * e1[2], n[0] * e1[1] - n[1] * e1[0]};
    HoAngIdx ord[HALO_MAX_FACE_VTX];
    for (int k = 0; k < n_on; k++) {
      double r[3] = {vx[on[k]][0] - c[0], vx[on[k]][1] - c[1], vx[on[k]][2] - c[2]};
      ord[k].ang = (k == 0) ? 0.0 : atan2(r[0] * e2[0] + r[1] * e2[1] + r[2] * e2[2], r[0] * e1[0] + r[1] * e1[1] + r[2] * e1[2]);
      if (ord[k].ang < 0.0) ord[k].ang += 2.0 * kPiD;
      ord[k].idx = on[k];
    }
    qsort(ord, (size_t)n_on, sizeof(HoAngIdx), cmp_ang);
    g->face_present[s] = 1;
    g->face_vtx_cnt[s] = n_on;
    float* base = g->face_vtx + (size_t)s * HALO_MAX_FACE_VTX * 3;
    for (int k = 0; k < n_on; k++) for (int a = 0; a < 3; a++) base[k * 3 + a] = (float)vx[ord[k].idx][a];
    present_cnt++;
  }
  return present_cnt;
}

int ho_pyramid_face_mask(float wedge_u, float wedge_l, float h1, float h2, float h3, const float dist[6], int* vtx_cnt) {
  HoCfGeom g;
  ho_pyramid_topology(wedge_u, wedge_l, h1, h2, h3, dist, &g, vtx_cnt);
  int mask = 0;
  for (int s = 0; s < 20; s++) if (g.face_present[s]) mask |= 1 << s;
  return mask;
}

void ho_pyramid_geometry(float wedge_u, float wedge_l, float h1, float h2, float h3, const float dist[6], HaloGeomTables* out) {
  HoCfGeom g;
  int nv = 0;
  int present = ho_pyramid_topology(wedge_u, wedge_l, h1, h2, h3, dist, &g, &nv);
  int tris = 0;
  for (int s = 0; s < 20; s++) if (g.face_present[s]) tris += g.face_vtx_cnt[s] - 2;
  /* IsValidClosedFormPyramid crystal.cpp:93-101; and a vertex / face table that fails Euler's count (fan triangles != 2 V - 4: an
   * ill-conditioned concurrence, 2 of 10^4 deliberately degenerate draws) is a rejected sample like the reference's MakeCrystal rejects
   * (simulator.cpp:448): the empty crystal */
  if (present < 4 || tris != 2 * nv - 4) {
    memset(out, 0, sizeof(*out));
    return;
  }
  cf_geom_to_tables(&g, out);
}

/* ======================================================================================== */
/* PartitionCrystalRayNum — simulator.cpp:519-582                                            */
/* ======================================================================================== */
void ho_partition(const float* proportions, int crystal_cnt, uint64_t ray_num, double* carry, uint64_t* c_num) {
  for (int i = 0; i < crystal_cnt; i++) c_num[i] = 0;
  if (crystal_cnt == 0 || ray_num == 0) return;
  float total_prop = 0.0f;
  for (int ci = 0; ci < crystal_cnt; ci++) total_prop += fmaxf(0.0f, proportions[ci]);
  if (total_prop <= 0.0f) return;
  uint64_t assigned = 0;
  for (int ci = 0; ci < crystal_cnt; ci++) {
    double ideal = carry[ci] + ((double)fmaxf(0.0f, proportions[ci]) / total_prop) * ray_num;
    uint64_t alloc = (uint64_t)fmax(0.0, ideal);
    carry[ci] = ideal - (double)alloc;
    c_num[ci] = alloc;
    assigned += alloc;
  }
  if (assigned == ray_num) return;
  int idx[HALO_MAX_ENTRIES * 4];
  if (assigned < ray_num) {
    uint64_t deficit = ray_num - assigned;
    /* partial_sort by carry descending: repeatedly take the max among the not-yet-chosen.  std::partial_sort
     * is not stable; ties are broken by lowest index here (the reference's tests never tie). */
    int used[HALO_MAX_ENTRIES * 4] = {0};
    for (uint64_t i = 0; i < deficit && i < (uint64_t)crystal_cnt; i++) {
      int best = -1;
      for (int c = 0; c < crystal_cnt; c++)
        if (!used[c] && (best < 0 || carry[c] > carry[best])) best = c;
      used[best] = 1;
      c_num[best]++;
      carry[best] -= 1.0;
    }
  } else {
    uint64_t surplus = assigned - ray_num;
    int used[HALO_MAX_ENTRIES * 4] = {0};
    for (uint64_t i = 0; i < surplus; i++) {
      int best = -1;
      for (int c = 0; c < crystal_cnt; c++)
        if (!used[c] && c_num[c] > 0 && (best < 0 || carry[c] < carry[best])) best = c;
      if (best < 0) break;
      used[best] = 1;
      c_num[best]--;
      carry[best] += 1.0;
    }
  }
  (void)idx;
}

/* ======================================================================================== */
/* spectrum — util/illuminant.cpp:13-134, wl_pool.hpp:49-91, color_util.hpp:29-41            */
/* ======================================================================================== */
static float daylight_spd(float cct, float wavelength) { /* illuminant.cpp:13-87 */
  if (wavelength < HO_DAY_MIN_NM || wavelength > HO_DAY_MAX_NM) return 0.0f;
  float t_inv = 1.0f / cct, t_inv2 = t_inv * t_inv, t_inv3 = t_inv2 * t_inv;
  float x_d = (cct <= 7000.0f) ? 0.244063f + 0.09911e3f * t_inv + 2.9678e6f * t_inv2 - 4.6070e9f * t_inv3
                               : 0.237040f + 0.24748e3f * t_inv + 1.9018e6f * t_inv2 - 2.0064e9f * t_inv3;
  float y_d = -3.000f * x_d * x_d + 2.870f * x_d - 0.275f;
  float denom = 0.0241f + 0.2562f * x_d - 0.7341f * y_d;
  float m1 = (-1.3515f - 1.7703f * x_d + 5.9114f * y_d) / denom;
  float m2 = (0.0300f - 31.4424f * x_d + 30.0717f * y_d) / denom;
  const int np = (int)(sizeof(HO_DAYLIGHT) / sizeof(HO_DAYLIGHT[0]));
  float fi = (wavelength - HO_DAY_MIN_NM) / (float)HO_DAY_STEP_NM;
  int i0 = (int)fi;
  float frac = fi - (float)i0;
  if (i0 >= np - 1) { i0 = np - 1; frac = 0.0f; }
  int i1 = i0 + (i0 < np - 1 ? 1 : 0);
  float s0 = HO_DAYLIGHT[i0][0] + frac * (HO_DAYLIGHT[i1][0] - HO_DAYLIGHT[i0][0]);
  float s1 = HO_DAYLIGHT[i0][1] + frac * (HO_DAYLIGHT[i1][1] - HO_DAYLIGHT[i0][1]);
  float s2 = HO_DAYLIGHT[i0][2] + frac * (HO_DAYLIGHT[i1][2] - HO_DAYLIGHT[i0][2]);
  return s0 + m1 * s1 + m2 * s2;
}

float ho_illuminant_spd(int type, float wavelength) { /* illuminant.cpp:113-134; illuminant_data.hpp:127-140 */
  switch (type) {
    case HALO_ILLUM_D50: return daylight_spd(5003.0f, wavelength);
    case HALO_ILLUM_D55: return daylight_spd(5503.0f, wavelength);
    case HALO_ILLUM_D65: return daylight_spd(6504.0f, wavelength);
    case HALO_ILLUM_D75: return daylight_spd(7504.0f, wavelength);
    case HALO_ILLUM_A: {
      if (wavelength < HO_DAY_MIN_NM || wavelength > HO_DAY_MAX_NM) return 0.0f;
      if (wavelength <= 0.0f) return 0.0f;
      float ratio = 560.0f / wavelength;
      float ratio5 = ratio * ratio * ratio * ratio * ratio;
      float exp_ref = expf(1.4388e7f / (2856.0f * 560.0f));
      float exp_lam = expf(1.4388e7f / (2856.0f * wavelength));
      return 100.0f * ratio5 * (exp_ref - 1.0f) / (exp_lam - 1.0f);
    }
    case HALO_ILLUM_E:
      if (wavelength < HO_DAY_MIN_NM || wavelength > HO_DAY_MAX_NM) return 0.0f;
      return 1.0f;
  }
  return 0.0f;
}

void ho_cmf(float wl, float* x, float* y, float* z) { /* wl_pool.hpp:49-59 */
  int key = (int)(wl + 0.5f);
  if (key < HO_CMF_MIN_NM || key > HO_CMF_MAX_NM) { *x = *y = *z = 0.0f; return; }
  int idx = key - HO_CMF_MIN_NM;
  *x = HO_CMF[idx][0];
  *y = HO_CMF[idx][1];
  *z = HO_CMF[idx][2];
}

/* ======================================================================================== */
/* emit-gate filters: FilterSpec::Check semantics (core/filter_spec.hpp:42-45, filter_spec.cpp)  */
/* in the byte-sequence form of shared/filter_shared.h:53-315; canonicalisation                  */
/* Crystal::ReduceRaypath crystal.cpp:514-600; sigma_a / D-applicability crystal.cpp:708-730     */
/* ======================================================================================== */
static void p_canonical_shift(uint8_t* data, int size) { /* crystal.cpp:514-532 */
  int first_pri = -1;
  for (int i = 0; i < size; i++) {
    int x = data[i];
    if (x < 3) continue;
    int pyr = x / 10, pri = x % 10;
    if (first_pri < 0) first_pri = pri;
    pri += 6 - first_pri;
    pri %= 6;
    pri += 3;
    data[i] = (uint8_t)(pyr * 10 + pri);
  }
}

static int seq_less(const uint8_t* a, const uint8_t* b, int n) {
  for (int i = 0; i < n; i++) if (a[i] != b[i]) return a[i] < b[i];
  return 0;
}

void ho_reduce_raypath(const uint8_t* rp, int n, int symmetry, int sigma_a, int d_applicable, uint8_t* out) {
  memcpy(out, rp, (size_t)n);
  if (symmetry == 0) return;
  uint8_t alt[HALO_MAX_HITS + 1];
  if (symmetry & HALO_SYM_P) p_canonical_shift(out, n);
  if ((symmetry & HALO_SYM_D) && d_applicable) {
    for (int i = 0; i < n; i++) {
      int x = out[i];
      if (x < 3) { alt[i] = (uint8_t)x; continue; }
      int pyr = x / 10, pri = x % 10 - 3;
      pri = (sigma_a - pri + 6) % 6;
      alt[i] = (uint8_t)(pyr * 10 + pri + 3);
    }
    if (symmetry & HALO_SYM_P) p_canonical_shift(alt, n);
    if (seq_less(alt, out, n)) memcpy(out, alt, (size_t)n);
  }
  if (symmetry & HALO_SYM_B) {
    int changed = 0;
    for (int i = 0; i < n; i++) {
      int x = out[i];
      if (x <= 2) { alt[i] = (uint8_t)(3 - x); changed = 1; }
      else if (x >= 13 && x <= 18) { alt[i] = (uint8_t)(x + 10); changed = 1; }
      else if (x >= 23 && x <= 28) { alt[i] = (uint8_t)(x - 10); changed = 1; }
      else alt[i] = (uint8_t)x;
    }
    if (changed && seq_less(alt, out, n)) memcpy(out, alt, (size_t)n);
  }
}

int ho_compute_sigma_a(float roll_mean_deg) { /* crystal.cpp:720-726 */
  if (fabsf(roll_mean_deg) > 1e6f) return 0;
  int n = ((int)roundf(roll_mean_deg / 30.0f) % 6 + 6) % 6;
  return (6 - n) % 6;
}

int ho_is_d_applicable(const HaloAxis* d) { /* crystal.cpp:708-730, math.cpp IsAzRotationallySymmetric */
  int az_sym = d->azimuth.type == HALO_DIST_UNIFORM && float_equal(d->azimuth.spread, 360.0f);
  float remainder = fmodf(fmodf(d->roll.center, 30.0f) + 30.0f, 30.0f);
  return az_sym && (float_equal(remainder, 0.0f) || float_equal(remainder, 30.0f));
}

static int filter_match_term(const HaloFilterTerm* t, int symmetry, int sigma_a, int dap, const uint8_t* path, int len,
                             const float dir[3], int crystal_id) {
  uint8_t a[HALO_MAX_HITS + 1], b[HALO_MAX_HITS + 1];
  switch (t->type) {
    case HALO_FILTER_NONE: return 1;
    case HALO_FILTER_RAYPATH: {
      if (len != t->raypath_len) return 0;
      ho_reduce_raypath(path, len, symmetry, sigma_a, dap, a);
      ho_reduce_raypath(t->raypath, t->raypath_len, symmetry, sigma_a, dap, b);
      return memcmp(a, b, (size_t)len) == 0;
    }
    case HALO_FILTER_ENTRY_EXIT: {
      if (len == 0 || (uint32_t)len < t->min_len) return 0;
      if (t->max_len != 0 && (uint32_t)len > t->max_len) return 0;
      if (!t->has_entry && !t->has_exit) return 1;
      uint8_t ee[2], want[2];
      int n = 0;
      if (t->has_entry) { ee[n] = path[0]; want[n] = (uint8_t)t->entry; n++; }
      if (t->has_exit) { ee[n] = path[len - 1]; want[n] = (uint8_t)t->exit_face; n++; }
      ho_reduce_raypath(ee, n, symmetry, sigma_a, dap, a);
      ho_reduce_raypath(want, n, symmetry, sigma_a, dap, b);
      return memcmp(a, b, (size_t)n) == 0;
    }
    case HALO_FILTER_DIRECTION: { /* device_filter_desc.cpp:57-65 + filter_shared.h:226-229 */
      float lon = t->az * HO_DEG2RAD, lat = t->el * HO_DEG2RAD;
      float fx = cosf(lat) * cosf(lon), fy = cosf(lat) * sinf(lon), fz = sinf(lat);
      float rc = cosf(t->radii * HO_DEG2RAD);
      return fx * dir[0] + fy * dir[1] + fz * dir[2] > rc;
    }
    case HALO_FILTER_CRYSTAL: return crystal_id == t->crystal_id;
  }
  return 0;
}

int ho_filter_check(const HaloFilter* f, const HaloAxis* axis, const uint8_t* path, int len, const float dir_world[3], int crystal_id) {
  int dap = ho_is_d_applicable(axis);
  int sigma_a = dap ? ho_compute_sigma_a(axis->roll.center) : 0;
  int m = 0;
  if (!f->is_complex) {
    m = filter_match_term(&f->terms[0], f->symmetry, sigma_a, dap, path, len, dir_world, crystal_id);
  } else { /* ComplexSpec::Match: OR of AND-clauses; empty → false (filter_spec.cpp:274-288) */
    int idx = 0;
    for (int o = 0; o < f->or_count && !m; o++) {
      int all = 1;
      for (int k = 0; k < f->and_counts[o]; k++)
        if (all && !filter_match_term(&f->terms[idx + k], f->symmetry, sigma_a, dap, path, len, dir_world, crystal_id)) all = 0;
      idx += f->and_counts[o];
      m = all;
    }
  }
  return f->action == 0 ? m : !m;
}

/* The colour pass of the emit gate for one exit — simulator.cpp:689-716 (per ColorSpecGroup: CheckSummandMask, mapped bits OR'd into
 * the carried mask), cuda_trace_backend.cu:498-527 ApplyLayerColorBits.  Each predicate carries its own symmetry. */
uint64_t ho_color_mask(const HaloColorSet* cs, const HaloAxis* axis, const uint8_t* path, int len, const float dir_world[3], int crystal_id,
                       uint64_t carried) {
  int dap = ho_is_d_applicable(axis);
  int sigma_a = dap ? ho_compute_sigma_a(axis->roll.center) : 0;
  uint64_t m = carried;
  for (int k = 0; k < cs->term_count; k++) {
    const HaloColorTerm* ct = &cs->terms[k];
    if (ct->bit >= 0 && ct->bit < 64 && filter_match_term(&ct->predicate, ct->symmetry, sigma_a, dap, path, len, dir_world, crystal_id))
      m |= 1ull << ct->bit;
  }
  return m;
}

/* ======================================================================================== */
/* whole path: the backend state machine of include/halo_trace.h on the CPU                  */
/* ======================================================================================== */
typedef struct { float n_idx, spd_weight, cmf[3]; } HoWlEntry; /* wl_pool.hpp:29-35 */
#define HO_CONT_STRIDE 7 /* dx dy dz w wl_idx | colour mask (2 x 32 bit) */

/* Stream nonces — cuda_trace_backend.cu:259-278, pcg_shared.h:119-120 */
#define NONCE_TRANSIT 0xA5A5A5A5u
#define NONCE_GATE 0x5A5A5A5Au
#define NONCE_GEN 0x3C9A7F11u
#define NONCE_SHUFFLE 0xB17CA3D9u
#define NONCE_WL 0x9E3779B9u
#define NONCE_SHAPE_HOST 0x6A09E667u /* ours: host shape-scalar stream (the reference draws shapes with mt19937) */

struct HoBackend {
  uint32_t seed;
  uint64_t gen_count, gate_count, transit_count, shape_count;
  int capture, geom_clock, threads;
  HaloScene scene;
  HaloRender render;
  HaloWl wl;
  HoProjParams proj;
  HoWlEntry pool[HALO_WL_POOL_MAX];
  uint32_t pool_size;
  int in_session, layer_idx;
  double carry[HALO_MAX_LAYERS][HALO_MAX_ENTRIES];
  /* accumulator */
  float* xyz;
  /* option "acc64" (not the reference's arithmetic — a checker's precaution, off by default): pixel hits are summed in double, first in a
   * small per-thread pixel cache, then in xyz64 / lanes64, and reach the float image at readback.  The reference adds every hit into a
   * float image (accum_shared.h:56-62), which stops being exact once a pixel is large: a 6 Mi-ray session on a 512x256 image already
   * reads 0.12 % high in sum(Y) against landed weight x cmf_y — rounding noise of the accumulator, not of the algorithm, and larger than
   * the 3e-3 image tolerance of the production-size parity tests.  The per-thread caches also take the hot pixels off the shared image
   * (the contended omp atomics were what kept the all-cores CPU baseline from scaling). */
  int acc64;
  int rehit_cuda; /* next-face strategy: 0 = legacy CPU (relaxed threshold, optics.cpp:127-146), 1 = CUDA (source-face skip, cu:1027-1050) */
  double* xyz64;
  double* lanes64;
  int acc_w, acc_h;
  double landed;
  /* continuation pool: 5 floats per ray (d, w, wl_idx) */
  float* cont;
  uint64_t cont_n, cont_cap;
  float* cont_in;
  uint64_t cont_in_n;
  int cont_shuffle;
  HaloFilter* filters;
  int filter_count;
  HaloColorSet* color_sets;
  int color_set_count;
  HaloColorClass color_classes[HALO_COLOR_MAX_CLASSES];
  int color_class_count;
  float* lanes; /* class_count x W x H */
  int lanes_w, lanes_h;
  /* consumer: RenderConsumer::internal_xyz_ / comp_xyz_ / total_intensity_ (server/render.hpp) */
  float* cons_sum;
  float* cons_comp;
  int cons_w, cons_h;
  float total_intensity; /* float in the reference (render.hpp) */
  /* capture */
  HaloExitRecord* exits;
  uint64_t exit_n, exit_cap;
};

static void materialise_acc64(HoBackend* b);

HoBackend* ho_create(uint32_t seed) {
  HoBackend* b = (HoBackend*)calloc(1, sizeof(HoBackend));
  b->seed = seed;
  b->geom_clock = 32; /* simulator.hpp:144 kSmallBatchRayNum */
  b->threads = 1;
  return b;
}
void ho_destroy(HoBackend* b) {
  if (!b) return;
  free(b->xyz);
  free(b->xyz64);
  free(b->lanes64);
  free(b->cont);
  free(b->cont_in);
  free(b->exits);
  free(b->cons_sum);
  free(b->cons_comp);
  free(b->filters);
  free(b->color_sets);
  free(b->lanes);
  free(b);
}
int ho_set_option(HoBackend* b, const char* key, int64_t v) {
  if (!strcmp(key, "capture_exits")) b->capture = (int)v;
  else if (!strcmp(key, "geom_clock")) b->geom_clock = (int)(v > 0 ? v : 32);
  else if (!strcmp(key, "threads")) b->threads = (int)(v > 0 ? v : 1);
  else if (!strcmp(key, "acc64")) b->acc64 = v ? 1 : 0;
  else if (!strcmp(key, "rehit_strategy")) b->rehit_cuda = v ? 1 : 0;
  else if (!strcmp(key, "rank")) {
    uint64_t base = (uint64_t)v << 40; /* disjoint 64-bit counter ranges per shard */
    b->gen_count = b->gate_count = b->transit_count = b->shape_count = base;
  } else if (!strcmp(key, "ray_base")) { /* first 64-bit ray index of the session (SplitPcgRayBase trace_backend.hpp:184) */
    b->gen_count = b->gate_count = b->transit_count = b->shape_count = (uint64_t)v;
  } else return HALO_FATAL;
  return HALO_OK;
}

static void build_wl_pool(HoBackend* b) { /* wl_pool.hpp:67-91 */
  if (b->wl.illuminant >= 0) {
    uint32_t M = b->wl.pool_size > 0 ? (uint32_t)b->wl.pool_size : 64u;
    if (M > HALO_WL_POOL_MAX) M = HALO_WL_POOL_MAX;
    b->pool_size = M;
    for (uint32_t m = 0; m < M; ++m) {
      float wl = 380.0f + ((float)m + 0.5f) * 400.0f / (float)M;
      HoWlEntry* e = &b->pool[m];
      e->n_idx = (float)ho_ice_refractive_index(wl);
      e->spd_weight = ho_illuminant_spd(b->wl.illuminant, wl);
      ho_cmf(wl, &e->cmf[0], &e->cmf[1], &e->cmf[2]);
    }
  } else {
    b->pool_size = 1;
    HoWlEntry* e = &b->pool[0];
    e->n_idx = (float)ho_ice_refractive_index(b->wl.wavelength);
    e->spd_weight = b->wl.weight;
    ho_cmf(b->wl.wavelength, &e->cmf[0], &e->cmf[1], &e->cmf[2]);
  }
}

int ho_set_filters(HoBackend* b, const HaloFilter* filters, int32_t count) {
  free(b->filters);
  b->filters = NULL;
  b->filter_count = count;
  if (count > 0) {
    b->filters = (HaloFilter*)malloc((size_t)count * sizeof(HaloFilter));
    memcpy(b->filters, filters, (size_t)count * sizeof(HaloFilter));
  }
  return HALO_OK;
}

int ho_set_color(HoBackend* b, const HaloColorSet* sets, int32_t n_sets, const HaloColorClass* classes, int32_t n_classes) {
  if (n_classes > HALO_COLOR_MAX_CLASSES) return HALO_FATAL;
  free(b->color_sets);
  b->color_sets = NULL;
  b->color_set_count = n_sets;
  if (n_sets > 0) {
    b->color_sets = (HaloColorSet*)malloc((size_t)n_sets * sizeof(HaloColorSet));
    memcpy(b->color_sets, sets, (size_t)n_sets * sizeof(HaloColorSet));
  }
  b->color_class_count = n_classes;
  if (n_classes > 0) memcpy(b->color_classes, classes, (size_t)n_classes * sizeof(HaloColorClass));
  return HALO_OK;
}

/* TraceBackend::ReadbackClassLanes trace_backend.hpp:471-493: copy + zero */
int ho_readback_class_lanes(HoBackend* b, float* lanes, int width, int height, int class_count) {
  if (class_count != b->color_class_count || width != b->lanes_w || height != b->lanes_h || !b->lanes) return HALO_FATAL;
  materialise_acc64(b);
  size_t n = (size_t)class_count * width * height;
  memcpy(lanes, b->lanes, n * sizeof(float));
  memset(b->lanes, 0, n * sizeof(float));
  return HALO_OK;
}

int ho_begin(HoBackend* b, const HaloScene* scene, const HaloRender* render, const HaloWl* wl, uint64_t hint) {
  (void)hint;
  if (b->in_session) return HALO_FATAL;
  b->scene = *scene;
  b->render = *render;
  b->wl = *wl;
  ho_build_proj_params(render, &b->proj);
  build_wl_pool(b);
  if (!b->xyz || b->acc_w != render->width || b->acc_h != render->height) {
    free(b->xyz);
    b->acc_w = render->width;
    b->acc_h = render->height;
    b->xyz = (float*)calloc((size_t)b->acc_w * b->acc_h * 3, sizeof(float));
    free(b->xyz64);
    b->xyz64 = NULL;
    b->landed = 0.0;
  }
  if (b->color_class_count > 0 && (!b->lanes || b->lanes_w != render->width || b->lanes_h != render->height)) {
    free(b->lanes);
    b->lanes_w = render->width;
    b->lanes_h = render->height;
    b->lanes = (float*)calloc((size_t)b->color_class_count * b->lanes_w * b->lanes_h, sizeof(float));
    free(b->lanes64);
    b->lanes64 = NULL;
  }
  b->in_session = 1;
  b->layer_idx = 0;
  b->cont_n = 0;
  b->cont_in_n = 0;
  b->exit_n = 0;
  /* CpuTraceBackend resets the partition carry per session (cpu_trace_backend.cpp:301); the legacy path and
   * CUDA keep it across batches (simulator.cpp:1180-1187).  We follow legacy: keep. */
  return HALO_OK;
}

int ho_end(HoBackend* b) {
  b->in_session = 0;
  return HALO_OK;
}

/* per-(layer,ci) immutable context */
typedef struct {
  const HoBackend* b;
  int layer, ci;
  int final_layer;
  float prob;
  int max_hits;
  HoGenParams gp;
  float lut_theta[HALO_LUT_NODES], lut_cdf[HALO_LUT_NODES], lut_flip[HALO_LUT_NODES];
  float sun_lon, sun_lat, sun_half;
  const HaloGeomTables* shapes; /* pool */
  uint32_t shape_cnt;
  uint32_t geom_clock;
  uint32_t gen_seed, gate_seed, transit_seed;
  uint64_t gen_base, gate_base, transit_base;
  /* continuation input slice */
  const float* cont_in;
  uint64_t cont_in_n, ci_start;
  int shuffle;
  uint32_t shuffle_seed;
  const HaloHostRays* host;
  int crystal_id;
  const HaloFilter* filter; /* NULL = pass-all */
  const HaloAxis* axis;
  const HaloColorSet* color; /* NULL = this entry sets no colour bits */
} HoCiCtx;

#define HO_PC_LOG2 13
typedef struct { /* option "acc64": one thread's pixel cache, direct-mapped; pix < 0 = free */
  int64_t pix[1 << HO_PC_LOG2];
  double v[1 << HO_PC_LOG2][3];
} HoPixCache;

typedef struct { /* per-thread output sink */
  HoBackend* b;
  HoPixCache* pc; /* NULL = add into the float image directly (the reference's arithmetic) */
  double landed;
  double exit_w_sum;
  uint64_t exit_count;
  uint64_t pixel_hits;
} HoSink;

static void pixcache_flush_slot(HoBackend* b, HoPixCache* pc, uint32_t slot) {
  double* dst = b->xyz64 + (size_t)pc->pix[slot] * 3;
  for (int k = 0; k < 3; k++) {
    double v = pc->v[slot][k];
    pc->v[slot][k] = 0.0;
    if (v != 0.0) {
#ifdef _OPENMP
#pragma omp atomic
#endif
      dst[k] += v;
    }
  }
  pc->pix[slot] = -1;
}

/* option "acc64": what the double accumulators hold goes into the float image (called by every reader of b->xyz / b->lanes) */
static void materialise_acc64(HoBackend* b) {
  if (b->xyz64) {
    size_t n = (size_t)b->acc_w * b->acc_h * 3;
    for (size_t i = 0; i < n; i++) {
      b->xyz[i] = (float)((double)b->xyz[i] + b->xyz64[i]);
      b->xyz64[i] = 0.0;
    }
  }
  if (b->lanes64 && b->lanes) {
    size_t n = (size_t)b->color_class_count * b->lanes_w * b->lanes_h;
    for (size_t i = 0; i < n; i++) {
      b->lanes[i] = (float)((double)b->lanes[i] + b->lanes64[i]);
      b->lanes64[i] = 0.0;
    }
  }
}

static void emit_pixel(HoBackend* b, HoSink* sink, const HoWlEntry* wle, const float exit_world[3], float w, uint64_t cmask, int32_t* primary_pix) {
  /* EmitToDeviceXyz cuda_trace_backend.cu:433-480 == ScatterOutgoingToXyz scatter_accum.hpp:47-110 */
  HoProjResult r = ho_project_exit_to_pixel(&b->proj, exit_world[0], exit_world[1], exit_world[2]);
  *primary_pix = -1;
  for (int hi = 0; hi < r.count; ++hi) {
    int px = r.hits[hi].px, py = r.hits[hi].py;
    if (px >= 0 && px < b->proj.img_w && py >= 0 && py < b->proj.img_h) {
      size_t pix = (size_t)py * (size_t)b->proj.img_w + (size_t)px;
      float* dst = b->xyz + pix * 3; /* AccumXyzToPixel accum_shared.h:56-62 */
      float a0 = wle->cmf[0] * w, a1 = wle->cmf[1] * w, a2 = wle->cmf[2] * w;
      if (sink->pc) {
        HoPixCache* pc = sink->pc;
        uint32_t slot = ((uint32_t)pix * 2654435761u) >> (32 - HO_PC_LOG2);
        if (pc->pix[slot] != (int64_t)pix) {
          if (pc->pix[slot] >= 0) pixcache_flush_slot(b, pc, slot);
          pc->pix[slot] = (int64_t)pix;
        }
        pc->v[slot][0] += (double)a0;
        pc->v[slot][1] += (double)a1;
        pc->v[slot][2] += (double)a2;
      } else {
#ifdef _OPENMP
#pragma omp atomic
      dst[0] += a0;
#pragma omp atomic
      dst[1] += a1;
#pragma omp atomic
      dst[2] += a2;
#else
      dst[0] += a0;
      dst[1] += a1;
      dst[2] += a2;
#endif
      }
      /* FanColorClassLanes cu:535-556: primary AND overlap hits feed every class the mask satisfies */
      for (int k = 0; k < b->color_class_count; k++) {
        uint64_t bits = b->color_classes[k].bits;
        if (bits == 0) continue;
        uint64_t matched = cmask & bits;
        int ok = b->color_classes[k].combine_all ? (matched == bits) : (matched != 0);
        if (ok) {
          float* lane = b->lanes + (size_t)k * (size_t)b->proj.img_w * (size_t)b->proj.img_h + pix;
          float yv = wle->cmf[1] * w;
          if (sink->pc) {
            double* l64 = b->lanes64 + (size_t)k * (size_t)b->proj.img_w * (size_t)b->proj.img_h + pix;
#ifdef _OPENMP
#pragma omp atomic
#endif
            *l64 += (double)yv;
          } else {
#ifdef _OPENMP
#pragma omp atomic
#endif
          *lane += yv;
          }
        }
      }
      sink->pixel_hits++;
      if (r.hits[hi].bump_landed) {
        sink->landed += (double)w;
        *primary_pix = (int32_t)pix;
      }
    }
  }
}

typedef struct { float d[3], p[3], w; int face; int depth; } HoSeg;

/* Emit gate for one outgoing candidate — CollectData simulator.cpp:665-762 (no filter: pass-all). */
static void emit_gate(const HoCiCtx* c, HoSink* sink, HoStream* gate, const float rot[9], const HoWlEntry* wle,
                      uint32_t wl_idx, const float d_local[3], float w, uint32_t root, int seq, const uint8_t* path,
                      int path_len, uint64_t carried) {
  HoBackend* b = sink->b;
  float exit_world[3];
  ho_apply_mat9(rot, d_local, exit_world);
  /* physical filter: fail = the ray terminates (CollectData simulator.cpp:689,725-728) */
  if (c->filter && !ho_filter_check(c->filter, c->axis, path, path_len, exit_world, c->crystal_id)) return;
  /* raypath colour, non-destructive, after the physical filter (ApplyLayerColorBits cu:498-527) */
  uint64_t cmask = carried;
  if (c->color) cmask = ho_color_mask(c->color, c->axis, path, path_len, exit_world, c->crystal_id, carried);
  int pass_prob = 0;
  if (c->prob > 0.0f) pass_prob = (c->prob >= 1.0f) ? 1 : (ho_pcg_uniform(gate) < c->prob); /* rng.GetUniform() < prob_ :719 */
  if (pass_prob) {
    if (c->final_layer) return; /* continue with no next layer → dropped (simulator.cpp:719-722, cu:966-970) */
    uint64_t slot;
#ifdef _OPENMP
#pragma omp atomic capture
#endif
    slot = b->cont_n++;
    if (slot < b->cont_cap) {
      float* o = b->cont + slot * HO_CONT_STRIDE;
      o[0] = exit_world[0]; o[1] = exit_world[1]; o[2] = exit_world[2]; o[3] = w; o[4] = (float)wl_idx;
      memcpy(o + 5, &cmask, 8); /* the colour mask rides with the continuation (cu:922,1129) */
    }
    return;
  }
  int32_t pix = -1;
  emit_pixel(b, sink, wle, exit_world, w, cmask, &pix);
  sink->exit_count++;
  sink->exit_w_sum += (double)w;
  if (b->capture) {
    uint64_t slot;
#ifdef _OPENMP
#pragma omp atomic capture
#endif
    slot = b->exit_n++;
    if (slot < b->exit_cap) {
      HaloExitRecord* rec = &b->exits[slot];
      memset(rec, 0, sizeof(*rec));
      rec->dir[0] = exit_world[0]; rec->dir[1] = exit_world[1]; rec->dir[2] = exit_world[2];
      rec->weight = w;
      rec->root = root;
      rec->seq = (uint16_t)seq;
      rec->layer = (uint8_t)c->layer;
      rec->path_len = (uint8_t)(path_len < HALO_PATH_CAP ? path_len : HALO_PATH_CAP);
      memcpy(rec->path, path, rec->path_len);
      rec->pixel = pix;
      rec->crystal_id = (uint16_t)c->crystal_id;
      rec->wl_idx = (uint16_t)wl_idx;
      rec->color_mask = cmask;
    }
  }
}

/* PropagateSlab for one ray — optics.cpp:64-158.  The reference excludes the face a ray stands on in two ways (traversal_shared.h:23-29):
 * its CPU path by a relaxed accept threshold after the loop (here, skip_src = 0), its CUDA backend by skipping the face inside the loop
 * (cu:1027-1050; skip_src = 1).  They agree on a convex body; they part ways where an entry point lies off its face's plane by more than
 * 1e-5 (the fan of a face whose corners the vertex merge moved): the CPU path lets the outside reflection "re-hit" the entry face. */
static int propagate_slab(const HaloGeomTables* g, const float d[3], const float p[3], int src, float p_out[3], int skip_src) {
  float t_far = 1e30f;
  int far_face = -1;
  for (int fi = 0; fi < g->face_cnt; fi++) {
    if (skip_src && fi == src) continue;
    float t = ho_slab_face_t(d, p, g->face_n + fi * 3, g->face_d[fi]);
    if (t < t_far) { t_far = t; far_face = fi; }
  }
  float eps_thr = (skip_src || (src >= 0 && far_face != src)) ? -HO_FLOAT_EPS : HO_FLOAT_EPS;
  if (far_face >= 0 && t_far > eps_thr) {
    p_out[0] = p[0] + t_far * d[0];
    p_out[1] = p[1] + t_far * d[1];
    p_out[2] = p[2] + t_far * d[2];
    return far_face;
  }
  p_out[0] = p[0]; p_out[1] = p[1]; p_out[2] = p[2];
  return -1;
}

/* One root ray through one crystal: SimulateOneWavelength hit loop simulator.cpp:1308-1336
 * (TraceRayBasicInfo :585, HitSurface optics.cpp:18-53, CollectData :665). Generic two-children form. */
static void trace_root(const HoCiCtx* c, HoSink* sink, const HaloGeomTables* g, const float rot[9], float n_idx,
                       const HoWlEntry* wle, uint32_t wl_idx, const float d0[3], const float p0[3], float w0, int face0,
                       uint32_t root, HoStream* gate, uint64_t carried) {
  if (face0 < 0 || face0 >= g->face_cnt) return; /* HitSurface kInvalidId guard optics.cpp:27-32 */
  HoSeg cur[2], nxt[4];
  uint8_t path_cur[2][HALO_MAX_HITS + 2], path_nxt[4][HALO_MAX_HITS + 2];
  int plen_cur[2], plen_nxt[4];
  int n_cur = 1;
  memcpy(cur[0].d, d0, 12);
  memcpy(cur[0].p, p0, 12);
  cur[0].w = w0;
  cur[0].face = face0;
  path_cur[0][0] = (uint8_t)g->face_number[face0]; /* InitRay_other_info RecorderAppend(GetFn(to_face_)) :270 */
  plen_cur[0] = 1;
  for (int i = 0; i < c->max_hits && n_cur > 0; i++) {
    int n_nxt = 0;
    for (int k = 0; k < n_cur && k < 2; k++) {
      HoSeg* s = &cur[k];
      const float* nrm = g->face_n + s->face * 3;
      float cos_theta = s->d[0] * nrm[0] + s->d[1] * nrm[1] + s->d[2] * nrm[2];
      float rr = cos_theta > 0 ? n_idx : 1.0f / n_idx;
      float dd = (1.0f - rr * rr) / (cos_theta * cos_theta) + rr * rr;
      int tir = dd <= 0.0f;
      float w_refl = ho_reflect_ratio(fmaxf(dd, 0.0f), rr) * s->w;
      float w_refr = tir ? -1.0f : s->w - w_refl;
      float d_refl[3], d_refr[3];
      for (int j = 0; j < 3; j++) {
        d_refl[j] = s->d[j] - 2 * cos_theta * nrm[j];
        d_refr[j] = tir ? d_refl[j] : rr * s->d[j] - (rr - sqrtf(dd)) * cos_theta * nrm[j];
      }
      const float* dirs[2] = {d_refl, d_refr};
      float ws[2] = {w_refl, w_refr};
      for (int ch = 0; ch < 2; ch++) {
        if (ws[ch] < 0) continue; /* TIR child: dropped (CollectData case 0) */
        float p_new[3];
        /* CUDA strategy: the child that leaves through the face it stands on (the reflection of an entering ray, the refraction of an inner
         * one) is emitted where it is — cu:1012-1170 propagates the inner ray only */
        const int outward = (cos_theta < 0) ? (ch == 0) : (ch == 1 && !tir);
        int f_new = (c->b->rehit_cuda && outward) ? -1 : propagate_slab(g, dirs[ch], s->p, s->face, p_new, c->b->rehit_cuda);
        if (f_new < 0) { /* outgoing candidate */
          emit_gate(c, sink, gate, rot, wle, wl_idx, dirs[ch], ws[ch], root, 2 * i + ch, path_cur[k], plen_cur[k], carried);
        } else if (n_nxt < 4) {
          HoSeg* o = &nxt[n_nxt];
          memcpy(o->d, dirs[ch], 12);
          memcpy(o->p, p_new, 12);
          o->w = ws[ch];
          o->face = f_new;
          memcpy(path_nxt[n_nxt], path_cur[k], (size_t)plen_cur[k]);
          plen_nxt[n_nxt] = plen_cur[k];
          if (plen_nxt[n_nxt] < HALO_MAX_HITS + 1) path_nxt[n_nxt][plen_nxt[n_nxt]++] = (uint8_t)g->face_number[f_new]; /* FillRayOtherInfo :645 */
          n_nxt++;
        }
      }
    }
    n_cur = n_nxt < 2 ? n_nxt : 2;
    for (int k = 0; k < n_cur; k++) {
      cur[k] = nxt[k];
      memcpy(path_cur[k], path_nxt[k], (size_t)plen_nxt[k]);
      plen_cur[k] = plen_nxt[k];
    }
  }
}

/* Entry sampling: InitRay_p_fid simulator.cpp:133-192 in its device form gen_root_kernel cu:1556-1597 */
static int sample_entry(HoStream* s, const HaloGeomTables* g, const float d_crystal[3], float p[3]) {
  if (g->tri_cnt == 0) { p[0] = p[1] = p[2] = 0.0f; (void)ho_pcg_uniform(s); return -1; }
  float proj_prob[HALO_MAX_TRIS];
  for (int t = 0; t < g->tri_cnt; ++t) {
    float dot = d_crystal[0] * g->tri_n[t * 3 + 0] + d_crystal[1] * g->tri_n[t * 3 + 1] + d_crystal[2] * g->tri_n[t * 3 + 2];
    proj_prob[t] = fmaxf(-dot * g->tri_area[t], 0.0f);
  }
  float u_cat = ho_pcg_uniform(s);
  uint32_t tri_id = ho_categorical_sample(proj_prob, (uint32_t)g->tri_cnt, u_cat);
  ho_sample_triangle(s, g->tri_v + tri_id * 9, p);
  return g->tri_face[tri_id];
}

static void run_ray(const HoCiCtx* c, HoSink* sink, uint32_t tid) {
  const HoBackend* b = c->b;
  float rot[9], d_crystal[3], p[3], w;
  int face;
  uint32_t wl_idx = 0;
  uint64_t carried = 0; /* colour mask inherited from earlier scattering layers */
  const HaloGeomTables* g = &c->shapes[(c->shape_cnt > 1) ? (tid / c->geom_clock) : 0];
  HoStream gate;
  {
    uint32_t lo = (uint32_t)(c->gate_base & 0xFFFFFFFFull), hi = (uint32_t)(c->gate_base >> 32);
    gate.seed = ho_pcg_seed_with_high(c->gate_seed, ho_pcg_advance_hi(lo, hi, tid));
    gate.global_idx = lo + tid;
    gate.slot = 0;
  }
  if (c->host) { /* host-ray injection: crystal-local, identity rotation (cpu_trace_backend.cpp:121-144) */
    static const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(rot, ident, sizeof(rot));
    memcpy(d_crystal, c->host->d + (size_t)tid * 3, 12);
    memcpy(p, c->host->p + (size_t)tid * 3, 12);
    w = c->host->w[tid];
    face = (int)c->host->tf[tid];
  } else if (c->layer == 0) { /* gen_root_kernel cu:1460-1616 */
    uint32_t lo = (uint32_t)(c->gen_base & 0xFFFFFFFFull), hi = (uint32_t)(c->gen_base >> 32);
    uint32_t gidx = lo + tid;
    uint32_t mixed = ho_pcg_seed_with_high(c->gen_seed, ho_pcg_advance_hi(lo, hi, tid));
    HoStream wls = {mixed ^ NONCE_WL, gidx, 0};
    wl_idx = (uint32_t)(ho_pcg_uniform(&wls) * (float)b->pool_size);
    if (wl_idx >= b->pool_size) wl_idx = b->pool_size - 1u;
    HoStream s = {mixed, gidx, 0};
    float lon, lat, roll;
    ho_sample_lat_lon_roll(&s, &c->gp, c->lut_theta, c->lut_cdf, c->lut_flip, &lon, &lat, &roll);
    ho_build_crystal_rotation_9(lon, lat, roll, rot);
    float d_world[3];
    ho_sample_sph_cap(&s, c->sun_lon, c->sun_lat, c->sun_half, d_world);
    ho_apply_inverse_mat9(rot, d_world, d_crystal);
    face = sample_entry(&s, g, d_crystal, p);
    w = b->pool[wl_idx].spd_weight;
    if (face < 0) w = 0.0f;
  } else { /* transit_multi_ms_kernel cu:1258-1403, reading through the Feistel gather (shuffle_cont_kernel cu:1633) */
    uint32_t lo = (uint32_t)(c->transit_base & 0xFFFFFFFFull), hi = (uint32_t)(c->transit_base >> 32);
    uint32_t gidx = lo + tid;
    uint32_t mixed = ho_pcg_seed_with_high(c->transit_seed, ho_pcg_advance_hi(lo, hi, tid));
    uint64_t pos = c->ci_start + tid;
    uint64_t src = c->shuffle ? ho_feistel_bijection((uint32_t)pos, (uint32_t)c->cont_in_n, c->shuffle_seed) : pos;
    const float* in = c->cont_in + src * HO_CONT_STRIDE;
    memcpy(&carried, in + 5, 8);
    HoStream s = {mixed, gidx, 0};
    float lon, lat, roll;
    ho_sample_lat_lon_roll(&s, &c->gp, c->lut_theta, c->lut_cdf, c->lut_flip, &lon, &lat, &roll);
    ho_build_crystal_rotation_9(lon, lat, roll, rot);
    ho_apply_inverse_mat9(rot, in, d_crystal);
    face = sample_entry(&s, g, d_crystal, p);
    w = in[3];
    wl_idx = (uint32_t)in[4];
    if (face < 0) w = 0.0f;
  }
  const HoWlEntry* wle = &b->pool[wl_idx];
  trace_root(c, sink, g, rot, wle->n_idx, wle, wl_idx, d_crystal, p, w, face, (uint32_t)(c->ci_start + tid), &gate, carried);
}

/* shape scalars: SamplePrismShapeScalars simulator.cpp:405-412 + SyncGroupSampler :361-393, host PCG stream */
static int crystal_is_deterministic(const HaloCrystal* cr) { /* IsDeterministic simulator.cpp:453-471 */
  int nh = cr->kind == HALO_CRYSTAL_PRISM ? 1 : 3;
  for (int i = 0; i < nh; i++) if (cr->height[i].type != HALO_DIST_NONE) return 0;
  for (int i = 0; i < 6; i++) if (cr->face_dist[i].type != HALO_DIST_NONE) return 0;
  return 1;
}

typedef struct { HoStream* s; int grp[9]; float val[9]; int cnt; } HoSync;
static float sync_draw(HoSync* y, int group, const HaloDist* d) {
  if (group == 0) return ho_pcg_get_dist(y->s, (uint32_t)d->type, d->center, d->spread);
  for (int i = 0; i < y->cnt; i++) if (y->grp[i] == group) return y->val[i];
  float v = ho_pcg_get_dist(y->s, (uint32_t)d->type, d->center, d->spread);
  y->grp[y->cnt] = group;
  y->val[y->cnt] = v;
  y->cnt++;
  return v;
}

/* The nine shape scalars of crystal instance `shape_index` in slot order [h0, h1, h2, d0..d5] (prism: h0 and d0..d5), heights folded
 * with fabs, face distances signed — SamplePrismShapeScalars / SamplePyramidShapeScalars (simulator.cpp:405-425) over the host PCG
 * stream of this repo (the reference draws from mt19937).  Draw order = the contract of test_crystal_sync_group_sampling.cpp:81-165:
 * heights first (upper, prism, lower), then d[0..5]; later members of a sync group reuse the group's first RAW draw. */
void ho_shape_scalars(const HaloCrystal* cr, uint32_t seed, uint64_t shape_index, float out9[9]) {
  HoStream s;
  uint32_t lo = (uint32_t)(shape_index & 0xFFFFFFFFull), hi = (uint32_t)(shape_index >> 32);
  s.seed = ho_pcg_seed_with_high(seed ^ NONCE_SHAPE_HOST, hi);
  s.global_idx = lo;
  s.slot = 0;
  HoSync y;
  memset(&y, 0, sizeof(y));
  y.s = &s;
  for (int i = 0; i < 9; i++) out9[i] = 0.0f;
  int nh = cr->kind == HALO_CRYSTAL_PRISM ? 1 : 3;
  for (int i = 0; i < nh; i++) out9[i] = fabsf(sync_draw(&y, cr->sync_group[i], &cr->height[i]));
  for (int i = 0; i < 6; i++) out9[3 + i] = sync_draw(&y, cr->sync_group[3 + i], &cr->face_dist[i]);
}

static void make_shape(const HoBackend* b, const HaloCrystal* cr, uint64_t shape_index, HaloGeomTables* out) {
  float sc[9];
  ho_shape_scalars(cr, b->seed, shape_index, sc);
  if (cr->kind == HALO_CRYSTAL_PRISM) ho_prism_geometry(sc[0], sc + 3, out);
  else ho_pyramid_geometry(cr->wedge_upper_deg, cr->wedge_lower_deg, sc[0], sc[1], sc[2], sc + 3, out); /* simulator.cpp:416-425 */
}

int ho_trace_layer(HoBackend* b, uint64_t count, const HaloHostRays* rays, HaloLayerStats* stats) {
  if (!b->in_session) return HALO_FATAL;
  int layer = b->layer_idx;
  if (layer >= b->scene.layer_count) return HALO_FATAL;
  const HaloLayer* L = &b->scene.layers[layer];
  uint64_t n = (layer == 0) ? count : b->cont_in_n;
  int final_layer = (layer == b->scene.layer_count - 1);
  /* continuation capacity for this layer's output: every root can emit ≤ max_hits continuations */
  if (!final_layer) {
    uint64_t need = n * (uint64_t)b->scene.max_hits;
    if (need > b->cont_cap) {
      free(b->cont);
      b->cont = (float*)malloc((size_t)(need ? need : 1) * HO_CONT_STRIDE * sizeof(float));
      b->cont_cap = need;
    }
  }
  b->cont_n = 0;
  if (b->capture) {
    uint64_t need = b->exit_n + n * (uint64_t)b->scene.max_hits;
    if (need > b->exit_cap) {
      b->exits = (HaloExitRecord*)realloc(b->exits, (size_t)(need ? need : 1) * sizeof(HaloExitRecord));
      b->exit_cap = need;
    }
  }
  float props[HALO_MAX_ENTRIES];
  uint64_t per_ci[HALO_MAX_ENTRIES];
  for (int ci = 0; ci < L->entry_count; ci++) props[ci] = L->entries[ci].proportion;
  ho_partition(props, L->entry_count, n, b->carry[layer], per_ci);
  if (rays && layer == 0) { /* host ingest: single population (cpu_trace_backend.cpp:121-127) */
    for (int ci = 0; ci < L->entry_count; ci++) per_ci[ci] = 0;
    per_ci[0] = n;
  }
  if (b->acc64 && !b->xyz64) b->xyz64 = (double*)calloc((size_t)b->acc_w * b->acc_h * 3, sizeof(double));
  if (b->acc64 && b->lanes && !b->lanes64) b->lanes64 = (double*)calloc((size_t)b->color_class_count * b->lanes_w * b->lanes_h, sizeof(double));
  HoSink total;
  memset(&total, 0, sizeof(total));
  uint64_t ci_start = 0;
  for (int ci = 0; ci < L->entry_count; ci++) {
    uint64_t n_ci = per_ci[ci];
    if (n_ci == 0) continue;
    const HaloEntry* E = &L->entries[ci];
    HoCiCtx* c = (HoCiCtx*)calloc(1, sizeof(HoCiCtx));
    c->b = b;
    c->layer = layer;
    c->ci = ci;
    c->final_layer = final_layer;
    c->prob = L->prob;
    c->max_hits = b->scene.max_hits;
    c->crystal_id = E->crystal_config_id;
    c->filter = (E->filter_id > 0 && E->filter_id <= b->filter_count) ? &b->filters[E->filter_id - 1] : NULL;
    c->color = (b->color_class_count > 0 && E->color_id > 0 && E->color_id <= b->color_set_count) ? &b->color_sets[E->color_id - 1] : NULL;
    c->axis = &E->axis;
    /* BuildTransitGpParams / BuildGenGpParams cuda_trace_backend.cu:342-399 */
    c->gp.lat_path = ho_select_lat_path(&E->axis);
    c->gp.lat_mean_rad = E->axis.latitude.center * HO_DEG2RAD;
    c->gp.lat_std_rad = E->axis.latitude.spread * HO_DEG2RAD;
    c->gp.lat_lut_n = (c->gp.lat_path == HO_LAT_LUT) ? HALO_LUT_NODES : 0u;
    c->gp.az_type = (uint32_t)E->axis.azimuth.type;
    c->gp.az_mean_rad = E->axis.azimuth.center * HO_DEG2RAD;
    c->gp.az_std_rad = E->axis.azimuth.spread * HO_DEG2RAD;
    c->gp.roll_type = (uint32_t)E->axis.roll.type;
    c->gp.roll_mean_rad = E->axis.roll.center * HO_DEG2RAD;
    c->gp.roll_std_rad = E->axis.roll.spread * HO_DEG2RAD;
    if (c->gp.lat_path == HO_LAT_LUT) ho_build_lat_lut(&E->axis.latitude, c->lut_theta, c->lut_cdf, c->lut_flip);
    c->sun_lon = (b->scene.sun_azimuth + 180.0f) * HO_DEG2RAD;
    c->sun_lat = -b->scene.sun_altitude * HO_DEG2RAD;
    c->sun_half = (b->scene.sun_diameter * 0.5f) * HO_DEG2RAD;
    c->geom_clock = (uint32_t)b->geom_clock;
    /* shape pool: one shape per geom_clock rays when stochastic (simulator.cpp:1244-1275), else one */
    /* HostRayBatch::crystal (trace_backend.hpp:230-239, cpu_trace_backend.cpp:121-144): injected rays may bring the crystal they were sampled
     * on — traced as it is, no MakeCrystal draw */
    const int host_crystal = rays && rays->crystal && layer == 0;
    int det = host_crystal || crystal_is_deterministic(&E->crystal);
    uint32_t P = det ? 1u : (uint32_t)((n_ci + c->geom_clock - 1) / c->geom_clock);
    HaloGeomTables* shapes = (HaloGeomTables*)malloc((size_t)P * sizeof(HaloGeomTables));
    for (uint32_t k = 0; k < P; k++) {
      if (host_crystal) shapes[k] = *rays->crystal;
      else make_shape(b, &E->crystal, det ? 0 : (b->shape_count + k), &shapes[k]);
    }
    if (!det) b->shape_count += P;
    c->shapes = shapes;
    c->shape_cnt = P;
    c->gen_seed = b->seed ^ NONCE_GEN;
    c->gate_seed = b->seed ^ NONCE_GATE;
    c->transit_seed = b->seed ^ NONCE_TRANSIT;
    c->gen_base = b->gen_count;
    c->gate_base = b->gate_count;
    c->transit_base = b->transit_count;
    c->cont_in = b->cont_in;
    c->cont_in_n = b->cont_in_n;
    c->ci_start = ci_start;
    c->shuffle = b->cont_shuffle;
    c->shuffle_seed = (b->seed ^ NONCE_SHUFFLE) ^ (uint32_t)layer; /* cu:4541 */
    c->host = (layer == 0) ? rays : NULL;
#ifdef _OPENMP
#pragma omp parallel num_threads(b->threads)
#endif
    {
      HoSink sink;
      memset(&sink, 0, sizeof(sink));
      sink.b = b;
      if (b->acc64) {
        sink.pc = (HoPixCache*)malloc(sizeof(HoPixCache));
        for (int k = 0; k < (1 << HO_PC_LOG2); k++) {
          sink.pc->pix[k] = -1;
          sink.pc->v[k][0] = sink.pc->v[k][1] = sink.pc->v[k][2] = 0.0;
        }
      }
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4096)
#endif
      for (int64_t t = 0; t < (int64_t)n_ci; t++) run_ray(c, &sink, (uint32_t)t);
      if (sink.pc) {
        for (uint32_t k = 0; k < (1u << HO_PC_LOG2); k++)
          if (sink.pc->pix[k] >= 0) pixcache_flush_slot(b, sink.pc, k);
        free(sink.pc);
      }
#ifdef _OPENMP
#pragma omp critical
#endif
      {
        total.landed += sink.landed;
        total.exit_count += sink.exit_count;
        total.exit_w_sum += sink.exit_w_sum;
        total.pixel_hits += sink.pixel_hits;
      }
    }
    if (layer == 0 && !rays) b->gen_count += n_ci;
    if (layer > 0) b->transit_count += n_ci;
    b->gate_count += n_ci;
    ci_start += n_ci;
    free(shapes);
    free(c);
  }
  b->landed += total.landed;
  if (b->cont_n > b->cont_cap) b->cont_n = b->cont_cap;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->root_count = n;
    stats->exit_count = total.exit_count;
    stats->exit_w_sum = total.exit_w_sum;
    stats->continuation_count = final_layer ? 0 : b->cont_n;
    stats->pixel_hits = total.pixel_hits;
  }
  return HALO_OK;
}

int ho_recombine(HoBackend* b, int shuffle, uint64_t* continuation_count) {
  if (!b->in_session) return HALO_FATAL;
  /* swap pools: this layer's output becomes the next layer's input; the permutation itself is applied
   * as a gather at read time (same result as shuffle_cont_kernel's out[tid] = in[feistel(tid)]). */
  float* t = b->cont_in;
  b->cont_in = b->cont;
  b->cont_in_n = b->cont_n;
  b->cont = t;
  b->cont_cap = 0; /* the old input buffer's capacity is unknown → force re-allocation */
  free(b->cont);
  b->cont = NULL;
  b->cont_n = 0;
  b->cont_shuffle = shuffle;
  b->layer_idx++;
  if (continuation_count) *continuation_count = b->cont_in_n;
  return HALO_OK;
}

uint64_t ho_continuation_dump(HoBackend* b, float* out5, uint64_t cap) {
  uint64_t n = b->cont_n < cap ? b->cont_n : cap;
  if (out5 && b->cont)
    for (uint64_t i = 0; i < n; i++) memcpy(out5 + i * 5, b->cont + i * HO_CONT_STRIDE, 5 * sizeof(float));
  return b->cont_n;
}

int ho_drain_exits(HoBackend* b, HaloExitRecord* out, uint64_t cap, uint64_t* count) {
  uint64_t n = b->exit_n < b->exit_cap ? b->exit_n : b->exit_cap;
  if (count) *count = n;
  if (out) memcpy(out, b->exits, (size_t)(n < cap ? n : cap) * sizeof(HaloExitRecord));
  b->exit_n = 0;
  return HALO_OK;
}

int ho_readback_xyz64(HoBackend* b, float* xyz, int width, int height, double* landed) {
  if (!b->xyz || width != b->acc_w || height != b->acc_h) return HALO_FATAL;
  materialise_acc64(b);
  size_t n = (size_t)width * height * 3;
  memcpy(xyz, b->xyz, n * sizeof(float));
  memset(b->xyz, 0, n * sizeof(float));
  if (landed) *landed = b->landed;
  b->landed = 0.0;
  return HALO_OK;
}

/* ======================================================================================== */
/* consumer: server/render.cpp:96-201,465-578; util/color_space.cpp; shared/accum_shared.h:70 */
/* ======================================================================================== */
static const float kWhitePointD65[3] = {0.95047f, 1.00000f, 1.08883f}; /* util/color_data.hpp:6 */
static const float kXyzToRgb[9] = {3.2404542f, -1.5371385f, -0.4985314f, -0.9692660f, 1.8760108f,
                                   0.0415560f, 0.0556434f,  -0.2040259f, 1.0572252f}; /* util/color_data.hpp:8-12 */

void ho_neumaier_add(float* sum, float* comp, float delta) { /* accum_shared.h:70-74 */
  float new_sum = *sum + delta;
  *comp += (fabsf(delta) < fabsf(*sum)) ? ((*sum - new_sum) + delta) : ((delta - new_sum) + *sum);
  *sum = new_sum;
}

void ho_gamut_clip_xyz(const float xyz[3], float clipped[3]) { /* color_space.cpp:13-39 */
  float gray[3], diff[3];
  for (int j = 0; j < 3; j++) gray[j] = kWhitePointD65[j] * xyz[1];
  float s = 1.0f;
  for (int j = 0; j < 3; j++) diff[j] = xyz[j] - gray[j];
  for (int j = 0; j < 3; j++) {
    float a = 0, b = 0;
    for (int k = 0; k < 3; k++) {
      a += -gray[k] * kXyzToRgb[j * 3 + k];
      b += diff[k] * kXyzToRgb[j * 3 + k];
    }
    if (a * b > 0 && a / b < s) s = a / b;
  }
  for (int j = 0; j < 3; j++) clipped[j] = diff[j] * s + gray[j];
}

void ho_xyz_to_linear_rgb(const float xyz[3], float rgb[3]) { /* color_space.cpp:41-49 */
  for (int j = 0; j < 3; j++) {
    float v = 0;
    for (int k = 0; k < 3; k++) v += xyz[k] * kXyzToRgb[j * 3 + k];
    rgb[j] = clampf(v, 0.0f, 1.0f);
  }
}

float ho_linear_to_srgb(float linear) { /* color_space.cpp:51-56 */
  if (linear < 0.0031308f) return linear * 12.92f;
  return 1.055f * powf(linear, 1.0f / 2.4f) - 0.055f;
}

int ho_consumer_fold(HoBackend* b) { /* ConsumeDeviceFused render.cpp:138-149 */
  if (!b->xyz) return HALO_FATAL;
  materialise_acc64(b);
  size_t n = (size_t)b->acc_w * b->acc_h * 3;
  if (!b->cons_sum || b->cons_w != b->acc_w || b->cons_h != b->acc_h) {
    free(b->cons_sum);
    free(b->cons_comp);
    b->cons_sum = (float*)calloc(n, sizeof(float));
    b->cons_comp = (float*)calloc(n, sizeof(float));
    b->cons_w = b->acc_w;
    b->cons_h = b->acc_h;
    b->total_intensity = 0.0f;
  }
  for (size_t i = 0; i < n; i++) ho_neumaier_add(&b->cons_sum[i], &b->cons_comp[i], b->xyz[i]);
  memset(b->xyz, 0, n * sizeof(float));
  b->total_intensity += (float)b->landed;
  b->landed = 0.0;
  return HALO_OK;
}

/* RenderConsumer::ConsumeDeviceFused(const SimData&) render.cpp:138-201 for a drained image the caller holds */
int ho_consumer_consume(HoBackend* b, const float* xyz, int width, int height, float landed, const float* lanes, int class_count) {
  if (!xyz || width <= 0 || height <= 0) return HALO_FATAL;
  size_t n = (size_t)width * height * 3;
  if (b->cons_sum && (b->cons_w != width || b->cons_h != height)) return HALO_FATAL;
  if (!b->cons_sum) {
    b->cons_sum = (float*)calloc(n, sizeof(float));
    b->cons_comp = (float*)calloc(n, sizeof(float));
    b->cons_w = width;
    b->cons_h = height;
    b->total_intensity = 0.0f;
  }
  for (size_t i = 0; i < n; i++) ho_neumaier_add(&b->cons_sum[i], &b->cons_comp[i], xyz[i]);
  b->total_intensity += landed;
  if (lanes && class_count > 0) {
    if (class_count != b->color_class_count) return HALO_FATAL;
    size_t nl = (size_t)class_count * width * height;
    if (!b->lanes || b->lanes_w != width || b->lanes_h != height) {
      free(b->lanes);
      b->lanes = (float*)calloc(nl, sizeof(float));
      b->lanes_w = width;
      b->lanes_h = height;
    }
    for (size_t i = 0; i < nl; i++) b->lanes[i] += lanes[i];
  }
  return HALO_OK;
}

int ho_consumer_snapshot(HoBackend* b, const HaloDisplay* dsp, uint8_t* rgb_out, float* xyz_out, double* total_intensity) {
  if (!b->cons_sum) return HALO_FATAL;
  int total_pix = b->cons_w * b->cons_h;
  float snapshot_intensity = b->total_intensity;
  if (total_intensity) *total_intensity = (double)snapshot_intensity;
  float scale = (total_pix <= 0 || snapshot_intensity <= 0.0f) ? 0.0f : dsp->intensity_factor * 0.08f * total_pix / snapshot_intensity; /* :96-102 */
  int use_real_color = dsp->ray_color[0] < 0;
  for (int i = 0; i < total_pix; i++) {
    float xyz[3], rgb[3];
    for (int j = 0; j < 3; j++) {
      float raw = b->cons_sum[i * 3 + j] + b->cons_comp[i * 3 + j]; /* PrepareSnapshot :480-482 */
      if (xyz_out) xyz_out[i * 3 + j] = raw;
      xyz[j] = raw * scale;
    }
    if (!rgb_out) continue;
    if (scale == 0.0f) { rgb_out[i * 3] = rgb_out[i * 3 + 1] = rgb_out[i * 3 + 2] = 0; continue; }
    if (use_real_color) {
      float clipped[3];
      ho_gamut_clip_xyz(xyz, clipped);
      ho_xyz_to_linear_rgb(clipped, rgb);
    } else {
      float gray[3];
      for (int j = 0; j < 3; j++) gray[j] = kWhitePointD65[j] * xyz[1];
      for (int j = 0; j < 3; j++) {
        float v = 0;
        for (int k = 0; k < 3; k++) v += gray[k] * kXyzToRgb[j * 3 + k];
        rgb[j] = v * dsp->ray_color[j];
      }
    }
    for (int j = 0; j < 3; j++) {
      rgb[j] += dsp->background[j];
      rgb[j] = clampf(rgb[j], 0.0f, 1.0f);
      rgb[j] = ho_linear_to_srgb(rgb[j]);
      rgb_out[i * 3 + j] = (uint8_t)(rgb[j] * 255);
    }
  }
  return HALO_OK;
}

/* ---- display-side composite of the class lanes ------------------------------------------------------------- */
/* server/component_compositor.cpp, restated: GatherActiveClasses :24-54, the three per-pixel modes :56-114, ParseCompositeMode :118-134,
 * ComputeParticipatingP99Y :138-163, CompositeColorClassesLinear :180-285, LinearRgbToSrgbU8 :292-300; RenderConsumer::
 * ParticipatingExposureScale render.cpp:120-135. */
float ho_participating_exposure_scale(float intensity_factor, float snapshot_intensity, float p99) {
  if (p99 <= 0.0f || snapshot_intensity <= 0.0f) return 0.0f;
  const float target_srgb = 135.0f / 255.0f;
  const float target_linear = target_srgb <= 0.04045f ? target_srgb / 12.92f : powf((target_srgb + 0.055f) / 1.055f, 2.4f);
  if (target_linear <= 0.0f) return 0.0f;
  return intensity_factor * target_linear / p99;
}
int ho_parse_composite_mode(const char* mode) {
  if (mode && strcmp(mode, "dominant") == 0) return HALO_COMPOSITE_DOMINANT;
  if (mode && strcmp(mode, "additive") == 0) return HALO_COMPOSITE_ADDITIVE;
  return HALO_COMPOSITE_PAINTER; /* "painter" and every unknown string */
}
static int cmp_float_asc(const void* a, const void* b) {
  float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}
int ho_composite(const float* lanes, int width, int height, int class_count, uint64_t referenced_mask, float total_intensity,
                 const HaloComposite* spec, float* out_rgb, uint8_t* out_srgb, float* p99_out, int32_t* produced) {
  if (!spec || !produced || class_count != spec->class_count || class_count > HALO_COLOR_MAX_CLASSES) return HALO_FATAL;
  *produced = 0;
  if (referenced_mask == 0) return HALO_OK;
  size_t n = (size_t)width * (size_t)height;
  if (out_rgb) memset(out_rgb, 0, n * 3 * sizeof(float));
  if (n == 0) {
    if (p99_out) *p99_out = 0.0f;
    *produced = 1;
    return HALO_OK;
  }
  /* active classes in draw order: stable by z_order (insertion sort keeps equal keys in list order) */
  int order[HALO_COLOR_MAX_CLASSES], active[HALO_COLOR_MAX_CLASSES], na = 0, any_solo = 0;
  for (int c = 0; c < class_count; c++) {
    order[c] = c;
    if (spec->classes[c].solo) any_solo = 1;
  }
  for (int i = 1; i < class_count; i++) {
    int v = order[i], j = i - 1;
    while (j >= 0 && spec->classes[order[j]].z_order > spec->classes[v].z_order) {
      order[j + 1] = order[j];
      j--;
    }
    order[j + 1] = v;
  }
  for (int i = 0; i < class_count; i++) {
    const HaloCompositeClass* k = &spec->classes[order[i]];
    if (any_solo ? k->solo : k->visible) active[na++] = order[i];
  }
  if (na == 0) {
    if (out_srgb) memset(out_srgb, 0, n * 3);
    if (p99_out) *p99_out = 0.0f;
    *produced = 1;
    return HALO_OK;
  }
  /* participating P99: the order statistic nth_element picks */
  float p99 = 0.0f;
  {
    float* y = (float*)malloc((size_t)na * n * sizeof(float));
    size_t m = 0;
    for (int a = 0; a < na; a++)
      for (size_t p = 0; p < n; p++) {
        float v = lanes[(size_t)active[a] * n + p];
        if (v > 0.0f) y[m++] = v;
      }
    if (m > 0) {
      size_t idx = (size_t)((float)m * 0.99f);
      if (idx >= m) idx = m - 1;
      qsort(y, m, sizeof(float), cmp_float_asc);
      p99 = y[idx];
    }
    free(y);
  }
  const float A = ho_participating_exposure_scale(spec->intensity_factor, total_intensity, p99);
  if (A <= 0.0f) {
    if (p99_out) *p99_out = p99;
    if (out_rgb) memset(out_rgb, 0, n * 3 * sizeof(float)); /* the caller's buffer was "assigned" zeros before the early return (:189) */
    return HALO_OK;
  }
  const float s = A * spec->display_exposure_scale;
  for (size_t p = 0; p < n; p++) {
    float out[3] = {0.0f, 0.0f, 0.0f};
    if (spec->mode == HALO_COMPOSITE_DOMINANT) {
      int best = -1;
      float best_ey = 0.0f;
      for (int a = 0; a < na; a++) {
        float ey = lanes[(size_t)active[a] * n + p] * s;
        if (ey > best_ey) {
          best_ey = ey;
          best = a;
        }
      }
      if (best >= 0)
        for (int j = 0; j < 3; j++) out[j] = spec->classes[active[best]].color[j] * best_ey;
    } else if (spec->mode == HALO_COMPOSITE_ADDITIVE) {
      float acc[3] = {0.0f, 0.0f, 0.0f};
      for (int a = 0; a < na; a++) {
        float ey = lanes[(size_t)active[a] * n + p] * s;
        if (ey <= 0.0f) continue;
        for (int j = 0; j < 3; j++) acc[j] += spec->classes[active[a]].color[j] * ey;
      }
      for (int j = 0; j < 3; j++) out[j] = acc[j] < 0.0f ? 0.0f : (acc[j] > 1.0f ? 1.0f : acc[j]);
    } else {
      float T = 1.0f;
      for (int a = 0; a < na && T > 0.0f; a++) {
        float alpha = lanes[(size_t)active[a] * n + p] * A;
        if (alpha > 1.0f) alpha = 1.0f;
        if (alpha <= 0.0f) continue;
        for (int j = 0; j < 3; j++) out[j] += T * alpha * spec->classes[active[a]].color[j];
        T *= (1.0f - alpha);
      }
      for (int j = 0; j < 3; j++) {
        float v = out[j] * spec->display_exposure_scale;
        out[j] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
      }
    }
    if (out_rgb)
      for (int j = 0; j < 3; j++) out_rgb[p * 3 + j] = out[j];
    if (out_srgb)
      for (int j = 0; j < 3; j++) {
        float c = out[j] < 0.0f ? 0.0f : (out[j] > 1.0f ? 1.0f : out[j]);
        out_srgb[p * 3 + j] = (uint8_t)(ho_linear_to_srgb(c) * 255.0f);
      }
  }
  if (p99_out) *p99_out = p99;
  *produced = 1;
  return HALO_OK;
}
