// ref_shared_driver.cpp — extern "C" handles onto the REFERENCE's own single-source math headers.
//
// TEST INFRASTRUCTURE, this container only.  Compiled by oracle/Makefile straight from the files
// where they lie under /root/reference (nothing is copied, nothing is stubbed):
//   src/core/shared/{lm_shims,pcg_shared,optics_shared,traversal_shared,projection_shared,accum_shared}.h
//   src/core/color_util.hpp + src/util/color_data.hpp      (std headers only)
//   test/support/exact_prism_oracle.hpp                     (std headers only)
//   src/util/color_space.cpp (+ color_space.hpp, color_data.hpp)   (std headers only; compiled as a second TU)
// Output: oracle/_ref/libref_shared.so (git-ignored).  Used to validate oracle/halo_oracle.c
// function-by-function and to generate tests/golden/ref_shared_fixture.npz.
//
// Everything else on the path (lat_lut.cpp, geo3d_closedform.cpp, crystal.cpp, optics.cpp,
// lens_proj_build.hpp, simulator.cpp) includes core/math.hpp → nlohmann/json.hpp (needs >= 3.4 for
// NLOHMANN_JSON_SERIALIZE_ENUM; the image holds 3.1.1 at a foreign path) and/or util/logger.hpp →
// spdlog (absent): unbuildable here without stand-ins, so not built.
#include <cstdint>
#include <cstring>

#include "core/shared/accum_shared.h"
#include "core/shared/optics_shared.h"
#include "core/shared/pcg_shared.h"
#include "core/shared/projection_shared.h"
#include "core/shared/traversal_shared.h"
#include "core/color_util.hpp"
#include "util/color_space.hpp"
#include "support/exact_prism_oracle.hpp"

using lm_pcg::PcgStream;

extern "C" {

uint32_t ref_pcg_hash(uint32_t x) { return lm_pcg::pcg_hash(x); }
float ref_u01_from_hash(uint32_t h) { return lm_pcg::u01_from_hash(h); }
uint32_t ref_pcg_advance_hi(uint32_t lo, uint32_t hi, uint32_t tid) { return lm_pcg::pcg_advance_hi(lo, hi, tid); }
uint32_t ref_pcg_seed_with_high(uint32_t s, uint32_t hi) { return lm_pcg::pcg_seed_with_high(s, hi); }

// streams are passed as uint32[3] = {seed, global_idx, slot}
float ref_pcg_uniform(uint32_t* s3) {
  PcgStream s{s3[0], s3[1], s3[2]};
  float r = lm_pcg::pcg_uniform(s);
  s3[2] = s.slot;
  return r;
}
float ref_pcg_gaussian(uint32_t* s3) {
  PcgStream s{s3[0], s3[1], s3[2]};
  float r = lm_pcg::pcg_gaussian(s);
  s3[2] = s.slot;
  return r;
}
float ref_pcg_get_dist(uint32_t* s3, uint32_t dtype, float mean, float std_val) {
  PcgStream s{s3[0], s3[1], s3[2]};
  float r = lm_pcg::pcg_get_dist(s, dtype, mean, std_val);
  s3[2] = s.slot;
  return r;
}
void ref_normalize_latitude(float phi, float* phi_out, int* flip) {
  bool f = false;
  lm_pcg::normalize_latitude(phi, *phi_out, f);
  *flip = f ? 1 : 0;
}
float ref_invert_lat_lut(float xi, const float* th, const float* cdf, uint32_t n) { return lm_pcg::invert_lat_lut(xi, th, cdf, n); }
uint32_t ref_lat_lut_bin(float theta, const float* th, uint32_t n) { return lm_pcg::lat_lut_bin(theta, th, n); }

// gp10 = {lat_path, lat_mean, lat_std, lat_lut_n, az_type, az_mean, az_std, roll_type, roll_mean, roll_std} (u32/f32 bit-mixed)
void ref_sample_lat_lon_roll(uint32_t* s3, const void* gp10, const float* th, const float* cdf, const float* flip,
                             float* lon, float* lat, float* roll) {
  const uint32_t* u = static_cast<const uint32_t*>(gp10);
  const float* f = static_cast<const float*>(gp10);
  lm_pcg::GenRootKernelParams gp{};
  gp.lat_path = u[0];
  gp.lat_mean_rad = f[1];
  gp.lat_std_rad = f[2];
  gp.lat_lut_n = u[3];
  gp.az_type = u[4];
  gp.az_mean_rad = f[5];
  gp.az_std_rad = f[6];
  gp.roll_type = u[7];
  gp.roll_mean_rad = f[8];
  gp.roll_std_rad = f[9];
  PcgStream s{s3[0], s3[1], s3[2]};
  lm_pcg::sample_lat_lon_roll(s, gp, th, cdf, flip, *lon, *lat, *roll, nullptr);
  s3[2] = s.slot;
}
void ref_build_crystal_rotation_9(float lon, float lat, float roll, float* m) { lm_pcg::build_crystal_rotation_9(lon, lat, roll, m); }
void ref_apply_inverse_mat9(const float* m, const float* d, float* o) { lm_pcg::apply_inverse_mat9(m, d, o); }
void ref_sample_triangle(uint32_t* s3, const float* v9, float* p) {
  PcgStream s{s3[0], s3[1], s3[2]};
  lm_pcg::sample_triangle(s, v9, p);
  s3[2] = s.slot;
}
void ref_sample_sph_cap(uint32_t* s3, float lon, float lat, float half, float* d) {
  PcgStream s{s3[0], s3[1], s3[2]};
  lm_pcg::sample_sph_cap(s, lon, lat, half, d);
  s3[2] = s.slot;
}
uint32_t ref_feistel_bijection(uint32_t i, uint32_t n, uint32_t seed) { return lm_pcg::feistel_bijection(i, n, seed); }
uint32_t ref_categorical_sample(const float* w, uint32_t n, float u) { return lm_pcg::categorical_sample(w, n, u); }
uint32_t ref_wl_stream_seed(uint32_t mixed) { return lm_pcg::BuildWlStream(mixed, 0u).seed; }
uint32_t ref_geom_shape_stream_seed(uint32_t mixed) { return lm_pcg::BuildGeomShapeStream(mixed, 0u).seed; }

float ref_reflect_ratio(float delta, float rr) { return lm_optics::GetReflectRatio(delta, rr); }
float ref_slab_face_t(const float* d, const float* p, const float* n, float fd) {
  return lm_traversal::SlabFaceT(d[0], d[1], d[2], p[0], p[1], p[2], n[0], n[1], n[2], fd);
}

// ProjParams is passed as the raw 76-byte POD (same layout as lm_proj::ProjParams).
int ref_proj_params_size() { return static_cast<int>(sizeof(lm_proj::ProjParams)); }
// out7 = {count, px0, py0, landed0, px1, py1, landed1}
void ref_project_exit_to_pixel(const void* pp, float wx, float wy, float wz, int* out7) {
  lm_proj::ProjParams p;
  std::memcpy(&p, pp, sizeof(p));
  lm_proj::ProjResult r = lm_proj::ProjectExitToPixel(p, wx, wy, wz);
  out7[0] = r.count;
  for (int k = 0; k < 2; k++) {
    out7[1 + 3 * k] = (k < r.count) ? r.hits[k].px : 0;
    out7[2 + 3 * k] = (k < r.count) ? r.hits[k].py : 0;
    out7[3 + 3 * k] = (k < r.count) ? (r.hits[k].bump_landed ? 1 : 0) : 0;
  }
}
void ref_accum_xyz_to_pixel(float* buf, uint32_t pix, float cx, float cy, float cz, float w) { AccumXyzToPixel(buf, pix, cx, cy, cz, w); }
// color_util.hpp:29 SpectrumToXyz for one sample into xyz[3]
void ref_spectrum_to_xyz(float wl, float v, float* xyz3) { lumice::SpectrumToXyz(wl, &v, nullptr, xyz3, 1); }

// util/color_space.cpp + accum_shared.h NeumaierAdd
void ref_gamut_clip_xyz(const float* xyz, float* out) { lumice::GamutClipXyz(xyz, out); }
void ref_xyz_to_linear_rgb(const float* xyz, float* out) { lumice::XyzToLinearRgb(xyz, out); }
float ref_linear_to_srgb(float v) { return lumice::LinearToSrgb(v); }
void ref_xyz_to_srgb_u8(const float* xyz, unsigned char* out, int n, float scale) { lumice::XyzToSrgbUint8(xyz, out, n, scale); }
void ref_neumaier_add(float* sum, float* comp, float delta) { NeumaierAdd(*sum, *comp, delta); }

// test/support/exact_prism_oracle.hpp — out8 = {corner_count, refused, mask0..mask5}
void ref_exact_prism(const float* dist6, int* out8) {
  auto v = lumice::test_support::ExactPrism(dist6);
  out8[0] = v.corner_count;
  out8[1] = v.refused ? 1 : 0;
  for (int i = 0; i < 6; i++) out8[2 + i] = v.face_corner_mask[i];
}

}  // extern "C"
