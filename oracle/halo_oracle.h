/*
 * halo_oracle.h — CPU restatement (plain C) of the Lumice trace hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ice_halo_sim_amd/ may include, link or call this; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.  The
 * restatement is pinned three ways (tests/test_oracle_*.py):
 *   1. against oracle/_ref/libref_shared.so — the reference's own single-source math headers
 *      (src/core/shared/ *.h, src/util/color_data.hpp, test/support/exact_prism_oracle.hpp) compiled
 *      where they lie, no stand-ins — bit-exact on random inputs (this container only);
 *   2. against the golden vectors the reference's own tests hold (tests/golden/ref_test_vectors.json);
 *   3. against fixtures generated from (1) and committed (tests/golden/ref_shared_fixture.npz).
 * NOT pinned by an executable reference: BuildLatLut, closed-form prism/pyramid geometry,
 * BuildProjParams, PartitionCrystalRayNum, IceRefractiveIndex (their TUs need nlohmann>=3.4 / spdlog,
 * absent from the image → unbuildable here); those are pinned by (2) only.  The legacy
 * `Simulator` image path is unbuildable here for the same reason: end-to-end image parity against
 * the reference binary is UNPINNED.
 */
#ifndef HALO_ORACLE_H_
#define HALO_ORACLE_H_

#include <stdint.h>
#include "../include/halo_trace.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- src/core/shared/pcg_shared.h ------------------------------------------------------- */
typedef struct HoStream {
  uint32_t seed, global_idx, slot;
} HoStream;

/* GenRootKernelParams subset actually read by the samplers (pcg_shared.h:150-189). */
typedef struct HoGenParams {
  uint32_t lat_path;
  float lat_mean_rad, lat_std_rad;
  uint32_t lat_lut_n;
  uint32_t az_type;
  float az_mean_rad, az_std_rad;
  uint32_t roll_type;
  float roll_mean_rad, roll_std_rad;
} HoGenParams;

uint32_t ho_pcg_hash(uint32_t x);
float ho_u01_from_hash(uint32_t h);
uint32_t ho_pcg_advance_hi(uint32_t base_lo, uint32_t base_hi, uint32_t tid);
uint32_t ho_pcg_seed_with_high(uint32_t seed, uint32_t hi);
float ho_pcg_uniform(HoStream* s);
float ho_pcg_gaussian(HoStream* s);
float ho_pcg_get_dist(HoStream* s, uint32_t dtype, float mean, float std_val);
void ho_normalize_latitude(float phi, float* phi_out, int* flip);
float ho_invert_lat_lut(float xi, const float* theta_nodes, const float* cdf_nodes, uint32_t n_nodes);
uint32_t ho_lat_lut_bin(float theta, const float* theta_nodes, uint32_t n_nodes);
void ho_sample_lat_lon_roll(HoStream* s, const HoGenParams* gp, const float* lut_theta, const float* lut_cdf,
                            const float* lut_flip, float* lon, float* lat, float* roll);
void ho_build_crystal_rotation_9(float lon, float lat, float roll, float* mat9);
void ho_apply_inverse_mat9(const float* mat9, const float* d_world, float* d_crystal);
void ho_apply_mat9(const float* mat9, const float* v, float* out);
void ho_sample_triangle(HoStream* s, const float* vtx9, float* out_p);
void ho_sample_sph_cap(HoStream* s, float lon, float lat, float half_angle, float* out_d);
uint32_t ho_feistel_bijection(uint32_t i, uint32_t n, uint32_t seed);
uint32_t ho_categorical_sample(const float* weights, uint32_t n, float u_in);

/* ---- optics / traversal ---------------------------------------------------------------- */
float ho_reflect_ratio(float delta, float rr);                                   /* optics_shared.h:17 */
float ho_slab_face_t(const float d[3], const float p[3], const float n[3], float fd); /* traversal_shared.h:61 */
double ho_ice_refractive_index(double wavelength_nm);                            /* optics.cpp:180 */

/* ---- projection ------------------------------------------------------------------------ */
typedef struct HoProjParams { /* lm_proj::ProjParams, projection_shared.h:106-118 (76 bytes) */
  int32_t proj_type, img_w, img_h, visible_range, lens_shift_x, lens_shift_y;
  float scale, az0, r_scale, max_abs_dz;
  float rot[9];
} HoProjParams;
typedef struct HoPixelHit {
  int32_t px, py, bump_landed;
} HoPixelHit;
typedef struct HoProjResult {
  HoPixelHit hits[2];
  int32_t count;
} HoProjResult;
void ho_build_proj_params(const HaloRender* cfg, HoProjParams* out); /* lens_proj_build.hpp:79 */
HoProjResult ho_project_exit_to_pixel(const HoProjParams* p, float wx, float wy, float wz); /* projection_shared.h:196 */

/* ---- host tables ------------------------------------------------------------------------ */
void ho_build_lat_lut(const HaloDist* lat, float* theta, float* cdf, float* flip);  /* lat_lut.cpp:74 */
uint32_t ho_select_lat_path(const HaloAxis* axis);                                   /* lat_path_selection.hpp:62 */
void ho_prism_geometry(float h, const float dist[6], HaloGeomTables* out);           /* geo3d_closedform.cpp:1318, crystal.cpp:109-347 */
void ho_pyramid_geometry(float wedge_u, float wedge_l, float h1, float h2, float h3, const float dist[6], HaloGeomTables* out); /* crystal.cpp:379-426 */
int ho_pyramid_face_mask(float wedge_u, float wedge_l, float h1, float h2, float h3, const float dist[6], int* vtx_cnt); /* test hook: present-face bitmask */
int ho_prism_corner_ring(float h, const float dist[6], float* cx, float* cy, int* face_present8); /* test hook */
void ho_partition(const float* proportions, int n, uint64_t ray_num, double* carry, uint64_t* out); /* simulator.cpp:519 */
float ho_illuminant_spd(int illuminant, float wavelength_nm);                        /* util/illuminant.cpp:113 */
void ho_cmf(float wl, float* x, float* y, float* z);                                 /* wl_pool.hpp:49, color_util.hpp:29 */

/* ---- whole path -------------------------------------------------------------------------- */
typedef struct HoBackend HoBackend; /* mirrors the C ABI state machine of include/halo_trace.h */
HoBackend* ho_create(uint32_t seed);
void ho_destroy(HoBackend* b);
int ho_set_option(HoBackend* b, const char* key, int64_t value); /* capture_exits, geom_clock, rank, threads */
int ho_begin(HoBackend* b, const HaloScene* scene, const HaloRender* render, const HaloWl* wl, uint64_t ray_num_hint);
int ho_trace_layer(HoBackend* b, uint64_t count, const HaloHostRays* rays, HaloLayerStats* stats);
int ho_recombine(HoBackend* b, int shuffle, uint64_t* continuation_count);
int ho_drain_exits(HoBackend* b, HaloExitRecord* out, uint64_t cap, uint64_t* count);
int ho_end(HoBackend* b);
int ho_readback_xyz64(HoBackend* b, float* xyz, int width, int height, double* landed_weight);
/* emit-gate filters (core/filter_spec.cpp, shared/filter_shared.h, crystal.cpp:514-600,708-730) */
void ho_reduce_raypath(const uint8_t* rp, int n, int symmetry, int sigma_a, int d_applicable, uint8_t* out);
int ho_compute_sigma_a(float roll_mean_deg);
int ho_is_d_applicable(const HaloAxis* axis);
int ho_filter_check(const HaloFilter* f, const HaloAxis* axis, const uint8_t* path, int len, const float dir_world[3], int crystal_id);
uint64_t ho_color_mask(const HaloColorSet* cs, const HaloAxis* axis, const uint8_t* path, int len, const float dir_world[3], int crystal_id, uint64_t carried);
void ho_shape_scalars(const HaloCrystal* cr, uint32_t seed, uint64_t shape_index, float out9[9]); /* simulator.cpp:405-425, :361-393 */
int ho_set_filters(HoBackend* b, const HaloFilter* filters, int32_t count);
int ho_set_color(HoBackend* b, const HaloColorSet* sets, int32_t n_sets, const HaloColorClass* classes, int32_t n_classes);
int ho_readback_class_lanes(HoBackend* b, float* lanes, int width, int height, int class_count);
/* consumer (server/render.cpp:96-201,465-578; util/color_space.cpp) */
void ho_neumaier_add(float* sum, float* comp, float delta);
void ho_gamut_clip_xyz(const float xyz[3], float clipped[3]);
void ho_xyz_to_linear_rgb(const float xyz[3], float rgb[3]);
float ho_linear_to_srgb(float linear);
int ho_consumer_fold(HoBackend* b);
int ho_consumer_consume(HoBackend* b, const float* xyz, int width, int height, float landed, const float* lanes, int class_count);
int ho_consumer_snapshot(HoBackend* b, const HaloDisplay* display, uint8_t* rgb_out, float* xyz_out, double* total_intensity);
/* display-side composite of the class lanes (server/component_compositor.cpp:24-303; ParticipatingExposureScale render.cpp:120-135).
 * Stateless: `lanes` = class_count x W x H floats as ReadbackClassLanes hands them out, referenced_mask = OR of the classes' member
 * bits, total_intensity = the consumer's snapshot intensity.  *produced = what CompositeColorClassesLinear returns. */
float ho_participating_exposure_scale(float intensity_factor, float snapshot_intensity, float participating_p99_y);
int ho_parse_composite_mode(const char* mode);
int ho_composite(const float* lanes, int width, int height, int class_count, uint64_t referenced_mask, float total_intensity,
                 const HaloComposite* spec, float* linear_rgb_out, uint8_t* srgb_out, float* participating_p99_y, int32_t* produced);
/* continuation pool access for set-parity tests: n x {dx,dy,dz,w,wl_idx(as float)} */
uint64_t ho_continuation_dump(HoBackend* b, float* out5, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
