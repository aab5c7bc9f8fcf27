/*
 * halo_trace.h — C ABI of the MI355X-native ice-halo trace backend (libhalo_hip.so).
 *
 * This is the drop-in boundary for ONE path of Lumice (LoveDaisy/ice_halo_sim): the per-ray trace
 * hot path behind `lumice::TraceBackend` (reference: src/core/backend/trace_backend.hpp:367-641).
 * The reference has no C-ABI plugin loader (backends are C++ classes compiled in and picked by
 * `CreateBackend`, src/core/simulator.cpp:854-919); this header is the stable C boundary we put
 * UNDER a thin C++ adapter (`ice_halo_sim_amd/csrc/hip_trace_backend.hpp`).
 *
 * Each entry point names the reference virtual it replaces.  Plain pointers and sizes only.
 * All functions return HALO_OK (0), HALO_UNAVAILABLE (1 → adapter throws BackendUnavailableError,
 * trace_backend.hpp:140-158) or HALO_FATAL (2).  One handle = one backend instance; a handle is
 * single-threaded (trace_backend.hpp:114-116, simulator.cpp:970-979).
 *
 * Everything crossing this boundary is WORLD-space (trace_backend.hpp:71-89) except injected golden
 * rays, which follow the reference's host-ingest convention: crystal-local, identity rotation
 * (src/core/backend/cpu_trace_backend.cpp:121-144).
 */
#ifndef HALO_TRACE_H_
#define HALO_TRACE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HALO_ABI_VERSION 6   /* 2: HaloFilter holds up to 64 OR-clauses / 64 terms (was 8 / 16); 3: halo_last_route, piecewise
                                halo_drain_exits, option "shuffle_chunk", exit records carry full 64-face paths; 4: HaloRouteInfo
                                names the kernel mode (5 modes) and its specialisation, option "filter_fast"; 5: halo_consumer_composite /
                                halo_consumer_load_lanes (additive: nothing that existed changed); 6: HaloHostRays::crystal (the
                                seam's HostRayBatch::crystal), halo_consumer_consume (ConsumeDeviceFused of a drained image the CALLER holds) */

enum { HALO_OK = 0, HALO_UNAVAILABLE = 1, HALO_FATAL = 2 };

/* Compile-time caps — reference src/core/def.hpp:23-31 (kMaxMsNum, kMaxHits, kMaxCrystalNum). */
#define HALO_MAX_LAYERS 4
#define HALO_MAX_ENTRIES 16
#define HALO_MAX_HITS 64
#define HALO_MAX_FACES 20        /* CrystalGeom face slots, src/core/crystal.hpp:78 */
#define HALO_MAX_FACE_VTX 12     /* kCrystalGeomMaxVtxPerFace */
#define HALO_MAX_TRIS 64         /* lm_pcg::kMaxTriPerKernel, src/core/shared/pcg_shared.h:71 */
#define HALO_LUT_NODES 257       /* LatLut::kNodes, src/core/lat_lut.hpp:31 */
#define HALO_WL_POOL_MAX 255     /* kWlPoolSizeMax, src/core/backend/wl_pool.hpp:41 */
#define HALO_PATH_CAP 64         /* face numbers kept per exit record = the reference's ExitFaceSeq::kCap (exit_seam.hpp:22) */

/* DistributionType — src/core/math.hpp:123-130 (values are wire values, pcg_shared.h:64-69). */
enum {
  HALO_DIST_NONE = 0,
  HALO_DIST_UNIFORM = 1,   /* spread = FULL range */
  HALO_DIST_GAUSS = 2,
  HALO_DIST_ZIGZAG = 3,
  HALO_DIST_LAPLACIAN = 4,
  HALO_DIST_GAUSS_LEGACY = 5
};

/* Distribution{type, center, spread} — src/core/math.hpp:165-190. */
typedef struct HaloDist {
  int32_t type;
  float center;
  float spread;
} HaloDist;

/* AxisDistribution — src/core/math.hpp:271-310.  Degrees.  `latitude` is the INTERNAL latitude
 * (JSON `zenith` = 90 - latitude, src/core/math.cpp:679-714). */
typedef struct HaloAxis {
  HaloDist azimuth;
  HaloDist latitude;
  HaloDist roll;
} HaloAxis;

enum { HALO_CRYSTAL_PRISM = 0, HALO_CRYSTAL_PYRAMID = 1 };

/* PrismCrystalParam / PyramidCrystalParam — src/config/crystal_config.hpp.
 * Shape-scalar slot order = RNG draw order (src/core/simulator.cpp:405-425):
 *   prism  : height[0]=h, face_dist[0..5]
 *   pyramid: height[0]=upper_h, height[1]=prism_h, height[2]=lower_h, face_dist[0..5]
 * sync_group[k] (0 = independent) indexes [h0,h1,h2,d0..d5]; members of one group share one draw
 * (src/core/simulator.cpp:344-393). */
typedef struct HaloCrystal {
  int32_t kind;
  HaloDist height[3];
  HaloDist face_dist[6];
  int32_t sync_group[9];
  float wedge_upper_deg; /* pyramid only */
  float wedge_lower_deg;
} HaloCrystal;

/* One scattering-layer entry: MsInfo::setting_[ci] — src/config/proj_config.hpp:27-38. */
typedef struct HaloEntry {
  HaloCrystal crystal;
  HaloAxis axis;
  float proportion;
  int32_t crystal_config_id;
  int32_t filter_id;  /* 0 = no filter (pass-all); k > 0 = filters[k-1] of the table given to halo_set_filters */
  int32_t color_id;   /* 0 = no raypath-colour predicates; k > 0 = colour set k-1 of the table given to halo_set_color */
} HaloEntry;

/* Emit-gate filters — FilterConfig (src/config/filter_config.hpp:20-86) as the device matcher consumes it
 * (DeviceFilterDesc, src/core/device_filter_desc.hpp:60-100).  Face ids are crystal face NUMBERS. */
/* The reference's physical complex filters keep their AND-term counts in a flat host-built buffer (any number of OR-clauses up
 * to a 4096 sanity cap, device_filter_desc.hpp:56-79; its fixed 8 is the colour path's).  Here a complex filter is one
 * fixed-size record staged in LDS: up to 64 OR-clauses and 64 simple terms in all — the reference's test configs use at
 * most 12 (test/e2e/configs/parity_big_or_with_color.json). */
#define HALO_FILTER_MAX_OR 64
#define HALO_FILTER_MAX_TERMS 64  /* total simple terms over all AND-clauses of one complex filter */
enum { HALO_FILTER_NONE = 0, HALO_FILTER_RAYPATH = 1, HALO_FILTER_ENTRY_EXIT = 2, HALO_FILTER_DIRECTION = 3,
       HALO_FILTER_CRYSTAL = 4 };
enum { HALO_SYM_P = 1, HALO_SYM_B = 2, HALO_SYM_D = 4 }; /* FilterConfig::kSymP/B/D */
typedef struct HaloFilterTerm { /* SimpleFilterParam */
  int32_t type;
  int32_t raypath_len;
  uint8_t raypath[HALO_MAX_HITS]; /* raypath: face-number sequence */
  int32_t has_entry, entry, has_exit, exit_face; /* entry_exit: wildcards when has_* == 0 */
  uint32_t min_len, max_len;      /* entry_exit: path length bounds; max_len 0 = unbounded */
  float az, el, radii;            /* direction: degrees */
  int32_t crystal_id;             /* crystal: CrystalConfig::id_ */
} HaloFilterTerm;
typedef struct HaloFilter {
  int32_t action;    /* 0 = filter_in (match passes), 1 = filter_out */
  int32_t symmetry;  /* HALO_SYM_* bitmask */
  int32_t is_complex;
  int32_t or_count;  /* complex: number of OR-clauses; terms[] holds their AND-terms back to back */
  int32_t and_counts[HALO_FILTER_MAX_OR];
  HaloFilterTerm terms[HALO_FILTER_MAX_TERMS]; /* simple filter: terms[0] */
} HaloFilter;

/* Raypath colour (reference "Design 2", src/config/color_gate_table.hpp:25-81, cuda_trace_backend.cu:498-556): every
 * emitted exit carries a 64-bit component mask; a crystal entry's colour set lists predicates (SimpleFilterParam + P/B/D
 * symmetry) and the mask bit each sets when it matches the exit's raypath / direction / crystal; masks accumulate across
 * scattering layers (they ride with the continuation).  A colour CLASS is a bit set with an any/all rule; each in-frame
 * pixel hit of an exit whose mask satisfies class c adds cmf_y*w to lane c (class_count x W x H floats). */
#define HALO_COLOR_MAX_TERMS 16    /* predicates per crystal entry (ColorGatePlacement) */
#define HALO_COLOR_MAX_CLASSES 16
typedef struct HaloColorTerm { /* ColorGateEntry: predicate_, symmetry_, bit_ */
  HaloFilterTerm predicate;
  int32_t symmetry; /* HALO_SYM_* bitmask */
  int32_t bit;      /* 0..63 */
} HaloColorTerm;
typedef struct HaloColorSet {
  int32_t term_count;
  int32_t reserved;
  HaloColorTerm terms[HALO_COLOR_MAX_TERMS];
} HaloColorSet;
typedef struct HaloColorClass { /* ColorGateParams::color_class_bits / color_class_combine */
  uint64_t bits;
  int32_t combine_all; /* 0 = any bit of `bits` set, 1 = all of them */
  int32_t reserved;
} HaloColorClass;

typedef struct HaloLayer {
  float prob; /* MsInfo::prob_ — continuation probability to the next layer */
  int32_t entry_count;
  HaloEntry entries[HALO_MAX_ENTRIES];
} HaloLayer;

/* SceneConfig fields read on the path (sun: src/config/light_config.hpp SunParam). Degrees. */
typedef struct HaloScene {
  float sun_altitude;
  float sun_azimuth;
  float sun_diameter;
  int32_t max_hits; /* counts surface interactions INCLUDING the entry face (simulator.cpp:1308) */
  int32_t layer_count;
  HaloLayer layers[HALO_MAX_LAYERS];
} HaloScene;

/* LensParam::LensType integer values — src/core/shared/projection_shared.h:136-146. */
enum {
  HALO_LENS_LINEAR = 0,
  HALO_LENS_FISHEYE_EQUAL_AREA = 1,
  HALO_LENS_FISHEYE_EQUIDISTANT = 2,
  HALO_LENS_FISHEYE_STEREOGRAPHIC = 3,
  HALO_LENS_DUAL_FISHEYE_EQUAL_AREA = 4,
  HALO_LENS_DUAL_FISHEYE_EQUIDISTANT = 5,
  HALO_LENS_DUAL_FISHEYE_STEREOGRAPHIC = 6,
  HALO_LENS_RECTANGULAR = 7,
  HALO_LENS_FISHEYE_ORTHOGRAPHIC = 8,
  HALO_LENS_DUAL_FISHEYE_ORTHOGRAPHIC = 9,
  HALO_LENS_GLOBE = 10
};
enum { HALO_VISIBLE_UPPER = 0, HALO_VISIBLE_LOWER = 1, HALO_VISIBLE_FULL = 2 };

/* RenderConfig fields read by BuildProjParams — src/core/lens_proj_build.hpp:79-137,
 * src/config/render_config.hpp:71-103.  Degrees. */
typedef struct HaloRender {
  int32_t lens_type;
  float fov;
  int32_t width;
  int32_t height;
  int32_t lens_shift[2];
  float view_az;
  float view_el;
  float view_ro;
  int32_t visible;
  float overlap; /* dual-fisheye overlap band (max |dz|), 0 = off */
} HaloRender;

/* Illuminants — src/util/illuminant_data.hpp:12-19. */
enum { HALO_ILLUM_D50 = 0, HALO_ILLUM_D55 = 1, HALO_ILLUM_D65 = 2, HALO_ILLUM_D75 = 3, HALO_ILLUM_A = 4, HALO_ILLUM_E = 5 };

/* WlParam (discrete: one session per wavelength) or illuminant pool
 * (src/core/backend/wl_pool.hpp:67-91: M mid-point wavelengths on [380,780], per-ray pick). */
typedef struct HaloWl {
  float wavelength; /* nm; used when illuminant < 0 */
  float weight;     /* spectral weight; used when illuminant < 0 */
  int32_t illuminant; /* -1 = discrete wavelength, else HALO_ILLUM_* */
  int32_t pool_size;  /* illuminant pool entries M (0 → 64 default, cap 255) */
} HaloWl;

/* HostRayBatch — src/core/backend/trace_backend.hpp:230-239.  Crystal-local golden-ray ingest. */
typedef struct HaloHostRays {
  const float* d;      /* 3*count */
  const float* p;      /* 3*count */
  const float* w;      /* count */
  const uint32_t* tf;  /* count — compact polygon-face id of the entry face */
  const struct HaloGeomTables* crystal;   /* HostRayBatch::crystal (trace_backend.hpp:230-239): the crystal the rays were sampled on, or NULL =
                                             the entry's own.  A supplied crystal is traced as it is — it consumes no MakeCrystal draw, so it is
                                             not a stochastic sample (test_cpu_trace_backend.cpp:737-777) — whatever the entry's shape distributions */
} HaloHostRays;

/* LayerStats — trace_backend.hpp:296-299 (+ continuation count from LayerHandle). */
typedef struct HaloLayerStats {
  uint64_t root_count;
  uint64_t exit_count;       /* exits emitted to the image (or captured) in this layer */
  uint64_t continuation_count;
  double exit_w_sum;
  double kernel_ms;          /* HIP-event time of this layer's kernels on the backend stream */
  uint64_t pixel_hits;       /* in-frame pixel writes (primary + overlap): 3 fp32 accumulator RMWs each */
  uint64_t launches;         /* kernel launches this layer took */
} HaloLayerStats;

/* ExitRayRecord — src/core/exit_seam.hpp:40-53, trimmed to what the parity tests read.
 * Only produced when option "capture_exits" is on (test path; production accumulates on device). */
typedef struct HaloExitRecord {
  float dir[3];     /* world-space exit direction */
  float weight;
  uint32_t root;    /* root-ray index within the layer (entries are laid out back to back) */
  uint16_t seq;     /* 2*interaction_index + child (0 = reflected, 1 = refracted); 0 = entry-face external reflection */
  uint8_t layer;
  uint8_t path_len;
  uint8_t path[HALO_PATH_CAP]; /* crystal face NUMBERS (1..8 prism; pyramid 1,2,3-8,13-18,23-28) */
  int32_t pixel;    /* flat pixel index of the primary hit, -1 = culled / out of frame */
  uint16_t crystal_id;
  uint16_t wl_idx;
  uint64_t color_mask; /* raypath-colour component mask of the exit (0 without halo_set_color) */
} HaloExitRecord;

typedef struct HaloBackend* halo_handle_t;

/* --- lifecycle ---------------------------------------------------------------------------- */
/* Number of visible HIP devices (CudaDeviceAvailable analogue, cuda_trace_backend.cu:147-193). */
int halo_device_count(void);
/* CreateBackend(BackendKind) — simulator.cpp:854-919.  seed must be non-zero (effective_seed_,
 * simulator.cpp:782-798); the backend seeds ONCE and keeps monotone ray counters across sessions
 * (cpu_trace_backend.hpp:131-140, cuda_trace_backend.cu:3724-3741). */
int halo_create(int device_ordinal, uint32_t seed, halo_handle_t* out);
int halo_destroy(halo_handle_t h);
const char* halo_last_error(halo_handle_t h);
/* Options: "capture_exits" (0/1), "geom_clock" (rays per sampled shape, default 32 — simulator.hpp:144),
 * "seed" (re-seed the engine between sessions: the reference's backends are seeded by the first non-zero SessionSpec::seed,
 * cpu_trace_backend.cpp:248-257), "rank"/"world" (shard id mixed into the ray counters so ranks draw disjoint streams), "ray_base" (the 64-bit index of the next
 * root ray — SplitPcgRayBase, trace_backend.hpp:184; counters run on from it, across 2^32 with the carry of pcg_advance_hi),
 * "chunk" (max rays per kernel launch, default 256 Mi), "aggregate" (0 plain atomics, 1 LDS pixel cache [default], 2 diagnostic no-accumulate),
 * "mono" (one-channel accumulation for discrete-wavelength sessions, default 1), "mono_copies" (privatised copies of that
 * plane, power of two, default 8), "async" (queue final-layer dispatches without a host sync, see halo_collect_stats),
 * "bin" (binned accumulation: hits staged in LDS and flushed to per-tile hit lists by the trace kernel — one level of lists up to
 * 512 tiles of 16 Ki accumulator slots, two levels (coarse lists, split pass) beyond; -1 [default] = only for the full-sky
 * launches >= 2 Mi rays of sessions with one plane per pool entry ("lambda_planes" = 1), where the hit log cannot go; 0 = never;
 * 1 = always when applicable), "bin_l1" (coarse lists of the two-level route, default 128), "stoch_chunk" (rays per launch with
 * device-generated crystal pools; default 64 Mi),
 * "lambda_planes" (illuminant sessions: -1 [default] = batches >= 2 Mi rays on images above 512 Ki pixels keep X/Y/Z planes and
 * their launches log {slot, pool entry, weight}, the log's per-tile pass makes X, Y, Z; other batches >= 8 Mi rays keep one
 * scalar plane per wavelength-pool entry, CMF applied by the closing fold; the rest X/Y/Z planes by direct atomics; 0 = never
 * per-entry planes nor the X/Y/Z log; 1 = always per-entry planes),
 * "host_shapes" (1 = build stochastic shape pools on the host and upload them; default 0 = device generator),
 * "blocks_per_cu" (cap on workgroups per CU of one launch, default 24; launches are sized for >= 32 ray-loop passes per
 * workgroup below that cap; small launches take one workgroup per CU up to 3 x 2^17 rays and two up to 2^22),
 * "lazy_fold" (1 [default]: halo_end of a session on the backend's OWN accumulator leaves its planes unfolded; every reader of the
 * image — readback, consumer fold, reduce, take_landed, bind — folds first, and so does halo_begin of a session with other planes
 * (wavelength, image size, plane or copy count): many small equal sessions cost one fold per readback; an accumulator bound with
 * halo_bind_accumulator is always folded at halo_end, its owner reads it without asking; 0: fold at every halo_end),
 * "gen_serial" (0 [default]: stochastic pyramids are built by teams of 32 lanes per crystal; 1: one thread per crystal — the
 * same builder the host runs; records are bit-identical either way, A/B knob),
 * "hit_log" (-1 [default]: production-mode launches >= 2 Mi rays on one scalar plane (>= 512 Ki rays when the render's visible range is
 * FULL: every exit of a full-sky render lands, and that many direct atomics cost more than the log's passes), or on the X/Y/Z planes of an illuminant
 * session (>= 2 Mi rays, image above 512 Ki pixels), append the hits that miss the pixel cache to a log region per workgroup — plain stores instead of
 * memory-side fp32 atomics — which a split pass and a per-tile LDS pass then add to the plane(s); 0 = never (direct atomics),
 * 1 = whenever applicable), "hit_log_cap" (test knob: records per log region, 0 [default] = sized from the launch; what runs
 * over a region or a tile list is added directly),
 * "hex_fast" (1 [default]: one-shape dispatches of a REGULAR hexagonal prism run the instantiation whose next-face search has the
 * normals as literals; 0: the table-driven search — same candidates, order and comparisons, A/B knob),
 * "entry_fast" (1 [default]: one-shape dispatches of a full 8-face prism pick the entry face slab by slab, in registers;
 * 0: the generic walk over faces — same uniform, same cumulative order, A/B knob),
 * "pool_entry_fast" (1 [default]: logged launches over sampled PRISMS pick the entry face of every full eight-face prism slab by slab, from tables
 * its half-wave rebuilds each pass; 0: the walk over the fan triangles — same uniform, same cumulative order; the same prisms, when their slab normals are the regular prism's, search their next face with literal normals — same candidates and comparisons as the table-driven search; A/B knob),
 * "rehit_strategy" (which of the reference's two next-face strategies the trace follows where they part, src/core/shared/traversal_shared.h:23-29:
 * 1 [default] = its CUDA backend's — the child that leaves through the face the ray stands on is an outgoing candidate where it is; 0 = its
 * legacy CPU path's, PropagateSlab optics.cpp:64-158 — that child is propagated over all faces, the source face included, with the relaxed
 * accept threshold, and one that "re-hits" goes on as a segment.  The two agree on a convex crystal whose entry points lie on their faces'
 * planes; they part on a fan the vertex merge moved off its plane.  0 runs every launch on the generic kernels (no exit queue, no hit log:
 * several times slower) — for a caller that needs SURVEY's stated ground truth ray for ray; not inside a session),
 * "shuffle_chunk" (Recombine's shuffle permutes chunks of this many consecutive continuation-pool entries; power of two in
 * [1, 64], default 32 = one 128-byte line per plane read; 1 = the reference's per-ray permutation, cu:1633-1657),
 * scheduling (ABI 6; none of them changes a result): "overlap" (1 [default]: launches of <= 2 Mi rays alternate between two
 * trace streams, closing folds run on an auxiliary stream; 0 = everything on the one stream), "table_cache" (1 [default]: equal
 * sessions reuse their device tables), "small_blocks_per_cu" (default 5: workgroups per CU a small launch spreads over before a
 * workgroup takes a second pass), "defer_fold" (see halo_flush), "gen_ahead" (experiment knob, 0 [default]; 1: the crystal generator of a
 * launch that fills the chip is queued beside the previous launch's trace kernel — measured, gains nothing: both are VALU-bound). */
int halo_set_option(halo_handle_t h, const char* key, int64_t value);
/* Use an external HIP stream (e.g. torch's current stream) for all launches. NULL = own stream. */
int halo_set_stream(halo_handle_t h, void* hip_stream);
/* Bind an externally-owned device accumulator of width*height*3+4 floats (e.g. a torch tensor, so
 * torch.distributed can reduce it in place).  NULL = backend-owned.  Layout: the xyz image, then 4 reserved floats
 * (kept zero; the landed-weight tally is a separate fp64 scalar read with halo_take_landed / halo_readback_xyz64). */
int halo_bind_accumulator(halo_handle_t h, void* device_ptr, uint64_t n_floats);
/* The filter table HaloEntry::filter_id indexes (ConfigManager::filters_, config_manager.cpp:184-215). Copied; stays in
 * force until replaced.  A filter-failing exit is dropped: neither emitted nor continued (CollectData simulator.cpp:725). */
int halo_set_filters(halo_handle_t h, const HaloFilter* filters, int32_t count);

/* --- session ------------------------------------------------------------------------------ */
/* TraceBackend::BeginSession(SessionSpec) — trace_backend.hpp:374-378. scene/render are COPIED.  ray_num (SessionSpec::ray_num)
 * is the number of roots the session is going to trace; it only selects the accumulation layout of illuminant sessions
 * (see "lambda_planes"), results do not depend on it. */
int halo_begin(halo_handle_t h, const HaloScene* scene, const HaloRender* render, const HaloWl* wl, uint64_t ray_num);
/* TraceBackend::TraceLayer(RootRaySource) — trace_backend.hpp:380-389.  First call of a session:
 * host mode, `count` roots self-generated on device (rays == NULL) or injected (rays != NULL).
 * Later calls: device mode, consumes the continuation produced by halo_recombine (count ignored). */
int halo_trace_layer(halo_handle_t h, uint64_t count, const HaloHostRays* rays, HaloLayerStats* stats);
/* TraceBackend::Recombine(handle, RecombineSpec{shuffle}) — trace_backend.hpp:391-395.  No data moves: the pools swap
 * roles and the next layer reads its roots through a Feistel bijection of the pool positions.  The bijection permutes
 * chunks of 32 consecutive pool entries (32 different parent rays), not single entries like the reference's CUDA
 * shuffle_cont_kernel (cu:1633-1657): same decorrelation of position ranges, coalesced reads. */
int halo_recombine(halo_handle_t h, int shuffle, uint64_t* continuation_count);
/* TraceBackend::DrainExits — trace_backend.hpp:430-448 (only with "capture_exits").  Copies at most `cap` pending records
 * (oldest first) into `out`, *count = records copied; the rest stays pending, so a caller may drain in pieces.
 * out == NULL: *count = number of pending records, nothing is consumed. */
int halo_drain_exits(halo_handle_t h, HaloExitRecord* out, uint64_t cap, uint64_t* count);
/* TraceBackend::EndSession. The accumulator persists (SupportsThirdClockDrain, cu:4801-4808). */
int halo_end(halo_handle_t h);
/* TraceBackend::ReadbackXyzAccum — trace_backend.hpp:461-469, cuda_trace_backend.cu:4800-4846:
 * sync, ADD landed weight into *landed_weight, copy W*H*3 floats, zero the device accumulator. */
int halo_readback_xyz(halo_handle_t h, float* xyz, int width, int height, float* landed_weight);
/* Same, but returns landed weight in double and does not add. */
int halo_readback_xyz64(halo_handle_t h, float* xyz, int width, int height, double* landed_weight);
/* Raypath colour tables: `sets` is referenced by HaloEntry.color_id (1-based), `classes` defines the Y lanes.
 * n_sets = n_classes = 0 switches colour off (the default; the production kernels then carry no mask at all). */
int halo_set_color(halo_handle_t h, const HaloColorSet* sets, int n_sets, const HaloColorClass* classes, int n_classes);
/* TraceBackend::ReadbackClassLanes — trace_backend.hpp:471-493: copies class_count*W*H floats (lane c at
 * lanes[c*W*H + py*W + px]) and zeroes the device lanes.  Call after halo_readback_xyz's sync point or halo_sync. */
int halo_readback_class_lanes(halo_handle_t h, float* lanes, int width, int height, int class_count);
/* TraceBackend::GetLastBatchStochasticCrystalSampleCount / ...OrientationSampleCount (trace_backend.hpp:587,625), for the
 * session traced last (counters reset at halo_begin): crystal instances actually sampled (one per geom_clock rays of every
 * crystal entry that is not IsDeterministic; 0 for fixed shapes) and rays whose orientation was drawn (every ray of an entry
 * whose axis has a non-fixed distribution).  Real counts of what the kernels did, not estimates. */
int halo_last_sample_counts(halo_handle_t h, uint64_t* crystal_samples, uint64_t* orientation_samples);
/* Which kernels served the session traced last (masks reset at halo_begin).  Diagnostics for the parity tests and the bench:
 * a test that means to check the production binned shape-pool kernel asserts that it really ran. */
typedef struct HaloRouteInfo {
  uint32_t launches;     /* trace-kernel launches since halo_begin */
  uint32_t mode_mask;    /* bit m: a launch ran the MODE m instantiation: 0 production; 1 production + emit-gate filter, 4 production + filter and
                            raypath colour (both for max_hits <= 16: path in a register, predicates as host-built member tables); 2 + exit capture
                            (tests); 3 the generic filter / colour kernels (paths to 64 faces, or tables that do not fit the fast form) */
  uint32_t geom_mask;    /* bit g: GEOM g (0 one shape per dispatch, 1 pool of 4.1 KB records, 2 pool of prism records, 3 one shape = regular hexagonal prism) */
  uint32_t accum_mask;   /* bit 0 direct X/Y/Z planes, 1 direct scalar plane(s), 2 binned one level, 3 binned two levels, 4 hit log, 5 hit log of an illuminant session (X, Y, Z made in the per-tile pass), 6 none (a layer whose every exit continues) */
  uint32_t source_mask;  /* bit 0 generated roots, 1 continuation pool (layer >= 1), 2 host-injected rays */
  uint32_t plane_cnt;    /* accumulation planes of the session (1 discrete, 3 X/Y/Z, M per-entry) */
  uint32_t plane_copies; /* privatised copies of each plane */
  uint32_t shuffle_chunk;/* pool entries that move together through Recombine's shuffle */
  uint32_t spec_mask;    /* specialised instantiations that ran — bit 0: last-layer kernel (no continuation-append code), 1: lens as a compile-time
                            constant, 2: visible range as a constant, 3: closed gate (prob <= 0) as a constant */
  uint32_t generic_launches; /* launches that ran an instantiation with none of those specialisations */
} HaloRouteInfo;
int halo_last_route(halo_handle_t h, HaloRouteInfo* out);
/* Bring the accumulator up to date WITHOUT a host wait: the closing folds of the ended sessions are queued and the backend's stream is made
 * to wait for them, so that work queued on that stream afterwards (a collective on a bound accumulator, a copy) sees the finished image.
 * Needed with option "defer_fold" = 1 (halo_end then leaves the fold of a caller-bound accumulator pending so that the next session's trace
 * kernels run under it); harmless otherwise.  Not inside a session. */
int halo_flush(halo_handle_t h);
/* Kernel time of the sessions' LAST layers since the previous call (HIP events on the launch's stream, read after a host wait): trace_ms = the
 * trace kernels' own spans, post_ms = the spans of their accumulation passes (split + per-tile sums), launches = dispatches timed.  For
 * bench.py's roofline object, which prices the last layer's trace kernel; HaloLayerStats::kernel_ms is trace + passes of every layer. */
int halo_collect_timing(halo_handle_t h, double* trace_ms, double* post_ms, uint64_t* launches);
int halo_sync(halo_handle_t h);
/* Tallies of every layer traced since the previous call (summed), after waiting for the stream. With option
 * "async" = 1 a final-layer halo_trace_layer only queues its dispatches (its `stats` carry root_count alone) and the
 * exit / pixel-hit / kernel-time tallies are delivered here — the reference's LayerStats are diagnostics
 * (trace_backend.hpp:296-299), nothing on the path waits for them. */
int halo_collect_stats(halo_handle_t h, HaloLayerStats* out);
/* Read AND zero the device landed-weight tally (fp64) without touching the image: used when the image lives in a
 * bound external accumulator that is reduced across ranks in place. */
int halo_take_landed(halo_handle_t h, double* landed_weight);

/* Multi-GPU drain for a C/C++ host (one process — or one thread — per GPU, one backend per GPU): sum-reduce this rank's XYZ
 * accumulator (width*height*3+4 floats, owned or bound) onto rank `root` with ONE ncclReduce over RCCL/xGMI, queued on the
 * backend's stream behind its trace kernels; every other rank's accumulator is then zeroed (drained) in stream order.
 * `nccl_comm` is the caller's ncclComm_t (rccl.h); `root` its root rank.  The landed-weight scalars stay on their devices: read
 * them with halo_take_landed and add on the host.  RCCL is loaded lazily (dlopen "librccl.so"): the library has no link-time
 * dependency on it, and a host that never calls this never loads it.  HALO_UNAVAILABLE when RCCL cannot be loaded.
 * Reference: the drain point is Simulator::DrainDeviceXyz (simulator.cpp:1409-1477); Lumice itself has no multi-GPU code. */
int halo_reduce_accumulator(halo_handle_t h, void* nccl_comm, int root, int this_rank);

/* --- consumer on device (RenderConsumer::ConsumeDeviceFused / PrepareSnapshot / PostSnapshot, server/render.cpp) -- */
/* HaloDisplay: the RenderConfig fields PostSnapshot reads (render_config.hpp:84-92). ray_color[0] < 0 = real colour. */
typedef struct HaloDisplay {
  float intensity_factor;
  float ray_color[3];
  float background[3];
} HaloDisplay;
/* ConsumeDeviceFused (render.cpp:138-201): fold the device accumulator into the running image with Neumaier-compensated
 * adds (accum_shared.h:70-74), add its landed weight to total_intensity, zero the accumulator.  All on device. */
int halo_consumer_fold(halo_handle_t h);
/* The same fold for a drained image the CALLER holds — RenderConsumer::ConsumeDeviceFused(const SimData&), render.cpp:138-201, as it is
 * written: xyz = SimData::xyz_pixel_data_ (W*H*3 floats, Neumaier-folded into the running image), landed = xyz_landed_weight_ (added to
 * total_intensity), lanes = lane_pixel_data_ (class_count x W x H floats added to the class lanes; NULL / 0 = none).  What another rank,
 * another backend or the legacy path drained becomes part of this consumer; the first call sizes the consumer to width x height. */
int halo_consumer_consume(halo_handle_t h, const float* xyz, int width, int height, float landed, const float* lanes, int class_count);
/* PrepareSnapshot + PostSnapshot (render.cpp:465-578): snapshot = sum + compensation; scale = intensity_factor * 0.08 *
 * N_pix / total_intensity (ExposureScale :96-102); XYZ → gamut clip → linear RGB → sRGB u8 (util/color_space.cpp).
 * rgb_out: W*H*3 bytes (nullable); xyz_out: W*H*3 floats raw snapshot (nullable); total_intensity out (nullable). */
int halo_consumer_snapshot(halo_handle_t h, const HaloDisplay* display, uint8_t* rgb_out, float* xyz_out, double* total_intensity);
int halo_consumer_reset(halo_handle_t h);

/* --- display-side composite of the raypath-colour class lanes (server/component_compositor.cpp) ------------- */
/* CompositeMode (component_compositor.hpp:21): how the per-class Y lanes become one colour per pixel. */
enum { HALO_COMPOSITE_DOMINANT = 0, HALO_COMPOSITE_ADDITIVE = 1, HALO_COMPOSITE_PAINTER = 2 };
/* The display half of a ColorClass (config/color_class_table.hpp:21-35): colour, visibility, solo, draw order.  Entry c belongs
 * to lane c of halo_set_color's `classes`; z_order re-orders the DRAW, never the lane binding (component_compositor.cpp:34-37). */
typedef struct HaloCompositeClass {
  float color[3];
  int32_t z_order;
  int32_t visible; /* ColorClass::visible_ */
  int32_t solo;    /* ColorClass::solo_: if any class is solo, only solo classes take part */
} HaloCompositeClass;
typedef struct HaloComposite {
  int32_t mode;                 /* HALO_COMPOSITE_* */
  float display_exposure_scale; /* GUI EV multiplier; 1 = none */
  float intensity_factor;       /* RenderConfig::intensity_factor_ (ParticipatingExposureScale, render.cpp:120-135) */
  int32_t class_count;          /* must equal the class count of halo_set_color */
  HaloCompositeClass classes[HALO_COLOR_MAX_CLASSES];
} HaloComposite;
/* CompositeColorClassesLinear + LinearRgbToSrgbU8 (component_compositor.cpp:180-303) on the device lanes, which stay as they are
 * (the consumer's lanes: they accumulate over sessions until halo_readback_class_lanes or halo_consumer_reset).  The participating
 * P99 is the exact order statistic the reference takes with nth_element (index = float(count) * 0.99f of the positive lane values
 * of the participating classes), found by a radix select over the float bit patterns; A = intensity_factor * target_linear / P99.
 * *produced = 0 where the reference returns false: no class bit referenced (outputs untouched), or total intensity / P99 zero (P99
 * written, the linear image zeroed as the reference's assign does, :189).  linear_rgb_out: W*H*3 floats, srgb_out: W*H*3 bytes (either
 * may be NULL).  total intensity = what halo_consumer_fold has summed (or halo_consumer_load_lanes set). */
int halo_consumer_composite(halo_handle_t h, const HaloComposite* spec, float* linear_rgb_out, uint8_t* srgb_out,
                            float* participating_p99_y, int32_t* produced);
/* Replace the device lanes with host values (class_count x W x H floats, lane c at lanes[c*W*H + py*W + px]): lanes summed over
 * ranks on the host, or a saved consumer.  total_intensity >= 0 also replaces the consumer's total intensity.  Needs halo_set_color
 * (the lane count is its class count); allocates the lanes when no session has yet. */
int halo_consumer_load_lanes(halo_handle_t h, const float* lanes, int width, int height, int class_count, double total_intensity);
/* ParseCompositeMode (component_compositor.cpp:118-134): "dominant" / "additive" / "painter"; anything else is painter. */
int halo_host_parse_composite_mode(const char* mode);

/* --- host-side pieces of the path, exported for parity tests (no GPU needed) ---------------- */
/* Geometry tables the kernels consume (reference: Crystal::PopulateFromCfGeom crystal.cpp:304-347,
 * detail::BuildEntrySubTris simulator.cpp:90-129). */
typedef struct HaloGeomTables {
  int32_t face_cnt;                 /* compact present faces */
  float face_n[HALO_MAX_FACES * 3];
  float face_d[HALO_MAX_FACES];
  int32_t face_number[HALO_MAX_FACES];
  int32_t tri_cnt;
  float tri_v[HALO_MAX_TRIS * 9];
  float tri_n[HALO_MAX_TRIS * 3];
  float tri_area[HALO_MAX_TRIS];
  int32_t tri_face[HALO_MAX_TRIS];
} HaloGeomTables;
/* Crystal::CreatePrism(h, dist) — crystal.cpp:349-377. Returns HALO_OK; face_cnt==0 = empty crystal. */
int halo_host_prism_geometry(float h, const float dist[6], HaloGeomTables* out);
/* Crystal::CreatePyramid(wedge_u, wedge_l, h1, h2, h3, dist) — crystal.cpp:379-426. */
int halo_host_pyramid_geometry(float wedge_upper_deg, float wedge_lower_deg, float h1, float h2, float h3,
                               const float dist[6], HaloGeomTables* out);
/* The nine shape scalars [h0, h1, h2, d0..d5] of crystal instance `shape_index` (SyncGroupSampler, simulator.cpp:361-393, over the PCG
 * shape stream): via_plan = 0 walks the draw sequence as the host builder does, 1 draws each scalar from the per-dispatch draw plan the
 * device generators use (one lane per scalar).  The two must agree bit for bit.  Parity-test hook, no device needed. */
int halo_host_shape_scalars(const HaloCrystal* crystal, uint32_t seed, uint64_t shape_index, int via_plan, float out9[9]);
/* Sample crystal instances [first_index, first_index + n) of `crystal` from the backend's shape-scalar stream
 * (MakeCrystal simulator.cpp:448 + SyncGroupSampler :361-393 + closed-form geometry) into `out[n]`:
 * on_device = 1 runs the device generator of the general (4.1 KB) records, 2 — prisms only — the generator of the prism pools' 1360-byte
 * records (one team of 8 lanes per crystal: what a stochastic-prism trace really runs), 0 the host builder.
 * All of them are the same geometry (csrc/halo_geom.h); the tables are bit-equal whenever the draws involve no libm call
 * (fixed / uniform distributions) and equal to float rounding otherwise.  Parity-test hook. */
int halo_generate_shapes(halo_handle_t h, const HaloCrystal* crystal, uint64_t first_index, uint32_t n, int on_device,
                         HaloGeomTables* out);
/* BuildLatLut — lat_lut.cpp:74-204. theta/cdf/flip are HALO_LUT_NODES floats each. */
int halo_host_build_lat_lut(const HaloDist* latitude, float* theta, float* cdf, float* flip);
/* BuildProjParams — lens_proj_build.hpp:79-137 → 19 x 4 bytes laid out as lm_proj::ProjParams. */
int halo_host_build_proj_params(const HaloRender* render, void* proj_params_76_bytes);
/* PartitionCrystalRayNum — simulator.cpp:519-582. carry has n entries and persists across calls. */
int halo_host_partition(const float* proportions, int n, uint64_t ray_num, double* carry, uint64_t* out_counts);
/* Crystal::ReduceRaypath(rp, symmetry, sigma_a, d_applicable) — crystal.cpp:536-600 (hexagonal families, fn_period 6). */
int halo_host_reduce_raypath(const uint8_t* rp, int32_t n, int32_t symmetry, int32_t sigma_a, int32_t d_applicable, uint8_t* out);
/* The emit-gate filter predicate as the production filter kernels evaluate it — DeviceFilterCheck, shared/filter_shared.h:308-315, on the
 * host-built member tables (csrc/halo_device.h FastTables; max_hits <= 16): filter `f` for a crystal with orientation `axis`, asked about an
 * exit with face-number path `path[0..n)` (n <= 16), world direction `dir` and crystal id `crystal_id`.  Writes 1 / 0 to *pass.  HALO_FATAL
 * when the filter does not fit the fast form (the backend then runs the generic kernels).  A host-side test hook: the tests compare it with
 * the reduction-based predicate (halo_host_reduce_raypath) over every short face sequence. */
int halo_host_filter_fast_check(const HaloFilter* f, const HaloAxis* axis, const uint8_t* path, int32_t n, const float dir[3], int32_t crystal_id, int32_t* pass);
/* The raypath-colour pass of the production colour kernels on the host — ApplyLayerColorBits, cuda_trace_backend.cu:498-527; CPU:
 * the colour loop of CollectData, simulator.cpp:689-716 (FilterSpec::CheckSummandMask per predicate): every predicate of `colors`
 * (with its OWN symmetry, evaluated for a crystal with orientation `axis`) that matches the exit ORs its bit into `carried`.
 * Same member tables as halo_host_filter_fast_check (n <= 16).  A host-side test hook for the reference's component-gate vectors. */
int halo_host_color_fast_mask(const HaloColorSet* colors, const HaloAxis* axis, const uint8_t* path, int32_t n, const float dir[3], int32_t crystal_id,
                              uint64_t carried, uint64_t* mask);
/* IceRefractiveIndex::Get — optics.cpp:180-197. */
double halo_host_refractive_index(double wavelength_nm);
/* GetIlluminantSpd(type, wavelength) — util/illuminant.cpp:113-134 (HALO_ILLUM_*; 0 outside the tabulated range). */
float halo_host_illuminant_spd(int illuminant, float wavelength_nm);
/* ComputeWlPool — core/backend/wl_pool.hpp:67-91: the session's wavelength entries as the kernels read them, 5 floats each
 * {refractive index, SPD weight, cmf_x, cmf_y, cmf_z}; a discrete HaloWl gives one entry.  Returns the entry count (<= cap
 * entries are written), 0 on error. */
int halo_host_wl_pool(const HaloWl* wl, float* entries5, int cap);
int halo_abi_version(void);
/* sizeof() of boundary structs as compiled (0 scene, 1 render, 2 wl, 3 exit record, 4 geom tables, 5 layer stats, 6 entry, 7 colour set, 8 colour class, 9 filter, 10 route info, 11 composite). */
uint64_t halo_abi_sizeof(int which);

#ifdef __cplusplus
}
#endif

/* Layout pins of every struct that crosses the boundary (LP64, natural alignment).  A field added, removed or re-typed on
 * either side of the ABI — here, in a binding (ice_halo_sim_amd/abi.py checks the same numbers through halo_abi_sizeof), or
 * in the reference-side glue (integration/hip_backend_glue.hpp) — fails to compile instead of shifting what the kernels read. */
#if defined(__cplusplus)
#define HALO_STATIC_ASSERT(c, m) static_assert(c, m)
#elif defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L
#define HALO_STATIC_ASSERT(c, m) _Static_assert(c, m)
#else
#define HALO_STATIC_ASSERT(c, m)
#endif
HALO_STATIC_ASSERT(sizeof(HaloDist) == 12, "HaloDist");
HALO_STATIC_ASSERT(sizeof(HaloAxis) == 36, "HaloAxis");
HALO_STATIC_ASSERT(sizeof(HaloCrystal) == 4 + 9 * 12 + 9 * 4 + 8, "HaloCrystal");
HALO_STATIC_ASSERT(sizeof(HaloEntry) == sizeof(HaloCrystal) + sizeof(HaloAxis) + 16, "HaloEntry");
HALO_STATIC_ASSERT(sizeof(HaloFilterTerm) == 8 + HALO_MAX_HITS + 16 + 8 + 12 + 4, "HaloFilterTerm");
HALO_STATIC_ASSERT(sizeof(HaloFilter) == 16 + 4 * HALO_FILTER_MAX_OR + HALO_FILTER_MAX_TERMS * sizeof(HaloFilterTerm), "HaloFilter");
HALO_STATIC_ASSERT(sizeof(HaloColorTerm) == sizeof(HaloFilterTerm) + 8, "HaloColorTerm");
HALO_STATIC_ASSERT(sizeof(HaloColorSet) == 8 + HALO_COLOR_MAX_TERMS * sizeof(HaloColorTerm), "HaloColorSet");
HALO_STATIC_ASSERT(sizeof(HaloColorClass) == 16, "HaloColorClass");
HALO_STATIC_ASSERT(sizeof(HaloLayer) == 8 + HALO_MAX_ENTRIES * sizeof(HaloEntry), "HaloLayer");
HALO_STATIC_ASSERT(sizeof(HaloScene) == 20 + HALO_MAX_LAYERS * sizeof(HaloLayer), "HaloScene");
HALO_STATIC_ASSERT(sizeof(HaloRender) == 44, "HaloRender");
HALO_STATIC_ASSERT(sizeof(HaloWl) == 16, "HaloWl");
HALO_STATIC_ASSERT(sizeof(HaloHostRays) == 5 * sizeof(void*), "HaloHostRays");
HALO_STATIC_ASSERT(sizeof(HaloLayerStats) == 56, "HaloLayerStats");
HALO_STATIC_ASSERT(sizeof(HaloExitRecord) == 40 + HALO_PATH_CAP, "HaloExitRecord");
HALO_STATIC_ASSERT(sizeof(HaloRouteInfo) == 40, "HaloRouteInfo");
HALO_STATIC_ASSERT(sizeof(HaloDisplay) == 28, "HaloDisplay");
HALO_STATIC_ASSERT(sizeof(HaloCompositeClass) == 24, "HaloCompositeClass");
HALO_STATIC_ASSERT(sizeof(HaloComposite) == 16 + HALO_COLOR_MAX_CLASSES * 24, "HaloComposite");
HALO_STATIC_ASSERT(sizeof(HaloGeomTables) == 4 + HALO_MAX_FACES * 20 + 4 + HALO_MAX_TRIS * (36 + 12 + 4 + 4), "HaloGeomTables");
#endif /* HALO_TRACE_H_ */
