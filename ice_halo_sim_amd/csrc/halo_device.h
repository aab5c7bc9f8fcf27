// halo_device.h — POD parameter blocks shared by the host orchestration (halo_backend.cpp) and the
// gfx950 kernels (halo_trace.inl, halo_kernels.hip).  gfx950 only; no other backend is supported or dispatched.
#ifndef HALO_DEVICE_H_
#define HALO_DEVICE_H_

#include <stdint.h>

#include "../../include/halo_trace.h"

namespace halo {

constexpr int kBlock = 256;          // 4 wave64 per workgroup: one per SIMD of a CU
constexpr int kMaxFaces = HALO_MAX_FACES;
constexpr int kMaxTris = HALO_MAX_TRIS;
constexpr int kLutNodes = HALO_LUT_NODES;

// lm_pcg::kLatPath* wire values (reference src/core/shared/pcg_shared.h:56-59)
enum : uint32_t { kLatFullSphere = 0u, kLatNoRandom = 1u, kLatGaussLegacy = 3u, kLatLut = 6u };

// PCG stream-family nonces (reference cuda_trace_backend.cu:259-278, pcg_shared.h:119-120)
constexpr uint32_t kNonceTransit = 0xA5A5A5A5u;
constexpr uint32_t kNonceGate = 0x5A5A5A5Au;
constexpr uint32_t kNonceGen = 0x3C9A7F11u;
constexpr uint32_t kNonceShuffle = 0xB17CA3D9u;
constexpr uint32_t kNonceWl = 0x9E3779B9u;
constexpr uint32_t kNonceShapeHost = 0x6A09E667u;

// One crystal shape as the kernels read it.  Face rows are {nx, ny, nz, d}; triangle rows carry the fan
// triangle (v0, v1, v2), its raw winding normal, area and compact face id (reference
// Crystal::PopulateFromCfGeom crystal.cpp:304-347, detail::BuildEntrySubTris simulator.cpp:90-129).
//
// Slab table for the next-face search: a convex crystal's faces mostly come in opposite pairs (both basal faces, prism
// sides i / i+3) whose normals are exact negatives, so one pair of dot products (n.d, n.p) serves both faces and only the
// face on the side the ray travels towards can be ahead.  slab[k] = {nx, ny, nz, d_plus | d_minus, id_plus, id_minus, -}
// (ids as int bits); faces without an exact opposite are listed in `single`.
constexpr int kMaxSlabs = kMaxFaces / 2;
struct ShapeDev {
  int32_t face_cnt;
  int32_t tri_cnt;
  int32_t slab_cnt, single_cnt;
  float face[kMaxFaces][4];
  float slab[kMaxSlabs][8];
  float tri_v[kMaxTris][9];
  float tri_na[kMaxTris][4];          // nx, ny, nz, area
  uint8_t tri_face[kMaxTris];
  uint8_t face_number[kMaxFaces];
  uint8_t single[kMaxFaces];
  uint8_t pad2[8];
};
static_assert(sizeof(ShapeDev) % 16 == 0, "ShapeDev rows are read as float4");

// What the kernels of ONE REGULAR hexagonal prism read of the dispatch's shape: the header, the face rows (normal + plane constant of the
// Fresnel split) and the four slab rows (the entry pick's dot products).  Everything else they take from EntryFastDev or have as literals,
// so their workgroups stage this 464-byte prefix of ShapeDev instead of its 4.1 KB — with that (and a byte per queued exit's pool entry) the
// kernel's LDS is 31 KB and five workgroups fit a CU where four did.
struct ShapeHead {
  int32_t face_cnt;
  int32_t tri_cnt;
  int32_t slab_cnt, single_cnt;
  float face[kMaxFaces][4];
  float slab[4][8];
};
static_assert(sizeof(ShapeHead) % 16 == 0, "ShapeHead rows are read as float4");

// A hexagonal prism never has more than 8 faces, 4 opposite-face slabs and 20 fan triangles: stochastic prism pools and
// their LDS copies use this third-size record (same member names, so geometry and trace code are generic over the two).
struct ShapePrism {
  int32_t face_cnt, tri_cnt, slab_cnt, single_cnt;
  float face[8][4];
  float slab[4][8];
  float tri_v[20][9];
  float tri_na[20][4];
  uint8_t tri_face[20];
  uint8_t face_number[8];
  uint8_t single[8];
  uint8_t pad[12];
};
static_assert(sizeof(ShapePrism) % 16 == 0, "ShapePrism rows are read as float4");

// Entry-pick view of a FULL hexagonal prism (8 faces present: slabs (0,1), (2,5), (3,6), (4,7), no single faces, at most 4
// fan triangles per face, triangles grouped face by face), built on the host for one-shape dispatches.  The projected-area
// pick then needs one dot product per SLAB (opposite faces see -d.n and +d.n, only one of them is lit), keeps the four
// products in registers and walks the seven candidate weights branch-free in face order — same uniform, same cumulative
// order, same partial sums as the walk over all faces (adding the unlit face's zero is exact).
constexpr int kEntryFastFaces = 8, kEntryFastTris = 20;
struct EntryFastDev {
  float tri_area[kEntryFastFaces][4];   // areas of the face's fan triangles, zero-padded
  uint32_t tri0n[kEntryFastFaces];      // first fan triangle | count << 8
  float slab_area[4][2];                // total area of {plus face, minus face} of each slab
  float tri_v[kEntryFastTris][12];      // fan-triangle corners in 48-byte rows (three 16-byte reads per pick)
  // REGULAR hexagonal prism (all eight faces, unit normals exactly (0, 0, +-1) and (cos, sin)(i x 60 deg) from the builder's
  // tables, equal side distances): the next-face search then needs no table at all — normals are literals in the instruction
  // stream, the two plane constants below are all that varies.  hex_regular = 0: not such a prism.
  uint32_t hex_regular;
  float hex_d_basal, hex_d_side, hex_pad;
};
static_assert(sizeof(EntryFastDev) % 16 == 0, "copied as float4");

// LDS slot of a general pool shape (the HBM record stays a ShapeDev).  A convex solid with V corners has 2 (V - 2) fan
// triangles whatever its faces look like (Euler), and the pyramid family has at most 24 corners (two hexagonal rings per
// cap) — 44 triangles, 48 rows.  The corner rows (tri_v, 36 B per triangle) are NOT staged: the entry pick reads one
// triangle's nine floats per ray, so they stay in the pool record (HBM / L2, the lines the staging copy has just touched) and the
// slot carries the pointer.  1.5 KB per slot instead of 3.2 KB: a workgroup's eight slots plus its tables take 34 KB, four
// workgroups per CU instead of three, and the compiler targets 128 VGPRs instead of 146 — configs[4p] 2.88 -> 3.19 G rays/s on
// one box.  stage_shape clamps to the slot's capacity like it always did.
struct ShapeSlot48 {
  int32_t face_cnt, tri_cnt, slab_cnt, single_cnt;
  float face[kMaxFaces][4];
  float slab[kMaxSlabs][8];
  const float (*tri_v)[9];   // the record's corner rows (global memory)
  uint64_t pad0;
  float tri_na[48][4];
  uint8_t tri_face[48];
  uint8_t face_number[kMaxFaces];
  uint8_t single[kMaxFaces];
  uint8_t pad[8];
};
static_assert(sizeof(ShapeSlot48) % 16 == 0, "rows are read as float4");

struct WlEntryDev {  // reference WlEntry, src/core/backend/wl_pool.hpp:29-35 (+pad to 32 B)
  float n_idx, spd_weight, cmf_x, cmf_y, cmf_z, pad0, pad1, pad2;
};

struct ProjDev {  // lm_proj::ProjParams (projection_shared.h:106-118), host-predigested
  int32_t proj_type, img_w, img_h, visible_range, lens_shift_x, lens_shift_y;
  float scale, az0, r_scale, max_abs_dz;
  float rot[9];
};

enum : uint32_t { kSrcGen = 0u, kSrcTransit = 1u, kSrcHost = 2u };

// Emit-gate filter as the kernel evaluates it: DeviceFilterDesc (reference src/core/device_filter_desc.hpp:60-100)
// with canonical face-number sequences already symmetry-reduced on the host.
constexpr int kFilterPathCap = HALO_MAX_HITS;  // ExitFaceSeq::kCap
struct FilterTermDev {
  uint8_t type, has_entry, has_exit, canonical_len;
  uint32_t min_len, max_len;
  float dir[3];
  float radii_c;
  uint32_t crystal_id;
  uint8_t canonical[kFilterPathCap];
  // the same canonical sequence packed big-endian and left-aligned into 128 bits (element 0 in the top byte) when it has
  // at most 16 elements: equal-length sequences then compare lexicographically as unsigned integers, in registers
  uint64_t canon_hi, canon_lo;
};
struct FilterDev {
  uint8_t is_complex, action, symmetry, d_applicable;
  int32_t sigma_a;
  uint32_t or_count;
  uint8_t and_counts[HALO_FILTER_MAX_OR];
  FilterTermDev terms[HALO_FILTER_MAX_TERMS];
};

// Raypath-colour tables of one dispatch (see halo_trace.h HaloColorSet / HaloColorClass): predicates are canonicalised on
// the host like filter terms, each with its own symmetry.
struct ColorTermDev {
  FilterTermDev t;
  uint8_t symmetry, d_applicable, bit, pad;
  int32_t sigma_a;
};
struct ColorDev {
  uint32_t term_cnt, class_cnt;
  uint64_t class_bits[HALO_COLOR_MAX_CLASSES];
  uint8_t class_all[HALO_COLOR_MAX_CLASSES];
  ColorTermDev terms[HALO_COLOR_MAX_TERMS];
};

// Fast form of the emit-gate filter and of the raypath-colour predicates for paths of at most 16 faces (max_hits <= 16): what the
// kFilter / kColor production kernels evaluate.  Nothing is symmetry-reduced on the device.  A raypath term matches exactly the
// sequences whose reduction (Crystal::ReduceRaypath, crystal.cpp:536-600) equals the term's canonical form; each of them is one of
// the <= 24 images of that form under the P / B / D group, so the host lists the members (FastTables::orbit: every image whose own
// reduction IS the canonical form — the reference's predicate, including the cases where its reduction is not a perfect orbit
// invariant) and the kernel compares the packed path with them; an entry/exit term becomes a 32 x 32 bit matrix over face numbers
// (row = entry face, bit = exit face).  All of it is dispatch-uniform and read with scalar loads.
constexpr int kFastOrbitCap = 1024;  // members over all raypath terms of a dispatch (filter + colour predicates), each term's list padded to a multiple of 8
constexpr int kFastEeCap = 24;       // entry/exit matrices
struct FastTerm {           // 8 dwords, read with ONE scalar load (s_load_dwordx8)
  uint32_t w0;              // type (HALO_FILTER_*) | last << 8 | bit << 16 | len << 24
                            //   last: 1 = last term of its AND-clause (the clause walk is one flat loop over the terms); bit: colour predicates, the
                            //   mask bit this predicate sets; len: raypath, length of the canonical sequence (255 = longer than any path here)
  uint32_t w1;              // min_len | max_len << 8 (entry/exit; 0 = unbounded above, 255 = beyond any path here) | orbit_n << 16
  uint32_t w2;              // orbit_off | ee_off << 16 — raypath: members in FastTables::orbit_lo / orbit_hi (orbit_n a multiple of 8, padded
                            //   with ~0); entry/exit: matrix in FastTables::ee, 0xFFFF = neither face constrained
  uint32_t crystal_id;
  float dir[3], radii_c;
};
static_assert(sizeof(FastTerm) == 32, "read with one s_load_dwordx8");
struct FastTables {
  // What the filter says about an exit whose path has L faces, two bits per L (bits 2L, 2L+1), folded on the host from everything that
  // is known before the launch — the lengths the path terms can match at all, the dispatch's crystal id: 0 = every such exit fails,
  // 1 = every such exit passes, 2 = evaluate the terms.  (A raypath filter costs nothing at the lengths it cannot match, an all-pass
  // or all-fail filter nothing at all.)  3 = the filter is ONE direction term (the common "mask the sun" filter): its constants ride in this
  // header (dir0, radii0), so the whole evaluation is the header's one scalar load and three FMAs.
  // The first eight dwords are what every emit reads: ONE s_load_dwordx8.
  uint64_t len_mode;
  uint32_t term_cnt, action;
  float dir0[3], radii0;
  uint32_t has_filter, color_terms, class_cnt, pad0;
  uint64_t class_bits[HALO_COLOR_MAX_CLASSES];
  uint32_t class_all[HALO_COLOR_MAX_CLASSES];
  FastTerm fterm[HALO_FILTER_MAX_TERMS + HALO_FILTER_MAX_OR];   // (+ one pass-all term per empty AND-clause)
  FastTerm cterm[HALO_COLOR_MAX_TERMS];
  uint64_t orbit_lo[kFastOrbitCap];   // the path register's layout: newest face in the low byte of lo
  uint64_t orbit_hi[kFastOrbitCap];
  uint32_t ee[kFastEeCap][32];
};

// Continuation pool sharding.  One device counter saturates at ~88 M returning atomics/s (measured, MI355X), and every
// wave appends once per emit site — so the pool is cut into kContShards regions, block b appends to region b % kContShards
// through that region's own counter (64 B apart: same-line atomics serialise), and the next layer reads logical index j
// through the prefix table of the fill counts.  Dense logical order, no holes, no compaction pass.
constexpr int kContShards = 256;
constexpr uint32_t kLogWlShift = 23u;   // DispatchParams::log_xyz records: slot in bits 0..22, CMF code above: a wavelength-pool entry, or pool size + c for a weight that is already channel c of X, Y, Z
constexpr int kContCntStride = 16;

struct HitRec {  // one staged pixel hit of the binned accumulation: slot inside plane 0 and the weight's bits
  uint32_t slot, w_bits;
};

// Per-dispatch constants and tallies as they sit in HBM: one H2D copy of the whole block per dispatch from a pinned
// mirror, taken from a ring so a dispatch can be queued while earlier ones still run (no host sync per launch).
struct DispatchSlot {
  alignas(16) float lut[3 * kLutNodes + 1];
  alignas(16) WlEntryDev wl[HALO_WL_POOL_MAX + 1];
  alignas(16) ShapeDev shape;
  alignas(16) FilterDev filter;
  alignas(16) uint32_t seg[kContShards + 4];
  alignas(16) ColorDev color;
  alignas(16) EntryFastDev efast;
  alignas(16) FastTables fast;
};

// Everything one (layer, crystal-entry) dispatch needs; passed by value as the kernel argument.
struct DispatchParams {
  // --- ray source -------------------------------------------------------------------------
  uint32_t source;       // kSrcGen | kSrcTransit | kSrcHost
  uint32_t n_rays;
  uint32_t layer;
  uint32_t final_layer;
  uint32_t max_hits;     // surface interactions incl. entry (legacy semantics, simulator.cpp:1308)
  float prob;
  uint32_t crystal_id;
  uint32_t capture;
  // --- RNG streams (64-bit counters split lo/hi, trace_backend.hpp:184) ---------------------
  uint32_t gen_seed, gen_lo, gen_hi;
  uint32_t gate_seed, gate_lo, gate_hi;
  uint32_t transit_seed, transit_lo, transit_hi;
  uint32_t shuffle, shuffle_seed;
  uint32_t shuffle_chunk_log2;   // Recombine's shuffle permutes chunks of 2^k consecutive pool entries (option "shuffle_chunk")
  // --- orientation sampler (GenRootKernelParams, pcg_shared.h:150-189) ------------------------
  uint32_t lat_path;
  float lat_mean_rad, lat_std_rad;
  uint32_t az_type;
  float az_mean_rad, az_std_rad;
  uint32_t roll_type;
  float roll_mean_rad, roll_std_rad;
  // --- sun cone: host-evaluated trig of (az+180, -alt, diameter/2) (cu:391-399) ------------------
  float c_cap, c_lon, s_lon, c_lat, s_lat;
  // --- tables ------------------------------------------------------------------------------
  uint32_t wl_pool_size;
  uint32_t shape_cnt;
  uint32_t geom_clock;
  ProjDev proj;
  float proj_pre[4];           // directly behind `proj` (ProjPre in halo_trace.inl): float(img_w) / 2, float(img_h) / 2, float(lens_shift_x), float(lens_shift_y)
  const float* lut;            // theta[257] | cdf[257] | flip[257]
  const WlEntryDev* wl_pool;
  const ShapeDev* shapes;
  const EntryFastDev* entry_fast;   // one-shape dispatch of a full prism: the fast entry pick's tables (else nullptr)
  // --- continuation pools (SoA: dx | dy | dz | w | wl_idx, each cont_stride apart) ---------------
  const float* cont_in;
  uint32_t cont_in_n;
  uint32_t cont_in_stride;
  uint32_t cont_in_region;     // slots per shard region of the input pool
  const uint32_t* cont_in_seg; // [kContShards+1] prefix of the shard fill counts: logical index → (shard, offset)
  uint32_t ci_start;
  float* cont_out;
  uint32_t cont_out_stride;
  uint32_t cont_out_cap;       // slots per shard region of the output pool
  uint32_t* cont_cnt;          // [kContShards * kContCntStride] fill count of each shard region
  uint32_t* counters;          // [0] continuation count, [1] captured exits, [2] exit count lo.. see kCnt*
  // --- host-injected rays (crystal-local) -------------------------------------------------------
  const float* host_d;
  const float* host_p;
  const float* host_w;
  const uint32_t* host_tf;
  // --- outputs -------------------------------------------------------------------------------
  float* mono;                 // scalar plane(s) for discrete-wavelength sessions, folded into xyz at EndSession:
                               // mono_copy_mask+1 copies of kMonoRows << mono_s_log2 floats, pixel p at MonoSlot(p)
  uint32_t mono_s_log2;
  uint32_t mono_copy_mask;
  double* ovf;                 // fp64 twin of the planes' copy 0 (nullptr = none): what a full hit-log region or a full tile list cannot hold is added
                               // HERE, not to the fp32 plane — a hot pixel's overflow is thousands of near-equal addends onto one large float, which
                               // fp32 atomics round the same way every time (5.6e-4 high, round 3); the closing fold takes the twin in when ovf_flag says so
  uint32_t* ovf_flag;          // set to 1 by whoever writes the twin
  uint32_t ovf_copies_log2;    // the twin shadows copy 0 only and is laid out without the other copies: plane p's twin starts at p << (mono_s_log2 + 10) (TwinOffset)
  HitRec* bin_list;             // binned accumulation (nullptr = off): bin_tiles lists of bin_cap {slot, weight} records
  uint32_t bin_cap;
  uint32_t bin_tiles;
  uint32_t bin_shift;          // list of a hit = (slot >> bin_shift) & (bin_tiles - 1): 0 = interleaved tiles, > 0 = contiguous slot ranges (two-level binning)
  uint32_t* bin_cnt;           // list fill counts, kBinCntStride apart
  uint32_t bin_log;            // 1 (production mode, one plane): hit log — bin_list holds one region of bin_cap records per WORKGROUP, a hit
                               // that misses the pixel cache is appended there instead of going out as a global atomic, and the
                               // kernel leaves each region's fill count in bin_cnt[blockIdx.x]
  uint32_t mono_by_wl;         // 1: plane index = the ray's wavelength-pool entry (illuminant session, one plane per entry)
  uint32_t log_xyz;            // hit log of an illuminant session on X, Y, Z planes (with bin_log, X/Y/Z kernels): a record is the hit's slot
                               // in ONE plane | CMF code << kLogWlShift, and the per-tile pass applies the code's CMF row
  uint32_t log_plane_stride;   // ... floats between the X, Y and Z planes (fallback atomics of a full log)
  uint32_t pool_entry_fast;    // prism pools under the hit log: 1 = a sampled FULL prism's entry face is picked slab by slab (halo_trace.inl SlotFast), 0 = the walk over its fan triangles
  uint32_t rehit_legacy;       // 1: the reference's CPU next-face strategy (option rehit_strategy = 0; generic kernels): a child leaving through the face it stands on is propagated, not emitted outright
  uint32_t no_land;            // 1: production-mode layer with prob >= 1 that is not the last — every exit continues, nothing reaches the image
  double* tally;               // kTallyLines lines of kTallyStride doubles, [kSum*] in each: landed weight, exit weight sum, exit count, pixel hits —
                               // cumulative over the backend's life (the host reads differences); a workgroup adds to line blockIdx % kTallyLines
  HaloExitRecord* exits;
  uint32_t exit_cap;
  uint32_t aggregate;          // 0 plain atomics | 1 LDS pixel cache | 2 diagnostic: no accumulation
  const FilterDev* filter;     // nullptr = pass-all
  const ColorDev* color;       // nullptr = no raypath colour: no masks carried, no lanes
  const FastTables* fast;      // kFilter / kColor kernels: the filter and the colour predicates in their fast form (else nullptr)
  double* lanes;               // class_cnt x lane_stride Y lanes, fp64: a hot pixel's lane passes 1e7 and fp32 atomics round a near-constant addend the same way every time (3e-3 low, round 3)
  uint32_t lane_stride;        // W*H
};

// Pixel → slot map of the mono plane.  The plane is kMonoRows rows of S = 2^s_log2 slots; pixel p sits in row p % kMonoRows
// at column hash(p / kMonoRows).  Horizontal neighbours are a whole row (>= 8 KB) apart and vertical neighbours are
// spread by the multiplicative hash, so the pixels of a bright feature never share a cache line and no line of the
// plane is hotter than its hottest single pixel (same-line atomics serialise at ~12 ns each).  Keeping p % kMonoRows as the
// row makes the fold a tiled transpose: coalesced on the plane side (consecutive columns) and on the image side
// (consecutive rows = consecutive pixels).
constexpr uint32_t kMonoRows = 1024u;
constexpr uint32_t kFoldGroup = 64u;            // planes folded per halo_fold_kernel launch (coefficients ride in the kernel argument)
// halo_consumer_composite: what the composite kernels read (server/component_compositor.cpp)
struct CompositeDev {
  uint32_t n_active;          // participating classes, in draw order (ascending z_order, stable)
  uint32_t mode;              // HALO_COMPOSITE_*
  float s;                    // A * display_exposure_scale (dominant / additive)
  float a;                    // A, the self-anchor (painter's alpha)
  float display;              // display_exposure_scale (painter's post-multiplier)
  uint32_t lane[HALO_COLOR_MAX_CLASSES];   // lane index of active class k
  float color[HALO_COLOR_MAX_CLASSES][3];
};

struct FoldCoef {
  float c[kFoldGroup][3];
};
constexpr uint32_t kMonoMul = 0x9E3779B1u;      // odd: a bijection on any 2^k columns
constexpr uint32_t kMonoMulInv = 0x0E8B2F51u;   // kMonoMul * kMonoMulInv == 1 (mod 2^32)
static_assert(static_cast<uint32_t>(kMonoMul * kMonoMulInv) == 1u, "column hash must be invertible");
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t MonoSlot(uint32_t pix, uint32_t s_log2) {
  const uint32_t col = ((pix / kMonoRows) * kMonoMul) & ((1u << s_log2) - 1u);
  return ((pix % kMonoRows) << s_log2) + col;
}

// Offset in the planes' fp64 twin of the plane float at `off` (a copy-0 slot): the planes lie [plane][copy][slot], the twin [plane][slot].
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t TwinOffset(size_t off, uint32_t plane_log2, uint32_t copies_log2) {
  return ((off >> (plane_log2 + copies_log2)) << plane_log2) | (off & ((static_cast<size_t>(1) << plane_log2) - 1u));
}

enum { kCntCont = 0, kCntExit = 1, kCntNum = 4 };
enum { kSumLanded = 0, kSumExitW = 1, kSumExitN = 2, kSumPixN = 3, kSumNum = 4 };
constexpr uint32_t kTallyLines = 16u, kTallyStride = 8u;   // 64-byte lines: same-line fp64 atomics serialise memory-side (~12 ns each)

}  // namespace halo

#endif
