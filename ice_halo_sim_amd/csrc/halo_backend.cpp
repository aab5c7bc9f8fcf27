// halo_backend.cpp — the C ABI of include/halo_trace.h: session state machine, per-(layer, entry) dispatch,
// device memory, streams/events.  Host C++ only; kernels live in halo_trace.inl (+ halo_trace_m*.hip) and halo_kernels.hip.
//
// State machine (reference trace_backend.hpp:91-116):
//   halo_begin → (halo_trace_layer → halo_recombine)* → halo_trace_layer → halo_end ; halo_readback_xyz any time.
// There is no CPU fallback in this library: every entry point that computes needs a gfx950 device and fails
// with HALO_UNAVAILABLE otherwise.
#include <cstdlib>
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "halo_device.h"
#include "halo_host.hpp"

#ifndef HALO_XYZ_LOG_MIN_PIX
#define HALO_XYZ_LOG_MIN_PIX (1u << 19)   // illuminant sessions on images ABOVE this many pixels take the X/Y/Z hit log (below: one scalar plane per pool entry)
#endif
namespace halo {
hipError_t launch_trace(const DispatchParams& P, int blocks, hipStream_t stream, int mode, int geom, bool mono);
hipError_t launch_bin_accumulate(float* plane, const HitRec* list, uint32_t cap, uint32_t* cnt, uint32_t tiles, uint32_t frac_bits, hipStream_t stream);
hipError_t launch_bin_two_level(float* plane, const HitRec* list1, uint32_t cap1, uint32_t* cnt1, uint32_t lists1, HitRec* list2, uint32_t cap2,
                                uint32_t* cnt2, uint32_t tiles, uint32_t fan_log2, uint32_t frac_bits, double* ovf, uint32_t* ovf_flag, hipStream_t stream, hipEvent_t before_sums);
hipError_t launch_log_route(float* plane, const HitRec* log, uint32_t cap1, const uint32_t* cnt1, uint32_t regions, HitRec* list2, uint32_t cap2, uint32_t* cnt2,
                            uint32_t tiles, uint32_t planes, uint32_t s_log2, bool interleaved, uint32_t frac_bits, double* ovf, uint32_t* ovf_flag, uint32_t copies_log2, hipStream_t stream,
                            hipEvent_t before_sums);
hipError_t launch_log_route_xyz(float* planes, uint32_t plane_stride, const HitRec* log, uint32_t cap1, const uint32_t* cnt1, uint32_t regions, HitRec* list2,
                                uint32_t cap2, uint32_t* cnt2, uint32_t tiles, uint32_t s_log2, const WlEntryDev* pool, uint32_t pool_size, uint32_t frac_bits,
                                double* ovf, uint32_t* ovf_flag, uint32_t copies_log2, hipStream_t stream, hipEvent_t before_sums);
hipError_t launch_shapegen(void* pool, bool prism_records, uint32_t n, uint32_t seed, const geom::CrystalRecipe& rc, uint64_t first_index,
                           hipStream_t stream, bool serial_pyramid);
hipError_t launch_fold(float* xyz, float* planes, uint32_t n_pix, uint32_t s_log2, uint32_t copies, uint32_t n_planes, const FoldCoef& coef,
                       double* ovf, const uint32_t* ovf_flag, hipStream_t stream);
hipError_t launch_consumer_fold(float* acc, float* sum, float* comp, uint32_t n, int blocks, hipStream_t stream);
hipError_t launch_lane_hist(const double* lanes, uint32_t n_pix, const CompositeDev& cd, uint32_t shift, uint32_t bits, uint32_t prefix, uint32_t* hist, int blocks,
                            hipStream_t stream);
hipError_t launch_composite(const double* lanes, uint32_t n_pix, const CompositeDev& cd, float* rgb_out, uint8_t* srgb_out, int blocks, hipStream_t stream);
hipError_t launch_lanes_load(const float* src, double* lanes, uint64_t n, int blocks, hipStream_t stream);
hipError_t launch_lanes_drain(double* lanes, float* dst, uint64_t n, int blocks, hipStream_t stream);
hipError_t launch_lanes_add(const float* src, double* lanes, uint64_t n, int blocks, hipStream_t stream);
hipError_t launch_post_snapshot(const float* sum, const float* comp, uint8_t* rgb_out, float* xyz_out, uint32_t n_pix, float scale,
                                const float ray_color[3], const float background[3], int blocks, hipStream_t stream);
}

using namespace halo;

namespace {

template <typename T>
struct DevBuf {
  T* ptr = nullptr;
  size_t cap = 0;  // elements
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), std::max<size_t>(n, 1) * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
};

}  // namespace

struct HaloBackend {
  int device = 0;
  uint32_t seed = 1;
  std::string error;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  int cu_count = 256;

  // options
  int capture = 0;
  uint32_t geom_clock = 32;  // simulator.hpp:144 (rays per sampled shape)
  uint64_t chunk = 1ull << 28;   // rays per launch: configs[2]'s 237 M continuations per wavelength run 3.8 % faster as one launch than as four
  uint64_t stoch_chunk = 0;            // rays per dispatch with device-generated crystal pools (0 = by record size, see chunk_of)
  uint32_t bin_l1 = 128u;              // coarse lists of the two-level binned route (measured: 512 -> 128 lists 6 % faster, 64 slower)
  int aggregate = 1;
  int mono_enabled = 1;
  int bin = -1;                // binned accumulation: -1 auto (discrete session, full-sky render, launch >= 4 Mi rays), 0 off, 1 on
  // Two sets of the hit-log / tile-list buffers, taken in turn by the launches that alternate between the two trace streams (<= 2^21 rays):
  // the one of exactly 2^21 rays that logs may run beside a neighbour.  Launches that fill the chip keep set 0.
  DevBuf<HitRec> bin_list_s[2], bin_list2_s[2];   // one-level tile lists / coarse lists / log regions; tile lists of the two-level and log routes
  DevBuf<uint32_t> bin_cnt_s[2], bin_cnt2_s[2];
  int log_set = 0;                      // the set the next logged launch writes
  bool set_used[2] = {false, false};    // ev_set_free[s] has been recorded: the set's last passes may still be running
  hipEvent_t ev_set_free[2] = {};
  // Auxiliary stream (round 5): the closing folds are queued here, behind everything the other streams hold, and nobody waits for them until
  // somebody reads the image or adds to the planes again — a session's fold (20 us) runs under the next session's first kernels instead of
  // in front of them.  (Two engines side by side on one GPU had measured +10 .. 15 % over one, tools/two_engines_probe.py; what that was,
  // looked at inside one engine, is the small things between kernels — folds, resets, uploads, the tails — not the trace kernels or their
  // passes, which gain nothing from running under each other: see the passes in halo_trace_layer.)
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool aux_pending = false;             // work has been queued on `aux` that the main stream has not waited for
  // Two trace streams: consecutive launches alternate between them, so that launch k+1's workgroups fill the CUs that launch k's last
  // workgroups leave idle (kernels of ONE stream run one after the other: the tail of every 1.8 ms trace kernel, and the whole of a
  // latency-bound small launch, went unshared).  Trace kernels of different launches have nothing to order between them: each logs to its own
  // buffer set or adds with atomics, tallies and continuation appends are atomic.  What they must follow is on the main stream (table
  // uploads) or the auxiliary one (a fold or per-tile pass before a launch that adds to the planes itself): waited for by events.
  hipStream_t cs[2] = {nullptr, nullptr};
  hipEvent_t ev_cs[2] = {nullptr, nullptr}, ev_main = nullptr, ev_aux = nullptr;
  bool cs_pending[2] = {false, false};
  int cs_next = 0;
  uint64_t aux_seq = 0, cs_seen_aux[2] = {0, 0};   // (a trace stream waits for the auxiliary stream only when that holds work it has not waited for yet)
  int overlap = 1;                      // option: 0 queues the passes and folds on the main stream (round 4 behaviour)
  int defer_fold = 0;                   // option: 1 = halo_end leaves the fold pending even when the accumulator is the caller's memory; the caller asks for
                                        // it (halo_sync, or any reader) before it looks at its memory
  int twin_set = 0;                     // which half of the fp64 twin the open / next session's overflows go to (the other half may still be folding)
  int lambda_planes = -1;      // illuminant sessions: -1 auto (by batch size), 0 never, 1 always one plane per pool entry
  int mono_copies = 8;         // power of two; copy = blockIdx & (copies-1)
  uint32_t mono_s_log2 = 0;    // log2 of the columns per row of the plane (kMonoRows rows; see MonoSlot)
  int blocks_per_cu = 24;      // cap on workgroups per CU of a launch (5 resident: several rounds even out the tail)
  int host_shapes = 0;         // 1: stochastic shape pools are built on the host and uploaded (A/B and test path)
  int gen_serial = 0;          // 1: pyramids are generated one thread per crystal (the serial builder) instead of one team of 32 lanes
  int small_blocks_per_cu = 0; // experiment knob: workgroups per CU that a small launch may spread over (0 = five, see blocks_of)
  int blocks_cap = 0;          // experiment knob: absolute cap on the workgroups of one launch (0 = none)
  int lazy_fold = 1;           // the closing fold of a session waits for the first READER of the image (or a session with other planes) when the accumulator is the backend's own
  int hit_log = -1;            // hit-log accumulation of cache misses: -1 auto (one-plane sessions, launches >= 2 Mi rays), 0 off, 1 on
  uint32_t hit_log_cap = 0;    // test knob: records per log region (0 = sized from the launch); the tile lists then get half their even share
  int hex_fast = 1;            // 1: regular hexagonal prisms of one-shape dispatches run the literal-normal next-face search
  int entry_fast = 1;          // 1: full prisms of one-shape dispatches take the slab-wise entry pick (EntryFastDev)
  int filter_fast = 1;         // 1: filtered / colour-tagged dispatches with max_hits <= 16 run the production-shaped kernels (FastTables); 0: the generic ones
  int async = 0;               // 1: final-layer dispatches are queued without a host sync; stats via halo_collect_stats
  uint32_t shuffle_chunk_log2 = 5;   // Recombine's shuffle moves chunks of 2^k pool entries (k = 0: per ray, like the reference)
  HaloRouteInfo route{};       // kernels that served the current / last session (halo_last_route)

  // monotone ray counters: seeded once, never reset per session (cuda_trace_backend.cu:3724-3741)
  uint64_t gen_count = 0, gate_count = 0, transit_count = 0, shape_count = 0;
  double sess_max_w = 1.0;     // largest initial ray weight of the open session (wavelength-pool spd weights)

  // session
  bool in_session = false;
  HaloScene scene{};
  HaloRender render{};
  HaloWl wl{};
  ProjDev proj{};
  int layer_idx = 0;
  double carry[HALO_MAX_LAYERS][HALO_MAX_ENTRIES] = {};

  // device state
  DevBuf<float> acc_own;       // W*H*3 + 4
  float* acc = nullptr;        // bound accumulator (own or external)
  uint64_t acc_floats = 0;
  int acc_w = 0, acc_h = 0;    // image the bound accumulator (own or external) currently holds
  int own_w = 0, own_h = 0;    // image the OWNED buffer was last sized and zeroed for
  std::vector<HaloFilter> filters;  // table referenced by HaloEntry::filter_id
  std::vector<HaloColorSet> color_sets;      // table referenced by HaloEntry::color_id
  std::vector<HaloColorClass> color_classes; // raypath-colour classes (Y lanes)
  DevBuf<double> lanes;                      // class_count x W x H, accumulated in fp64 (handed out as float)
  int lanes_w = 0, lanes_h = 0;
  DevBuf<FilterDev> filter_dev;
  std::unique_ptr<FastTables> fast_scratch;  // host staging of a dispatch's fast filter tables (20 KB: not on the stack)
  DevBuf<float> mono;          // accumulation planes (see MonoSlot): plane_cnt x plane_copies x (kMonoRows << s_log2) floats
  DevBuf<double> ovf;          // fp64 twin of the planes' copy 0, laid out [plane][slot] (TwinOffset): where full log regions / tile lists overflow to
  DevBuf<uint32_t> ovf_flag;   // one word: something was written to the twin since the last fold
  bool mono_session = false;   // kernel variant: true = one scalar per hit (plane 0 or plane wl_idx), false = X,Y,Z planes
  bool mono_by_wl = false;     // illuminant session with one plane per wavelength-pool entry
  bool xyz_log = false;        // illuminant session on X, Y, Z planes whose big production launches go through the hit log
  bool mono_dirty = false;
  // Two plane SETS for sessions whose every launch goes through the hit log onto one scalar plane (round 6): a fold reads and zeroes the set its
  // session added to while the passes of the session after it add to the other set — on a discrete spectrum of short sessions the passes of
  // session k + 1 otherwise wait for the fold of session k, which waits for the passes of session k: one serial chain (bench.py --config 4d).
  bool mono_two = false;
  int mono_set = 0;
  size_t mono_half = 0;                                   // floats per set
  hipEvent_t ev_set_folded[2] = {nullptr, nullptr};      // behind the fold that last zeroed the set
  bool set_folded_pending[2] = {false, false};
  uint32_t plane_cnt = 1, plane_copies = 8;
  std::vector<std::array<float, 3>> plane_coef;  // fold coefficients of the pending planes
  // consumer (RenderConsumer state, server/render.hpp): Neumaier running image + total landed intensity
  DevBuf<float> cons_sum, cons_comp, cons_xyz_out, cons_stage;
  DevBuf<uint8_t> cons_rgb;
  DevBuf<float> comp_rgb, lanes_stage;   // halo_consumer_composite's linear image, halo_consumer_load_lanes' staging
  DevBuf<uint32_t> comp_hist;            // radix-select histogram (2048 bins)
  int cons_w = 0, cons_h = 0;
  double total_intensity = 0.0;
  DevBuf<double> tally;        // kTallyLines x kTallyStride doubles, [kSum*] per line: CUMULATIVE tallies of every kernel this backend ever launched
  double* tally_host = nullptr;       // hipHostMalloc, one snapshot of `tally`
  double tally_seen[kSumNum] = {};    // cumulative values already handed on: [kSumLanded] to a readback / take_landed, the rest to the layer statistics
  bool tally_unread = false;
  double time_trace_ms = 0.0, time_post_ms = 0.0;   // halo_collect_timing: the last layers' trace kernels' own spans and their passes', since the last call
  uint64_t time_launches = 0;          // a dispatch has been queued since the statistics were last pulled
  DevBuf<uint32_t> counters;   // kCntNum
  // dispatch ring: device slots, pinned host mirrors, pinned tally read-back, per-slot events
  static constexpr int kRing = 32;
  DevBuf<DispatchSlot> ring_dev;
  DispatchSlot* ring_host = nullptr;   // hipHostMalloc
  hipEvent_t ring_ev0[kRing] = {}, ring_ev1[kRing] = {}, ring_ev2[kRing] = {}, ring_ev3[kRing] = {}, ring_done[kRing] = {};   // ev0..ev1 the trace kernel (main stream), ev2..ev3 its passes (post stream)
  bool ring_posts[kRing] = {};
  bool ring_final[kRing] = {};         // the slot's launch belongs to its session's last layer (halo_collect_timing counts those)
  bool ring_busy[kRing] = {};
  uint64_t ring_use[kRing] = {};       // how many launches have taken the slot (TableCacheEntry::read_use)
  int ring_next = 0;
  // Table cache (round 5): the dispatch-constant tables of a deterministic crystal entry (latitude LUT, wavelength pool, shape, entry-pick
  // tables, filter / colour tables) stay on the device from one dispatch to the next while scene, wavelength, filters, colour and options
  // are what they were — a server that sends one wavelength's rays as many equal small sessions (and every chunk of a long layer) then
  // pays neither the host-side builds (closed-form geometry, LUT, predicate tables) nor the 25-49 KB copy per dispatch.
  struct TableCacheEntry {
    bool valid = false;
    int mode = 0;
    bool use_filter = false, use_color = false, entry_fast = false, hex_regular = false, has_fast = false;
    // Two device slots per entry, taken in turn by the uploads: the kernels of the session before (queued on a trace stream, async) may still be
    // staging their tables out of the slot they were given.  An upload waits (on the stream, not the host) for the last kernel that read ITS
    // slot — two table sets back, as good as always finished — so consecutive sessions of different wavelengths still overlap.
    int cur = 0;
    // the slot's last reader = the dispatch-ring slot of the last launch that was given it (its ring_done event is recorded anyway: no
    // event of the cache's own, one HIP call less per small session) and that ring slot's use count then: a ring slot that has been taken
    // again since was harvested first, i.e. that launch is over
    int read_ring[2] = {-1, -1};
    uint64_t read_use[2] = {0, 0};
  };
  TableCacheEntry tcache[HALO_MAX_LAYERS][HALO_MAX_ENTRIES];
  DevBuf<DispatchSlot> tcache_dev;     // HALO_MAX_LAYERS x HALO_MAX_ENTRIES x 2 slots, reserved on first use
  HaloScene tcache_scene{};            // what the valid entries were built for
  HaloWl tcache_wl{};
  int table_cache = 1;                 // option: 0 uploads the tables with every dispatch (round 4 behaviour)
  std::map<std::string, int64_t> opt_last;   // halo_set_option: the value each key was last set to (an unchanged option keeps the table cache)
  HaloLayerStats pending{};    // harvested tallies not yet handed to the caller (async mode)
  HaloLayerStats layer_acc{};  // tallies of the layer being traced
  std::vector<WlEntryDev> wl_pool_host;
  uint32_t wl_pool_size = 0;
  struct LutKey { int32_t type; float center, spread; host::LatLut lut; };
  std::vector<LutKey> lut_cache;
  DevBuf<ShapeDev> shapes_s[2];   // sampled-crystal pools: launch k+1's generator must not overwrite what launch k still traces
  hipEvent_t ev_shapes_free[2] = {nullptr, nullptr};   // recorded behind the trace kernel that read the pool
  hipEvent_t ev_gen = nullptr;                         // the generator of a chip-filling launch, queued on trace stream 1 (see gen_ahead)
  bool shapes_used[2] = {false, false};
  int shapes_next = 0;
  // Plane ownership between the two trace streams.  The per-tile sums of a logged (or two-level binned) launch end in PLAIN read-modify-writes of
  // the planes; everything else that adds to them uses atomics.  So (1) a launch's sums wait for what the OTHER trace stream held when the launch
  // was queued (ev_gate), and (2) a later launch on the other stream whose trace kernel adds to the planes itself waits for those sums (ev_rmw).
  // Logged trace kernels touch no plane (their overflow goes to the fp64 twin), so they run beside a neighbour's passes freely.
  hipEvent_t ev_gate = nullptr, ev_rmw[2] = {nullptr, nullptr};
  bool rmw_pending[2] = {false, false};
  int log_tiles_log2 = 7;         // option (experiment knob): the scalar hit-log route cuts a plane into up to 2^this tiles (<= 8)
  int alt_log2 = 25;              // option (experiment knob): launches of up to 2^alt_log2 rays alternate between the two trace streams
  int rehit_strategy = 1;         // option: 1 [default] the reference's CUDA next-face strategy (the child that leaves through the face it stands on is emitted where it is);
                                  // 0 its legacy CPU strategy (that child is propagated over all faces with the relaxed accept threshold, optics.cpp:116-155 —
                                  // SURVEY's stated ground truth): every launch then takes the generic kernels, which carry the re-hit test
  int pool_entry_fast = 1;        // option (A/B knob): prism pools under the hit log pick a sampled full prism's entry face slab by slab; 0 = the walk over its fan triangles
  int gen_ahead = 0;              // option (experiment knob): 1 queues the generator of a chip-filling launch on trace stream 1, beside the previous launch's kernels
  DevBuf<float> cont[2];       // SoA continuation pools, 5 planes
  uint32_t cont_stride[2] = {0, 0};
  uint32_t cont_region[2] = {0, 0};          // slots per shard region
  DevBuf<uint32_t> cont_cnt;                 // kContShards * kContCntStride
  uint32_t cont_seg[kContShards + 4] = {};   // prefix of the input pool's shard fill counts
  int cont_out_slot = 0;
  uint64_t cont_in_n = 0;
  int cont_shuffle = 1;
  DevBuf<HaloExitRecord> exits;
  uint64_t exits_pending = 0;
  DevBuf<float> host_f;        // injected rays: d | p | w
  DevBuf<uint32_t> host_u;
  uint64_t sess_crystal_samples = 0, sess_orient_samples = 0;  // this session's stochastic draws (halo_last_sample_counts)
};

namespace {

int fail(HaloBackend* b, int code, const std::string& msg) {
  if (b) b->error = msg;
  return code;
}
int hip_fail(HaloBackend* b, hipError_t e, const char* what) {
  return fail(b, (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorOutOfMemory) ? HALO_UNAVAILABLE : HALO_FATAL,
              std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(b, call)                                   \
  do {                                                    \
    hipError_t e__ = (call);                              \
    if (e__ != hipSuccess) return hip_fail(b, e__, #call); \
  } while (0)

// The plane set the open / pending session adds to (HaloBackend::mono_two).
inline float* planes_of(HaloBackend* b) { return b->mono.ptr + (b->mono_two ? static_cast<size_t>(b->mono_set) * b->mono_half : 0u); }
// The stream the plane-touching passes of logged launches and the folds are queued on.
inline hipStream_t post_stream(HaloBackend* b) { return b->overlap ? b->aux : b->stream; }
// What is queued on the auxiliary stream from here on runs behind everything the main stream and the trace streams hold now.
int fork_aux(HaloBackend* b) {
  if (!b->overlap) return HALO_OK;
  HIPCHK(b, hipEventRecord(b->ev_fork, b->stream));
  HIPCHK(b, hipStreamWaitEvent(b->aux, b->ev_fork, 0));
  for (int k = 0; k < 2; k++)
    if (b->cs_pending[k]) {
      HIPCHK(b, hipEventRecord(b->ev_cs[k], b->cs[k]));
      HIPCHK(b, hipStreamWaitEvent(b->aux, b->ev_cs[k], 0));
    }
  b->aux_pending = true;
  b->aux_seq++;
  return HALO_OK;
}
// What the main stream is given from here on runs behind everything queued on the auxiliary and the trace streams (no host wait).
int join_aux(HaloBackend* b) {
  if (b->aux_pending) {
    HIPCHK(b, hipEventRecord(b->ev_join, b->aux));
    HIPCHK(b, hipStreamWaitEvent(b->stream, b->ev_join, 0));
    b->aux_pending = false;
  }
  for (int k = 0; k < 2; k++)
    if (b->cs_pending[k]) {
      HIPCHK(b, hipEventRecord(b->ev_cs[k], b->cs[k]));
      HIPCHK(b, hipStreamWaitEvent(b->stream, b->ev_cs[k], 0));
      b->cs_pending[k] = false;
      b->rmw_pending[k] = false;   // (every later launch waits for the main stream, which now waits for these sums)
    }
  return HALO_OK;
}
// Trace stream i goes on behind what the auxiliary stream holds now (a closing fold: it reads and zeroes the planes).
int wait_for_aux(HaloBackend* b, int i) {
  if (b->overlap && b->aux_pending && b->cs_seen_aux[i] != b->aux_seq) {
    HIPCHK(b, hipEventRecord(b->ev_aux, b->aux));
    HIPCHK(b, hipStreamWaitEvent(b->cs[i], b->ev_aux, 0));
    b->cs_seen_aux[i] = b->aux_seq;
  }
  return HALO_OK;
}
// The stream the next trace kernel goes to: behind what the main stream holds now (table uploads, counter resets) and — for a launch that
// adds to the planes itself — behind what the auxiliary stream holds (folds, per-tile passes).
int next_trace_stream(HaloBackend* b, bool touches_planes, bool alternate, hipStream_t* out, int* which) {
  if (!b->overlap) {
    *out = b->stream;
    *which = 0;
    return HALO_OK;
  }
  // Launches alternate streams only while they are latency-bound (up to 2^21 rays: tools/dispatch_probe.cpp, us per session with / without,
  // 2^16 18.8 / 30.6, 2^18 25.2 / 44.0, 2^20 65.5 / 110, 2^22 264 / 250, 2^24 817 / 820); a launch that fills the chip by itself gains nothing
  // from a neighbour and keeps stream 0, where its HIP events time it undisturbed.
  const int i = alternate ? b->cs_next : 0;
  if (alternate) b->cs_next ^= 1;
  HIPCHK(b, hipEventRecord(b->ev_main, b->stream));
  HIPCHK(b, hipStreamWaitEvent(b->cs[i], b->ev_main, 0));
  if (touches_planes)
    if (int rc = wait_for_aux(b, i)) return rc;
  b->cs_pending[i] = true;
  *out = b->cs[i];
  *which = i;
  return HALO_OK;
}
// Smallest launch the hit log takes by itself (option hit_log = -1).  Direct atomics retire memory-side at ~21 G/s whatever they hit; the log's
// passes cost a fixed ~75 us per launch plus ~10 ps per record, and on a small launch they run beside the next session's trace kernel.  A
// full-sky render lands every exit (5-6 per root: 0.8 M roots = 4.4 M atomics = 210 us against 102 us of logged trace kernel, bench.py --config 4d),
// a lens that sees part of the sky 1-2 per root.
inline uint64_t log_min_rays(int visible) { return visible == HALO_VISIBLE_FULL ? (1ull << 19) : (2ull << 20); }
// Host waits for all streams.
int sync_all(HaloBackend* b) {
  if (int rc = join_aux(b)) return rc;
  HIPCHK(b, hipStreamSynchronize(b->stream));
  return HALO_OK;
}
// DevBuf::reserve frees the old allocation when it grows: nothing on either stream may still be using it.
template <typename T>
int reserve_idle(HaloBackend* b, DevBuf<T>& buf, size_t n, hipError_t* err = nullptr) {
  if (n > buf.cap) {
    if (int rc = sync_all(b)) return rc;
  }
  const hipError_t e = buf.reserve(n);
  if (err) {
    *err = e;
    return HALO_OK;
  }
  HIPCHK(b, e);
  return HALO_OK;
}

// The cumulative landed tally up to here belongs to the OWNED image that is being left (re-sized: its light is discarded with it): whoever
// reads the new image must not be handed weight that landed in the old one and was never taken.
int take_landed_delta(HaloBackend* b, double* landed);
int forget_landed(HaloBackend* b) {
  if (!b->tally.ptr) return HALO_OK;
  if (int rc = join_aux(b)) return rc;
  double unused = 0.0;
  return take_landed_delta(b, &unused);
}

int ensure_accumulator(HaloBackend* b, int w, int h) {
  const uint64_t need = static_cast<uint64_t>(w) * h * 3 + 4;
  if (b->acc && b->acc != b->acc_own.ptr) {  // external binding
    if (b->acc_floats < need) return fail(b, HALO_FATAL, "bound accumulator smaller than width*height*3+4 floats");
    b->acc_w = w;
    b->acc_h = h;
    return HALO_OK;
  }
  // own_w/own_h describe the OWNED buffer's image (acc_w/acc_h are also written by the external-binding path above, so they
  // cannot decide whether the owned buffer fits); capacity is checked on its own
  if (!b->acc_own.ptr || b->acc_own.cap < need || b->own_w != w || b->own_h != h) {
    if (int rc = reserve_idle(b, b->acc_own, need)) return rc;
    if (int rc = join_aux(b)) return rc;   // (a fold of the old image may still be queued there)
    if (b->own_w != 0 || b->own_h != 0)
      if (int rc = forget_landed(b)) return rc;   // the old image goes, and its landed weight with it
    HIPCHK(b, hipMemsetAsync(b->acc_own.ptr, 0, need * sizeof(float), b->stream));
    b->own_w = w;
    b->own_h = h;
  }
  b->acc_w = w;
  b->acc_h = h;
  b->acc = b->acc_own.ptr;
  b->acc_floats = need;
  return HALO_OK;
}

// take the timing of one finished ring slot (the tallies are cumulative device counters: pull_tally)
void harvest_slot(HaloBackend* b, int k) {
  if (!b->ring_busy[k]) return;
  (void)hipEventSynchronize(b->ring_done[k]);
  float ms = 0.0f, ms2 = 0.0f;
  (void)hipEventElapsedTime(&ms, b->ring_ev0[k], b->ring_ev1[k]);
  if (b->ring_posts[k]) (void)hipEventElapsedTime(&ms2, b->ring_ev2[k], b->ring_ev3[k]);   // the passes' own span
  static const bool debug_timing = std::getenv("HALO_DEBUG_TIMING") != nullptr;
  if (debug_timing) std::fprintf(stderr, "[halo timing] slot %d final %d posts %d trace %.4f ms passes %.4f ms\n", k, int(b->ring_final[k]), int(b->ring_posts[k]), ms, ms2);
  // An event pair that reads backwards (seen once: two twelve-launch runs of one bench process summed to -10 ms; the rocprofv3 trace of the same
  // command had every kernel in order) is a measurement that did not happen: it is left out of the timing sums and of their launch count.
  const bool timed = ms >= 0.0f && ms2 >= 0.0f;
  if (timed) b->layer_acc.kernel_ms += ms + ms2;
  if (b->ring_final[k] && timed) {
    b->time_trace_ms += ms;
    b->time_post_ms += ms2;
    b->time_launches += 1;
  }
  b->layer_acc.launches += 1;
  b->ring_busy[k] = false;
}
// Snapshot of the cumulative tallies (one 1 KB D2H copy + a stream sync): the sum over the lines, per tally.
int read_tally(HaloBackend* b, double out[kSumNum]) {
  HIPCHK(b, hipMemcpyAsync(b->tally_host, b->tally.ptr, kTallyLines * kTallyStride * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(b, hipStreamSynchronize(b->stream));
  for (int t = 0; t < kSumNum; t++) {
    double v = 0.0;
    for (uint32_t l = 0; l < kTallyLines; l++) v += b->tally_host[l * kTallyStride + static_cast<uint32_t>(t)];
    out[t] = v;
  }
  return HALO_OK;
}
// What the kernels have tallied since the statistics were last pulled goes to layer_acc (exit weight, exit count, pixel hits).
int pull_tally(HaloBackend* b) {
  if (!b->tally_unread) return HALO_OK;
  double now[kSumNum];
  int rc = read_tally(b, now);
  if (rc != HALO_OK) return rc;
  b->layer_acc.exit_w_sum += now[kSumExitW] - b->tally_seen[kSumExitW];
  b->layer_acc.exit_count += static_cast<uint64_t>(now[kSumExitN] - b->tally_seen[kSumExitN] + 0.5);
  b->layer_acc.pixel_hits += static_cast<uint64_t>(now[kSumPixN] - b->tally_seen[kSumPixN] + 0.5);
  for (int t : {kSumExitW, kSumExitN, kSumPixN}) b->tally_seen[t] = now[t];
  b->tally_unread = false;
  return HALO_OK;
}
// The landed weight since the last taker (readback, take_landed, consumer fold).
int take_landed_delta(HaloBackend* b, double* landed) {
  double now[kSumNum];
  int rc = read_tally(b, now);
  if (rc != HALO_OK) return rc;
  *landed = now[kSumLanded] - b->tally_seen[kSumLanded];
  b->tally_seen[kSumLanded] = now[kSumLanded];
  return HALO_OK;
}
void harvest_all(HaloBackend* b) {
  for (int k = 0; k < HaloBackend::kRing; k++) harvest_slot(b, k);
}
void add_stats(HaloLayerStats& dst, const HaloLayerStats& src) {
  dst.root_count += src.root_count;
  dst.exit_count += src.exit_count;
  dst.continuation_count += src.continuation_count;
  dst.exit_w_sum += src.exit_w_sum;
  dst.kernel_ms += src.kernel_ms;
  dst.pixel_hits += src.pixel_hits;
  dst.launches += src.launches;
}
void drop_table_cache(HaloBackend* b) {
  for (auto& layer : b->tcache)
    for (auto& e : layer) e.valid = false;
}
const host::LatLut& cached_lut(HaloBackend* b, const HaloDist& d) {
  for (const auto& e : b->lut_cache)
    if (e.type == d.type && e.center == d.center && e.spread == d.spread) return e.lut;
  if (b->lut_cache.size() >= 64) b->lut_cache.clear();
  b->lut_cache.push_back({d.type, d.center, d.spread, host::BuildLatLut(d)});
  return b->lut_cache.back().lut;
}

}  // namespace

extern "C" {

static int fold_if_dirty(HaloBackend* b);
static int fold_queue(HaloBackend* b);
int halo_abi_version(void) { return HALO_ABI_VERSION; }

int halo_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int halo_create(int device_ordinal, uint32_t seed, halo_handle_t* out) {
  if (!out) return HALO_FATAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n) return HALO_UNAVAILABLE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return HALO_UNAVAILABLE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return HALO_UNAVAILABLE;  // this library carries gfx950 code only
  auto* b = new HaloBackend();
  b->device = device_ordinal;
  b->seed = seed ? seed : 1u;
  b->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipSetDevice(device_ordinal) != hipSuccess || hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete b;
    return HALO_UNAVAILABLE;
  }
  bool aux_ok = hipStreamCreateWithFlags(&b->aux, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&b->cs[0], hipStreamNonBlocking) == hipSuccess &&
                hipStreamCreateWithFlags(&b->cs[1], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&b->ev_cs[0], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&b->ev_cs[1], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&b->ev_main, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&b->ev_aux, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&b->ev_set_free[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&b->ev_set_free[1], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&b->ev_shapes_free[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&b->ev_shapes_free[1], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&b->ev_gen, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&b->ev_gate, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&b->ev_rmw[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&b->ev_rmw[1], hipEventDisableTiming) == hipSuccess;
  bool ring_ok = aux_ok && b->ring_dev.reserve(HaloBackend::kRing) == hipSuccess &&
                 hipHostMalloc(reinterpret_cast<void**>(&b->ring_host), HaloBackend::kRing * sizeof(DispatchSlot), hipHostMallocDefault) == hipSuccess &&
                 hipHostMalloc(reinterpret_cast<void**>(&b->tally_host), kTallyLines * kTallyStride * sizeof(double), hipHostMallocDefault) == hipSuccess;
  for (int k = 0; ring_ok && k < HaloBackend::kRing; k++)
    ring_ok = hipEventCreate(&b->ring_ev0[k]) == hipSuccess && hipEventCreate(&b->ring_ev1[k]) == hipSuccess && hipEventCreate(&b->ring_ev2[k]) == hipSuccess && hipEventCreate(&b->ring_ev3[k]) == hipSuccess &&
              hipEventCreateWithFlags(&b->ring_done[k], hipEventDisableTiming) == hipSuccess;
  if (!ring_ok) {
    halo_destroy(b);
    return HALO_UNAVAILABLE;
  }
  b->stream = b->own_stream;
  if (b->tally.reserve(kTallyLines * kTallyStride) != hipSuccess || b->counters.reserve(kCntNum) != hipSuccess) {
    halo_destroy(b);  // releases the stream, the ring, the pinned mirrors and the events as well
    return HALO_UNAVAILABLE;
  }
  (void)hipMemsetAsync(b->tally.ptr, 0, kTallyLines * kTallyStride * sizeof(double), b->stream);
  (void)hipMemsetAsync(b->counters.ptr, 0, kCntNum * sizeof(uint32_t), b->stream);
  *out = b;
  return HALO_OK;
}

int halo_destroy(halo_handle_t b) {
  if (!b) return HALO_OK;
  (void)hipSetDevice(b->device);
  (void)hipStreamSynchronize(b->stream);
  if (b->aux) (void)hipStreamSynchronize(b->aux);
  for (int k = 0; k < 2; k++)
    if (b->cs[k]) (void)hipStreamSynchronize(b->cs[k]);
  b->acc_own.release();
  b->tally.release();
  b->mono.release();
  b->ovf.release();
  b->ovf_flag.release();
  b->filter_dev.release();
  b->cons_sum.release();
  b->cons_comp.release();
  b->cons_xyz_out.release();
  b->cons_stage.release();
  b->cons_rgb.release();
  b->comp_rgb.release();
  b->lanes_stage.release();
  b->comp_hist.release();
  b->counters.release();
  b->ring_dev.release();
  b->tcache_dev.release();
  if (b->ring_host) (void)hipHostFree(b->ring_host);
  if (b->tally_host) (void)hipHostFree(b->tally_host);
  for (int k = 0; k < HaloBackend::kRing; k++) {
    if (b->ring_ev0[k]) (void)hipEventDestroy(b->ring_ev0[k]);
    if (b->ring_ev1[k]) (void)hipEventDestroy(b->ring_ev1[k]);
    if (b->ring_ev2[k]) (void)hipEventDestroy(b->ring_ev2[k]);
    if (b->ring_ev3[k]) (void)hipEventDestroy(b->ring_ev3[k]);
    if (b->ring_done[k]) (void)hipEventDestroy(b->ring_done[k]);
  }
  b->shapes_s[0].release();
  b->shapes_s[1].release();
  b->cont_cnt.release();
  b->lanes.release();
  for (int k = 0; k < 2; k++) {
    b->bin_list_s[k].release();
    b->bin_cnt_s[k].release();
    b->bin_list2_s[k].release();
    b->bin_cnt2_s[k].release();
    if (b->ev_set_free[k]) (void)hipEventDestroy(b->ev_set_free[k]);
    if (b->ev_shapes_free[k]) (void)hipEventDestroy(b->ev_shapes_free[k]);
    if (b->ev_rmw[k]) (void)hipEventDestroy(b->ev_rmw[k]);
  }
  for (hipEvent_t ev : b->ev_set_folded)
    if (ev) (void)hipEventDestroy(ev);
  if (b->ev_gen) (void)hipEventDestroy(b->ev_gen);
  if (b->ev_gate) (void)hipEventDestroy(b->ev_gate);
  if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
  if (b->ev_join) (void)hipEventDestroy(b->ev_join);
  if (b->aux) (void)hipStreamDestroy(b->aux);
  for (int k = 0; k < 2; k++) {
    if (b->ev_cs[k]) (void)hipEventDestroy(b->ev_cs[k]);
    if (b->cs[k]) (void)hipStreamDestroy(b->cs[k]);
  }
  if (b->ev_main) (void)hipEventDestroy(b->ev_main);
  if (b->ev_aux) (void)hipEventDestroy(b->ev_aux);
  b->cont[0].release();
  b->cont[1].release();
  b->exits.release();
  b->host_f.release();
  b->host_u.release();
  if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
  delete b;
  return HALO_OK;
}

const char* halo_last_error(halo_handle_t b) { return b ? b->error.c_str() : "null handle"; }

int halo_set_option(halo_handle_t b, const char* key, int64_t v) {
  if (!b || !key) return HALO_FATAL;
  const std::string k(key);
  {  // most options shape the tables or the kernels that read them: a CHANGED option drops the table cache.  A caller that sets its options before
     // every session (the Lumice glue does) sets them to what they are: the cache stays.  Counter / seed options always act (they move state).
    auto it = b->opt_last.find(k);
    const bool same = it != b->opt_last.end() && it->second == v && k != "rank" && k != "ray_base" && k != "seed";
    if (!same) drop_table_cache(b);
    b->opt_last[k] = v;
  }
  // the session's plane layout (privatised copies, hit-log eligibility) was decided at halo_begin from these two: changing them with a session
  // open would send direct atomics to a layout made for another route
  if ((k == "capture_exits" || k == "filter_fast") && b->in_session) return fail(b, HALO_FATAL, k + " cannot change inside a session");
  if (k == "capture_exits") b->capture = v ? 1 : 0;
  else if (k == "geom_clock") b->geom_clock = static_cast<uint32_t>(v > 0 ? v : 32);
  else if (k == "chunk") {  // the kernels' grid-stride index is 32-bit: n_rays + one stride of workgroups must stay below 2^32
    const uint64_t top = (1ull << 32) - (static_cast<uint64_t>(b->cu_count) * 64ull * kBlock) - kBlock;
    b->chunk = std::min<uint64_t>(static_cast<uint64_t>(v > 0 ? v : (1ll << 28)), top);
  }
  else if (k == "bin_l1") {  // the kernels index the coarse lists with `& (lists - 1)`: a power of two in [8, 512]
    if (v < 8 || v > 512 || (v & (v - 1)) != 0) return fail(b, HALO_FATAL, "bin_l1 must be a power of two in [8, 512]");
    b->bin_l1 = static_cast<uint32_t>(v);
  }
  else if (k == "stoch_chunk") b->stoch_chunk = static_cast<uint64_t>(v > 0 ? v : 0);
  else if (k == "aggregate") b->aggregate = static_cast<int>(v);
  else if (k == "mono") b->mono_enabled = v ? 1 : 0;
  else if (k == "async") b->async = v ? 1 : 0;
  else if (k == "lambda_planes") b->lambda_planes = static_cast<int>(v);
  else if (k == "bin") b->bin = static_cast<int>(v);
  else if (k == "host_shapes") b->host_shapes = v ? 1 : 0;
  else if (k == "entry_fast") b->entry_fast = v ? 1 : 0;
  else if (k == "hex_fast") b->hex_fast = v ? 1 : 0;
  else if (k == "filter_fast") b->filter_fast = v ? 1 : 0;
  else if (k == "hit_log") b->hit_log = static_cast<int>(v);
  else if (k == "hit_log_cap") b->hit_log_cap = static_cast<uint32_t>(std::max<int64_t>(v, 0));
  else if (k == "gen_serial") b->gen_serial = v ? 1 : 0;
  else if (k == "shuffle_chunk") {
    if (v < 1 || v > 64 || (v & (v - 1)) != 0) return fail(b, HALO_FATAL, "shuffle_chunk must be a power of two in [1, 64]");
    b->shuffle_chunk_log2 = 0;
    while ((1ll << b->shuffle_chunk_log2) < v) b->shuffle_chunk_log2++;
  }
  else if (k == "lazy_fold") b->lazy_fold = v ? 1 : 0;
  else if (k == "table_cache") b->table_cache = v ? 1 : 0;
  else if (k == "overlap") {
    if (int rc = sync_all(b)) return rc;
    b->overlap = v ? 1 : 0;
  }
  else if (k == "defer_fold") b->defer_fold = v ? 1 : 0;
  else if (k == "gen_ahead") b->gen_ahead = v ? 1 : 0;
  else if (k == "pool_entry_fast") b->pool_entry_fast = v ? 1 : 0;
  else if (k == "rehit_strategy") {
    if (b->in_session) return fail(b, HALO_FATAL, "rehit_strategy cannot change inside a session");
    b->rehit_strategy = v ? 1 : 0;
  }
  else if (k == "log_tiles_log2") b->log_tiles_log2 = static_cast<int>(std::min<int64_t>(std::max<int64_t>(v, 0), 8));
  else if (k == "alt_log2") b->alt_log2 = static_cast<int>(std::min<int64_t>(std::max<int64_t>(v, 10), 28));
  else if (k == "small_blocks_per_cu") b->small_blocks_per_cu = static_cast<int>(std::max<int64_t>(v, 0));
  else if (k == "blocks_cap") b->blocks_cap = static_cast<int>(std::max<int64_t>(v, 0));
  else if (k == "mono_copies") {
    if (b->in_session) return fail(b, HALO_FATAL, "mono_copies cannot change while a session's plane is pending");
    if (int rc = fold_if_dirty(b)) return rc;   // planes of ended sessions may still wait for their fold (lazy_fold)
    int c = 1;
    while (c < v && c < 64) c <<= 1;
    if (c != b->mono_copies) {
      if (int rc = sync_all(b)) return rc;       // (nothing may still be reading what is freed)
      b->mono.release();
    }
    b->mono_copies = c;
  }
  else if (k == "blocks_per_cu") b->blocks_per_cu = static_cast<int>(std::min<int64_t>(std::max<int64_t>(v, 1), 64));
  else if ((k == "rank" || k == "ray_base") && b->in_session) {
    // the layers of an open session draw their gate / transit / shape streams at indices that follow on from the first layer's
    return fail(b, HALO_FATAL, k + " cannot change inside a session");
  }
  else if (k == "rank") {
    // disjoint 64-bit counter ranges per shard: the hi word feeds pcg_seed_with_high, so ranks never share a stream
    const uint64_t base = static_cast<uint64_t>(v) << 40;
    b->gen_count = b->gate_count = b->transit_count = b->shape_count = base;
  } else if (k == "seed") {
    // re-seed between sessions (the reference's backends take their seed from the first non-zero SessionSpec::seed, cpu_trace_backend.cpp:252)
    if (b->in_session) return fail(b, HALO_FATAL, "seed cannot change inside a session");
    b->seed = static_cast<uint32_t>(v) ? static_cast<uint32_t>(v) : 1u;
  } else if (k == "ray_base") {
    // the session's first 64-bit ray index (SplitPcgRayBase, trace_backend.hpp:184: lo feeds the stream index, hi pcg_seed_with_high)
    const uint64_t base = static_cast<uint64_t>(v);
    b->gen_count = b->gate_count = b->transit_count = b->shape_count = base;
  } else return fail(b, HALO_FATAL, "unknown option: " + k);
  return HALO_OK;
}

int halo_set_stream(halo_handle_t b, void* s) {
  if (!b) return HALO_FATAL;
  (void)sync_all(b);
  b->stream = s ? static_cast<hipStream_t>(s) : b->own_stream;
  return HALO_OK;
}

int halo_bind_accumulator(halo_handle_t b, void* device_ptr, uint64_t n_floats) {
  if (!b) return HALO_FATAL;
  if (b->in_session) return fail(b, HALO_FATAL, "bind_accumulator inside a session");
  if (int rc = fold_if_dirty(b)) return rc;   // planes of ended sessions belong to the accumulator they were traced for
  // (The landed tally is NOT touched: it belongs to the sessions traced, and goes to whoever takes it — halo_take_landed, a readback — whichever
  //  tensor holds their light.  A caller that alternates two bound tensors, dist.ShardedTracer's drain, rebinds with nothing waiting on the host.)
  if (!device_ptr) {
    b->acc = b->acc_own.ptr;
    b->acc_floats = b->acc_own.cap;
    return HALO_OK;
  }
  b->acc = static_cast<float*>(device_ptr);
  b->acc_floats = n_floats;
  return HALO_OK;
}

int halo_set_filters(halo_handle_t b, const HaloFilter* filters, int32_t count) {
  if (!b || count < 0 || (count > 0 && !filters)) return HALO_FATAL;
  if (b->in_session) return fail(b, HALO_FATAL, "set_filters inside a session");
  for (int32_t i = 0; i < count; i++) {
    const HaloFilter& f = filters[i];
    int terms = f.is_complex ? 0 : 1;
    if (f.is_complex) {
      if (f.or_count < 0 || f.or_count > HALO_FILTER_MAX_OR) return fail(b, HALO_FATAL, "complex filter: too many OR-clauses");
      for (int o = 0; o < f.or_count; o++) terms += f.and_counts[o];
      if (terms > HALO_FILTER_MAX_TERMS) return fail(b, HALO_FATAL, "complex filter: too many terms");
    }
  }
  // (the Lumice glue hands the filter table over at every BeginSession: equal tables keep the table cache — ADVICE r5)
  const bool same = static_cast<size_t>(count) == b->filters.size() && (count == 0 || std::memcmp(b->filters.data(), filters, sizeof(HaloFilter) * static_cast<size_t>(count)) == 0);
  if (same) return HALO_OK;
  b->filters.assign(filters, filters + count);
  drop_table_cache(b);
  return HALO_OK;
}


int halo_begin(halo_handle_t b, const HaloScene* scene, const HaloRender* render, const HaloWl* wl, uint64_t ray_num) {
  if (!b || !scene || !render || !wl) return HALO_FATAL;
  if (b->in_session) return fail(b, HALO_FATAL, "BeginSession inside a session");
  if (scene->layer_count < 1 || scene->layer_count > HALO_MAX_LAYERS) return fail(b, HALO_FATAL, "layer_count out of range");
  if (scene->max_hits < 1 || scene->max_hits > HALO_MAX_HITS) return fail(b, HALO_FATAL, "max_hits out of range");
  if (render->width <= 0 || render->height <= 0) return fail(b, HALO_FATAL, "bad resolution");
  bool plain_scene = !b->capture;   // no filter, raypath colour or capture anywhere: every launch runs the production kernels
  for (int l = 0; l < scene->layer_count; l++) {
    if (scene->layers[l].entry_count < 1 || scene->layers[l].entry_count > HALO_MAX_ENTRIES)
      return fail(b, HALO_FATAL, "entry_count out of range");
    for (int e = 0; e < scene->layers[l].entry_count; e++) {
      const int fid = scene->layers[l].entries[e].filter_id;
      if (fid < 0 || fid > static_cast<int>(b->filters.size())) return fail(b, HALO_FATAL, "entry refers to a filter_id outside the table");
      const int cid = scene->layers[l].entries[e].color_id;
      if (cid < 0 || cid > static_cast<int>(b->color_sets.size())) return fail(b, HALO_FATAL, "entry refers to a color_id outside the table");
      if (fid > 0 || cid > 0) plain_scene = false;
      const int kind = scene->layers[l].entries[e].crystal.kind;
      if (kind != HALO_CRYSTAL_PRISM && kind != HALO_CRYSTAL_PYRAMID) return fail(b, HALO_FATAL, "unknown crystal kind");
    }
  }
  HIPCHK(b, hipSetDevice(b->device));
  if (std::memcmp(&b->tcache_scene, scene, sizeof(HaloScene)) != 0 || std::memcmp(&b->tcache_wl, wl, sizeof(HaloWl)) != 0) {
    drop_table_cache(b);   // another scene or wavelength: the cached tables are not this session's
    b->tcache_scene = *scene;
    b->tcache_wl = *wl;
  }
  b->scene = *scene;
  b->render = *render;
  b->wl = *wl;
  b->proj = host::BuildProj(*render);
  std::vector<WlEntryDev> pool = host::BuildWlPool(*wl);
  b->wl_pool_size = static_cast<uint32_t>(pool.size());
  if (pool.empty() || pool.size() > HALO_WL_POOL_MAX) return fail(b, HALO_FATAL, "wavelength pool size out of range");
  b->wl_pool_host = pool;  // travels to the device inside each dispatch slot
  b->sess_max_w = 0.0;     // no ray of this session weighs more (continuations only lose weight): bounds the per-tile fixed-point sums
  for (const WlEntryDev& e : pool) b->sess_max_w = std::max(b->sess_max_w, static_cast<double>(std::fabs(e.spd_weight)));
  // Accumulation planes of this session (kernels never touch the XYZ image; halo_fold_kernel closes the session):
  //  * discrete wavelength: ONE scalar plane, coefficient = CMF(lambda)            (1 atomic per hit)
  //  * illuminant, batch >= 8 Mi rays: one scalar plane per pool entry, coefficient = its CMF (1 atomic per hit, fold reads M planes)
  //  * illuminant, small batch or "mono" = 0: X, Y, Z planes, unit coefficients        (3 atomics per hit, cheap fold)
  const size_t npix = static_cast<size_t>(render->width) * render->height;
  // 2^25 pixels (an 8192x4096 panorama) is what the workgroup cache's 32-bit key names on its own; with a plane index beside the pixel the
  // key keeps 23 bits for the pixel, so sessions above 2^23 pixels take the layouts without per-entry planes (below: mono_by_wl)
  if (npix > (1u << 25)) return fail(b, HALO_UNAVAILABLE, "more than 2^25 pixels");   // recoverable: the caller falls back (INTEGRATION.md limits table)
  const bool discrete = wl->illuminant < 0;
  //  * illuminant, batch >= 2 Mi rays, hit log on (the default): X, Y, Z planes too, but the launches log {slot, pool entry, w} and
  //    the per-tile pass applies the CMF (halo_log_accumulate_kernel<3>): no plane per entry, no two-level split, a 3-plane fold
  //    (images above 512 Ki pixels: on small ones most hits land in the pixel cache, and the X/Y/Z cache pays three fp32 LDS adds — the
  //    slow kind on gfx950 — per hit where a scalar plane pays one: the reference's 512x256 D65 scene runs 2.8 vs 8.0 G rays/s)
  uint32_t s_log2 = 6;  // columns per plane row: at least one fold tile
  while ((static_cast<size_t>(kMonoRows) << s_log2) < npix) s_log2++;
  // the log's tile layout has limits (halo_trace_layer: log_layout_ok): X/Y/Z records keep the slot in 23 bits of which the per-tile pass
  // takes <= 4 Ki-slot tiles x 512, scalar planes 16 Ki-slot tiles x 256.  A session the log cannot serve keeps the routes it had before
  // (one plane per pool entry + binned lists; privatised copies for the direct atomics).
  const bool log_xyz_fits = s_log2 <= 11u, log_mono_fits = s_log2 <= 12u;
  // (the layout is worked out in locals first: whether the planes of earlier sessions must be folded before this one starts depends on it)
  const bool xyz_log_n = b->mono_enabled && !discrete && b->lambda_planes < 0 && b->hit_log != 0 && ray_num >= (2ull << 20) && npix > HALO_XYZ_LOG_MIN_PIX && log_xyz_fits;
  const bool mono_by_wl_n = b->mono_enabled && !discrete && !xyz_log_n && npix <= (1u << 23) && (b->lambda_planes < 0 ? ray_num >= (8ull << 20) : b->lambda_planes != 0);
  const bool mono_session_n = b->mono_enabled && (discrete || mono_by_wl_n);
  const uint32_t plane_cnt_n = mono_by_wl_n ? static_cast<uint32_t>(pool.size()) : (mono_session_n ? 1u : 3u);
  // privatised copies spread the direct atomics of hot pixels; per-entry planes spread them already, and a session whose
  // launches all go through the hit log (copy 0 only) would just make the closing fold read seven empty copies
  // ("all" cannot be known here: a layer's rays are dealt out to its crystal entries, and an entry's launch under 2 Mi rays adds directly.
  // So the copies go only when every layer has ONE entry — a layer is then launches of the whole batch, and what may still fall under the
  // threshold, a last chunk or a thin continuation layer, is little work by definition)
  bool one_entry_layers = true;
  for (int l = 0; l < scene->layer_count; l++) one_entry_layers = one_entry_layers && scene->layers[l].entry_count == 1;
  const bool fast_scene = b->rehit_strategy != 0 && (plain_scene || (!b->capture && b->filter_fast && scene->max_hits <= 16));   // (a dispatch whose tables do not fit the fast form still adds directly: copy 0 only, correct, slower)
  const bool all_logged = fast_scene && one_entry_layers && b->hit_log < 0 && b->aggregate == 1 && ray_num >= (discrete ? log_min_rays(render->visible) : (2ull << 20)) &&
                          (discrete ? log_mono_fits : xyz_log_n) && (mono_session_n || xyz_log_n);
  const uint32_t plane_copies_n = (mono_by_wl_n || all_logged) ? 1u : static_cast<uint32_t>(b->mono_copies);
  const bool two_sets_n = all_logged && discrete && mono_session_n && !mono_by_wl_n && b->overlap != 0;   // (see HaloBackend::mono_two)
  std::vector<std::array<float, 3>> coef_n;
  for (uint32_t m = 0; m < plane_cnt_n; m++) {
    if (mono_session_n) coef_n.push_back({pool[m].cmf_x, pool[m].cmf_y, pool[m].cmf_z});
    else coef_n.push_back({m == 0 ? 1.0f : 0.0f, m == 1 ? 1.0f : 0.0f, m == 2 ? 1.0f : 0.0f});
  }
  {
    // Planes that earlier sessions left unfolded (lazy_fold: halo_end of a session on the backend's own accumulator does not fold) stay as
    // they are when this session adds to the SAME planes with the SAME coefficients — a server that sends one wavelength's rays as many small
    // sessions (Lumice's CUDA-route dispatch is 2^18 rays, server.cpp:151) then pays one fold per readback instead of one per session (22 us
    // of a 0.15 ms session at 1920x1080 x 8 copies).  Anything else — another wavelength, image size, plane count or copy count — is folded
    // now, with the members still describing the old planes.
    const bool same_planes = b->mono_dirty && b->lazy_fold && b->acc != nullptr && b->acc == b->acc_own.ptr && b->acc_w == render->width &&
                             b->acc_h == render->height && b->mono_s_log2 == s_log2 && b->plane_cnt == plane_cnt_n && b->plane_copies == plane_copies_n &&
                             b->mono_two == two_sets_n && b->plane_coef == coef_n;
    if (!same_planes) {
      // (queued on the auxiliary stream; the HOST does not wait.  This session's trace kernels are queued behind it all the same
      // (next_trace_stream is always told the launch touches the planes): letting a logged kernel start under the fold was measured — the step
      // did not shrink, DESIGN.md 3.6 — so the simpler order stays, and the twin's two halves only keep a LATE fold and the open session apart)
      int rc = fold_queue(b);  // also: a session that was never ended still owes its plane to the accumulator (old layout)
      if (rc != HALO_OK) return rc;
      const bool caller_reads = b->acc != nullptr && b->acc != b->acc_own.ptr && !b->defer_fold;   // the caller's memory, read in stream order: the fold must be in that order too
      const bool layout_changes = b->mono_s_log2 != s_log2 || b->plane_cnt != plane_cnt_n || b->plane_copies != plane_copies_n || b->mono_two != two_sets_n;   // the twin's halves (and the plane sets) move
      if (caller_reads || layout_changes) {
        rc = join_aux(b);
        if (rc != HALO_OK) return rc;
      }
    }
    int rc = ensure_accumulator(b, render->width, render->height);
    if (rc != HALO_OK) return rc;
  }
  b->xyz_log = xyz_log_n;
  b->mono_by_wl = mono_by_wl_n;
  b->mono_session = mono_session_n;
  b->plane_cnt = plane_cnt_n;
  b->plane_copies = plane_copies_n;
  b->plane_coef = coef_n;
  {
    const size_t one = (static_cast<size_t>(kMonoRows) << s_log2) * b->plane_copies * b->plane_cnt;
    b->mono_two = two_sets_n;
    b->mono_half = one;
    if (!b->mono_two) b->mono_set = 0;
    const size_t need = one * (b->mono_two ? 2u : 1u);
    if (b->mono.cap < need) {
      if (int rc = reserve_idle(b, b->mono, need)) return rc;
      HIPCHK(b, hipMemsetAsync(b->mono.ptr, 0, b->mono.cap * sizeof(float), b->stream));
    }
    b->mono_s_log2 = s_log2;  // planes are all-zero whenever the layout changes (folded above), so it may change freely
    // the twin: one double per float of every plane's copy 0 (2.1 MB x planes at configs[1]'s image; a 64-plane session on 2048x1024 takes
    // 1 GB), kept all-zero between sessions like the planes; sessions whose twin would pass 4 GB go without (their overflow stays on the fp32
    // plane), and so does a session whose twin cannot be allocated
    // (two halves: a session's overflows go to one while the fold of the session before may still be reading the other, `twin_set`)
    const size_t twin_need = (static_cast<size_t>(kMonoRows) << s_log2) * b->plane_cnt;
    if (b->ovf.cap < 2u * twin_need && twin_need <= (512ull << 20)) {
      hipError_t oe = hipSuccess;
      if (int rc = reserve_idle(b, b->ovf, 2u * twin_need, &oe)) return rc;
      bool ok = oe == hipSuccess;
      if (ok && !b->ovf_flag.ptr) ok = b->ovf_flag.reserve(2) == hipSuccess && hipMemsetAsync(b->ovf_flag.ptr, 0, 2 * sizeof(uint32_t), b->stream) == hipSuccess;
      if (ok) {
        HIPCHK(b, hipMemsetAsync(b->ovf.ptr, 0, b->ovf.cap * sizeof(double), b->stream));
      } else {
        (void)hipGetLastError();   // out of memory is not this session's failure: the launches run without a twin
        b->ovf.release();
        b->ovf_flag.release();
      }
    }
  }
  if (!b->color_classes.empty()) {  // Y lanes persist like the accumulator, until halo_readback_class_lanes
    const size_t need = b->color_classes.size() * npix;
    if (b->lanes.cap < need || b->lanes_w != render->width || b->lanes_h != render->height) {
      if (int rc = reserve_idle(b, b->lanes, need)) return rc;
      if (int rc = join_aux(b)) return rc;
      HIPCHK(b, hipMemsetAsync(b->lanes.ptr, 0, b->lanes.cap * sizeof(double), b->stream));
      b->lanes_w = render->width;
      b->lanes_h = render->height;
    }
  }
  b->sess_crystal_samples = b->sess_orient_samples = 0;
  b->route = HaloRouteInfo{};
  b->route.plane_cnt = b->plane_cnt;
  b->route.plane_copies = b->plane_copies;
  b->route.shuffle_chunk = 1u << b->shuffle_chunk_log2;
  b->in_session = true;
  b->layer_idx = 0;
  b->cont_in_n = 0;
  b->cont_out_slot = 0;
  return HALO_OK;
}

// Fractional bits of the per-tile fixed-point sums for a launch of `m` rays whose weights are at most `max_w`: the largest F <= 32 with
// 4 * max_w * m * 2^F < 2^62 (see halo_kernels.hip FixQ for the factor 4).
static uint32_t fix_frac_bits(double max_w, uint64_t m) {
  const double bound = 4.0 * std::max(max_w, 1e-30) * static_cast<double>(std::max<uint64_t>(m, 1));
  int e = 0;
  (void)std::frexp(bound, &e);   // bound < 2^e
  return static_cast<uint32_t>(std::min(32, std::max(0, 62 - e)));
}

// The twin half a session of the current layout uses: valid only when the buffer holds two halves of this layout.
static double* twin_of(HaloBackend* b, int set) {
  const size_t half = (static_cast<size_t>(kMonoRows) << b->mono_s_log2) * b->plane_cnt;
  if (!b->ovf.ptr || !b->ovf_flag.ptr || b->ovf.cap < 2u * half) return nullptr;
  return b->ovf.ptr + static_cast<size_t>(set) * half;
}

// Queue the closing fold of the pending planes on the post stream (behind everything both streams hold); the main stream does NOT wait.
static int fold_queue(HaloBackend* b) {
  if (!b->mono_dirty) return HALO_OK;
  HIPCHK(b, hipSetDevice(b->device));
  if (int rc = fork_aux(b)) return rc;
  hipStream_t ps = post_stream(b);
  const uint32_t npix = static_cast<uint32_t>(b->acc_w) * static_cast<uint32_t>(b->acc_h);
  const size_t plane = (static_cast<size_t>(kMonoRows) << b->mono_s_log2) * b->plane_copies;
  const size_t twin_plane = static_cast<size_t>(kMonoRows) << b->mono_s_log2;
  double* twin = twin_of(b, b->twin_set);
  uint32_t* flag = twin ? b->ovf_flag.ptr + b->twin_set : nullptr;
  for (uint32_t first = 0; first < b->plane_cnt; first += kFoldGroup) {
    const uint32_t n = std::min<uint32_t>(kFoldGroup, b->plane_cnt - first);
    FoldCoef coef{};
    for (uint32_t m = 0; m < n; m++)
      for (int a = 0; a < 3; a++) coef.c[m][a] = b->plane_coef[first + m][static_cast<size_t>(a)];
    hipError_t e = launch_fold(b->acc, planes_of(b) + first * plane, npix, b->mono_s_log2, b->plane_copies, n, coef, twin ? twin + first * twin_plane : nullptr, flag, ps);
    if (e != hipSuccess) return hip_fail(b, e, "halo_fold_kernel launch");
  }
  if (flag) HIPCHK(b, hipMemsetAsync(flag, 0, sizeof(uint32_t), ps));   // every group has seen it; the fold zeroed what it took
  b->twin_set ^= 1;   // what traces from here on overflows into the other half
  if (b->mono_two) {    // ... and adds to the other plane set, once the fold that last zeroed THAT one is through (ev_set_folded)
    if (!b->ev_set_folded[b->mono_set]) HIPCHK(b, hipEventCreateWithFlags(&b->ev_set_folded[b->mono_set], hipEventDisableTiming));
    HIPCHK(b, hipEventRecord(b->ev_set_folded[b->mono_set], ps));
    b->set_folded_pending[b->mono_set] = true;
    b->mono_set ^= 1;
  }
  b->mono_dirty = false;
  return HALO_OK;
}
// ... and with the main stream waiting for it: for readers of the image and for whoever hands the image to somebody else.
static int fold_if_dirty(HaloBackend* b) {
  if (int rc = fold_queue(b)) return rc;
  return join_aux(b);
}

int halo_end(halo_handle_t b) {
  if (!b) return HALO_FATAL;
  // a session closes by folding its planes into the XYZ image (a discrete wavelength's CMF applied to the scalar plane) — at once when the image
  // is the caller's memory (halo_bind_accumulator: the caller reads it without asking), otherwise when somebody reads it (every reader folds
  // first) or a session with other planes begins
  int rc = HALO_OK;
  const bool own = b->acc != nullptr && b->acc == b->acc_own.ptr;
  if (!((b->lazy_fold && own) || (b->defer_fold && !own))) rc = fold_if_dirty(b);
  b->in_session = false;
  return rc;
}

int halo_trace_layer(halo_handle_t b, uint64_t count, const HaloHostRays* rays, HaloLayerStats* stats) {
  if (!b) return HALO_FATAL;
  if (!b->in_session) return fail(b, HALO_FATAL, "TraceLayer outside a session");
  const int layer = b->layer_idx;
  if (layer >= b->scene.layer_count) return fail(b, HALO_FATAL, "TraceLayer past the last layer");
  HIPCHK(b, hipSetDevice(b->device));
  const HaloLayer& L = b->scene.layers[layer];
  const uint64_t n = (layer == 0) ? count : b->cont_in_n;
  const bool final_layer = (layer == b->scene.layer_count - 1);
  if (n > 0xFFFFFFF0ull) return fail(b, HALO_FATAL, "more than 2^32 rays in one layer dispatch; split the batch");
  if (layer > 0 && rays) return fail(b, HALO_FATAL, "host rays are first-layer only");
  if (rays && rays->crystal) {   // an injected crystal is copied into fixed-size device tables: its counts and indices are the caller's, check them
    const HaloGeomTables& g = *rays->crystal;
    if (g.face_cnt < 0 || g.face_cnt > HALO_MAX_FACES || g.tri_cnt < 0 || g.tri_cnt > HALO_MAX_TRIS)   // (0 faces = the empty crystal: its rays carry no weight)
      return fail(b, HALO_FATAL, "host crystal: face_cnt / tri_cnt outside 0..HALO_MAX_FACES / 0..HALO_MAX_TRIS");
    for (int t = 0; t < g.tri_cnt; t++)
      if (g.tri_face[t] < 0 || g.tri_face[t] >= g.face_cnt) return fail(b, HALO_FATAL, "host crystal: tri_face refers to a face outside the table");
  }

  float props[HALO_MAX_ENTRIES];
  for (int ci = 0; ci < L.entry_count; ci++) props[ci] = L.entries[ci].proportion;
  std::vector<uint64_t> per_ci = host::Partition(props, L.entry_count, n, b->carry[layer]);
  if (rays && layer == 0) {  // host ingest is a single population (cpu_trace_backend.cpp:121-127)
    std::fill(per_ci.begin(), per_ci.end(), 0);
    per_ci[0] = n;
  }
  // launch plan: chunking bounds the host-built shape pool for stochastic geometry and keeps n_rays < 2^32
  const int max_blocks = b->cu_count * b->blocks_per_cu;
  auto chunk_of = [&](uint64_t left, const HaloCrystal& crystal) {
    uint64_t m = std::min<uint64_t>(left, b->chunk);
    // stochastic geometry: one pool record per geom_clock rays.  Device-generated prisms are 1360 B records (2.9 GB per
    // 64 Mi rays), other device-generated shapes 4.1 KB (2 GB per 16 Mi rays); host-built pools (pageable staging + H2D)
    // stay at 4 Mi rays
    if (!host::IsDeterministic(crystal) && !(rays != nullptr && rays->crystal != nullptr && layer == 0)) {
      const uint64_t dev = b->stoch_chunk ? b->stoch_chunk : (1ull << 26);   // records: 2.9 GB (prism) / 8.6 GB (general) per 64 Mi rays
      m = std::min<uint64_t>(m, b->host_shapes ? (1ull << 22) : dev);
    }
    return m;
  };
  auto blocks_of = [&](uint64_t m, bool pool) {
    // per-workgroup fixed costs (table staging, pixel-cache zero + flush) are paid per launch, a long tail is paid per
    // round of resident workgroups: aim at >= 32 passes of the ray loop per workgroup, between 4 and blocks_per_cu per CU
    // (measured: 1 M rays 0.27 -> 0.20 ms at 4/CU; 50 M rays 4.02 -> 3.67 ms at 24/CU instead of 8/CU)
    // Small launches — a Lumice server that keeps a GPU route's small dispatch (2^15 .. 2^18 rays per session, server.cpp:140-151) sends nothing
    // else — are latency: one pass of the ray loop is ~11 us on a wave that has its SIMD to itself, so they spread over as many workgroups as
    // a CU holds (five) before a workgroup takes a second pass.  (Round 4 had found the opposite — one per CU best — while every WAVE ended
    // with four same-line fp64 atomics; with those gone, tools/dispatch_probe.cpp, us per session at 1 / 2 / 3 / 4 / 5 / 8 per CU:
    // 2^18 rays 67 / 50 / 45 / 44 / 43 / 44, 2^20 188 / 133 / 122 / 115 / 110 / 111, 2^22 326 / 326 / 276 / 261 / 250 / 259.)
    const uint64_t want = m / (static_cast<uint64_t>(kBlock) * 32u);
    // (shape-pool kernels hold four workgroups a CU, not six, and every workgroup of a logged launch is a region for the split pass: three per CU
    //  there — bench.py --config 4d, 0.8 M rays per session, ms per step at 1 / 2 / 3 / 4 / 5 / 6 / 8 per CU: 7.26 / 5.99 / 5.73 / 5.77 / 5.86 / 6.14 / 6.39)
    const int small_k = b->small_blocks_per_cu > 0 ? b->small_blocks_per_cu : pool ? 3 : 5;
    const uint64_t lo_cap = static_cast<uint64_t>(b->cu_count) * static_cast<uint64_t>(std::min(b->blocks_per_cu, small_k));
    const uint64_t cap = std::min<uint64_t>(static_cast<uint64_t>(max_blocks), std::max<uint64_t>(lo_cap, want));
    uint64_t nb = std::min<uint64_t>((m + kBlock - 1) / kBlock, cap);
    if (b->blocks_cap > 0) nb = std::min<uint64_t>(nb, static_cast<uint64_t>(b->blocks_cap));
    return static_cast<int>(nb);
  };

  // continuation output pool: kContShards regions; a region must hold everything its blocks can emit over the layer's
  // dispatches (every root emits at most max_hits candidates)
  const int out_slot = b->cont_out_slot;
  uint32_t out_cap = 0;
  if (!final_layer) {
    uint64_t region = 0;
    for (int ci = 0; ci < L.entry_count; ci++) {
      const bool det = (rays != nullptr && rays->crystal != nullptr && layer == 0) || host::IsDeterministic(L.entries[ci].crystal);   // (as the launch below decides it: the same workgroup count)
      for (uint64_t off = 0; off < per_ci[ci];) {
        const uint64_t m = chunk_of(per_ci[ci] - off, L.entries[ci].crystal);
        const uint64_t nb = static_cast<uint64_t>(blocks_of(m, !det));
        const uint64_t per_block = (m + nb * kBlock - 1) / (nb * kBlock) * kBlock;   // grid-stride share, rounded up
        region += (nb + kContShards - 1) / kContShards * per_block * static_cast<uint64_t>(b->scene.max_hits);
        off += m;
      }
    }
    region = (region + 63) & ~63ull;
    const uint64_t stride = region * kContShards;
    if (stride > 0xFFFFFFF0ull) return fail(b, HALO_FATAL, "continuation pool would exceed 2^32 rays; split the batch");
    if (int rc = reserve_idle(b, b->cont[out_slot], stride * (b->color_classes.empty() ? 5 : 7))) return rc;  // + mask lo/hi planes with raypath colour
    b->cont_stride[out_slot] = static_cast<uint32_t>(stride);
    b->cont_region[out_slot] = static_cast<uint32_t>(region);
    out_cap = static_cast<uint32_t>(region);
    HIPCHK(b, b->cont_cnt.reserve(kContShards * kContCntStride));
    HIPCHK(b, hipMemsetAsync(b->cont_cnt.ptr, 0, kContShards * kContCntStride * sizeof(uint32_t), b->stream));
  }
  const bool defer = b->async && final_layer && !b->capture;  // nothing the caller needs before the next call
  if (!defer) {  // earlier queued dispatches go to `pending`, so layer_acc ends up holding this layer alone
    harvest_all(b);
    if (int rc = pull_tally(b)) return rc;   // (a copy + sync only when something was queued and never collected)
    add_stats(b->pending, b->layer_acc);
    b->layer_acc = HaloLayerStats{};
  }
  if (b->capture) {
    const uint64_t need = b->exits_pending + n * static_cast<uint64_t>(b->scene.max_hits + 1);
    if (need > b->exits.cap) {
      DevBuf<HaloExitRecord> bigger;
      HIPCHK(b, bigger.reserve(need));
      if (b->exits_pending)
        HIPCHK(b, hipMemcpyAsync(bigger.ptr, b->exits.ptr, b->exits_pending * sizeof(HaloExitRecord), hipMemcpyDeviceToDevice, b->stream));
      if (int rc = sync_all(b)) return rc;
      b->exits.release();
      b->exits = bigger;
    }
  }

  if (rays && layer == 0)   // host rays bring their own weights (and hand them on to the layers behind)
    for (uint64_t i = 0; i < n; i++) b->sess_max_w = std::max(b->sess_max_w, static_cast<double>(std::fabs(rays->w[i])));
  if (rays && layer == 0) {
    // a queued session's kernel on a trace stream (async + overlap) may still be reading the previous batch out of these buffers: the copies
    // go behind everything the trace / auxiliary streams hold, and a buffer only grows with all streams idle
    if (int rc = join_aux(b)) return rc;
    if (int rc = reserve_idle(b, b->host_f, n * 7)) return rc;
    if (int rc = reserve_idle(b, b->host_u, n)) return rc;
    HIPCHK(b, hipMemcpyAsync(b->host_f.ptr, rays->d, n * 3 * sizeof(float), hipMemcpyHostToDevice, b->stream));
    HIPCHK(b, hipMemcpyAsync(b->host_f.ptr + n * 3, rays->p, n * 3 * sizeof(float), hipMemcpyHostToDevice, b->stream));
    HIPCHK(b, hipMemcpyAsync(b->host_f.ptr + n * 6, rays->w, n * sizeof(float), hipMemcpyHostToDevice, b->stream));
    HIPCHK(b, hipMemcpyAsync(b->host_u.ptr, rays->tf, n * sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
    HIPCHK(b, hipStreamSynchronize(b->stream));
  }

  uint64_t ci_start = 0;
  for (int ci = 0; ci < L.entry_count; ci++) {
    const uint64_t n_ci = per_ci[ci];
    if (n_ci == 0) continue;
    const HaloEntry& E = L.entries[ci];
    DispatchParams P{};
    P.source = rays ? kSrcHost : (layer == 0 ? kSrcGen : kSrcTransit);
    P.layer = static_cast<uint32_t>(layer);
    P.final_layer = final_layer ? 1u : 0u;
    P.max_hits = static_cast<uint32_t>(b->scene.max_hits);
    P.prob = L.prob;
    P.crystal_id = static_cast<uint32_t>(E.crystal_config_id);
    P.capture = static_cast<uint32_t>(b->capture);
    P.gen_seed = b->seed ^ kNonceGen;
    P.gate_seed = b->seed ^ kNonceGate;
    P.transit_seed = b->seed ^ kNonceTransit;
    P.shuffle = static_cast<uint32_t>(b->cont_shuffle);
    P.shuffle_seed = (b->seed ^ kNonceShuffle) ^ static_cast<uint32_t>(layer);  // cuda_trace_backend.cu:4541
    P.shuffle_chunk_log2 = b->shuffle_chunk_log2;
    // orientation wire params (BuildTransitGpParams cuda_trace_backend.cu:342-383)
    P.lat_path = host::SelectLatPath(E.axis);
    P.lat_mean_rad = E.axis.latitude.center * host::kDegToRad;
    P.lat_std_rad = E.axis.latitude.spread * host::kDegToRad;
    P.az_type = static_cast<uint32_t>(E.axis.azimuth.type);
    P.az_mean_rad = E.axis.azimuth.center * host::kDegToRad;
    P.az_std_rad = E.axis.azimuth.spread * host::kDegToRad;
    P.roll_type = static_cast<uint32_t>(E.axis.roll.type);
    P.roll_mean_rad = E.axis.roll.center * host::kDegToRad;
    P.roll_std_rad = E.axis.roll.spread * host::kDegToRad;
    {  // BuildGenGpParams cuda_trace_backend.cu:391-399, trig evaluated once on the host
      const float sun_lon = (b->scene.sun_azimuth + 180.0f) * host::kDegToRad;
      const float sun_lat = -b->scene.sun_altitude * host::kDegToRad;
      const float half = (b->scene.sun_diameter * 0.5f) * host::kDegToRad;
      P.c_cap = std::cos(half);
      P.c_lon = std::cos(sun_lon);
      P.s_lon = std::sin(sun_lon);
      P.c_lat = std::cos(sun_lat);
      P.s_lat = std::sin(sun_lat);
    }
    P.wl_pool_size = b->wl_pool_size;
    P.proj = b->proj;
    P.proj_pre[0] = static_cast<float>(b->proj.img_w) / 2.0f, P.proj_pre[1] = static_cast<float>(b->proj.img_h) / 2.0f;   // the pixel formulas' conversions, once per dispatch
    P.proj_pre[2] = static_cast<float>(b->proj.lens_shift_x), P.proj_pre[3] = static_cast<float>(b->proj.lens_shift_y);
    P.cont_in = b->cont[out_slot ^ 1].ptr;
    P.cont_in_n = static_cast<uint32_t>(b->cont_in_n);
    P.cont_in_stride = b->cont_stride[out_slot ^ 1];
    P.cont_out = b->cont[out_slot].ptr;
    P.cont_out_stride = b->cont_stride[out_slot];
    P.cont_out_cap = out_cap;
    P.cont_in_region = b->cont_region[out_slot ^ 1];
    P.cont_cnt = b->cont_cnt.ptr;
    P.counters = b->counters.ptr;
    P.mono = planes_of(b);
    P.mono_s_log2 = b->mono_s_log2;
    P.ovf = twin_of(b, b->twin_set);
    P.ovf_copies_log2 = static_cast<uint32_t>(__builtin_ctz(b->plane_copies));
    P.ovf_flag = P.ovf ? b->ovf_flag.ptr + b->twin_set : nullptr;
    P.mono_copy_mask = b->plane_copies - 1u;
    P.mono_by_wl = b->mono_by_wl ? 1u : 0u;
    P.bin_list = nullptr;
    P.bin_log = 0u;
    P.log_xyz = 0u;
    P.log_plane_stride = 0u;
    P.pool_entry_fast = b->pool_entry_fast ? 1u : 0u;
    P.rehit_legacy = b->rehit_strategy == 0 ? 1u : 0u;
    P.bin_shift = 0u;
    P.tally = b->tally.ptr;
    P.exits = b->exits.ptr;
    P.exit_cap = static_cast<uint32_t>(std::min<uint64_t>(b->exits.cap, 0xFFFFFFFFull));
    P.aggregate = static_cast<uint32_t>(b->aggregate);
    P.geom_clock = b->geom_clock;
    // HostRayBatch::crystal (trace_backend.hpp:230-239): injected rays may bring the crystal they were sampled on; it is traced as it is —
    // one shape for the whole batch, no draw from the shape stream, no stochastic sample counted (test_cpu_trace_backend.cpp:737-777)
    const bool host_crystal = rays != nullptr && rays->crystal != nullptr && layer == 0;
    const bool deterministic = host_crystal || host::IsDeterministic(E.crystal);
    // the entry's tables may still be on the device from an earlier dispatch (table cache, see HaloBackend::TableCacheEntry)
    HaloBackend::TableCacheEntry* ce = (b->table_cache && deterministic && !host_crystal && P.source != kSrcTransit) ? &b->tcache[layer][ci] : nullptr;
    if (ce && !b->tcache_dev.ptr) {
      if (b->tcache_dev.reserve(static_cast<size_t>(HALO_MAX_LAYERS) * HALO_MAX_ENTRIES * 2u) != hipSuccess) {
        (void)hipGetLastError();
        ce = nullptr;   // no room for the cache: every dispatch uploads its tables
      }
    }
    DispatchSlot* const ce_dev = ce ? b->tcache_dev.ptr + (static_cast<size_t>(layer) * HALO_MAX_ENTRIES + static_cast<size_t>(ci)) * 2u : nullptr;   // the entry's two slots
    bool cached = ce && ce->valid;
    FilterDev fd{};
    bool use_filter = false;
    if (cached) use_filter = ce->use_filter;
    else if (E.filter_id > 0) {
      fd = host::BuildFilter(b->filters[static_cast<size_t>(E.filter_id - 1)], E.axis);
      use_filter = fd.is_complex || fd.terms[0].type != HALO_FILTER_NONE || fd.action != 0;
    }

    // raypath colour: this entry's predicates (canonicalised like filter terms, each with its own symmetry) + the classes
    ColorDev cd{};
    const bool use_color = !b->color_classes.empty();
    if (use_color && !cached) {
      cd.class_cnt = static_cast<uint32_t>(b->color_classes.size());
      for (uint32_t c = 0; c < cd.class_cnt; c++) {
        cd.class_bits[c] = b->color_classes[c].bits;
        cd.class_all[c] = b->color_classes[c].combine_all ? 1 : 0;
      }
      if (E.color_id > 0) {
        const HaloColorSet& cs = b->color_sets[static_cast<size_t>(E.color_id - 1)];
        cd.term_cnt = static_cast<uint32_t>(std::min(cs.term_count, HALO_COLOR_MAX_TERMS));
        for (uint32_t k = 0; k < cd.term_cnt; k++) {
          HaloFilter one{};
          one.symmetry = cs.terms[k].symmetry;
          one.terms[0] = cs.terms[k].predicate;
          const FilterDev f1 = host::BuildFilter(one, E.axis);
          cd.terms[k].t = f1.terms[0];
          cd.terms[k].symmetry = f1.symmetry;
          cd.terms[k].d_applicable = f1.d_applicable;
          cd.terms[k].sigma_a = f1.sigma_a;
          cd.terms[k].bit = static_cast<uint8_t>(cs.terms[k].bit & 0xFF);
        }
      }
    }

    // Which kernels: capture (tests) > filter / colour in their fast form (paths <= 16 faces, tables fit) > the generic filter kernels > plain
    int mode = 0;
    FastTables* fast_host = nullptr;
    if (cached) mode = ce->mode;
    else if (b->capture) mode = 2;
    else if (b->rehit_strategy == 0) mode = 3;   // the legacy next-face strategy lives in the generic kernels only
    else if (use_filter || use_color) {
      mode = 3;
      if (b->filter_fast && b->scene.max_hits <= 16) {
        b->fast_scratch.reset(new FastTables());
        std::memset(b->fast_scratch.get(), 0, sizeof(FastTables));
        const HaloColorSet* cs = (use_color && E.color_id > 0) ? &b->color_sets[static_cast<size_t>(E.color_id - 1)] : nullptr;
        if (host::BuildFastTables(use_filter ? &b->filters[static_cast<size_t>(E.filter_id - 1)] : nullptr, cs, E.axis, static_cast<uint32_t>(E.crystal_config_id), *b->fast_scratch)) {
          fast_host = b->fast_scratch.get();
          fast_host->class_cnt = use_color ? cd.class_cnt : 0u;
          for (uint32_t c = 0; c < fast_host->class_cnt; c++) {
            fast_host->class_bits[c] = cd.class_bits[c];
            fast_host->class_all[c] = cd.class_all[c];
          }
          mode = use_color ? 4 : 1;
        }
      }
    }
    const bool fast_mode = mode == 0 || mode == 1 || mode == 4;   // production-shaped kernels: exit queue, hit log, regular-prism search

    // chunked launches: bounds the host-built shape pool for stochastic geometry and keeps n_rays < 2^32
    for (uint64_t off = 0; off < n_ci;) {
      const uint64_t m = chunk_of(n_ci - off, E.crystal);
      const uint32_t shape_cnt = deterministic ? 1u : static_cast<uint32_t>((m + b->geom_clock - 1) / b->geom_clock);
      const bool host_pool = !deterministic && b->host_shapes;
      // 0 = one shape per dispatch, 1 = pool of ShapeDev records, 2 = pool of ShapePrism records (device-generated prisms)
      const int geom = deterministic ? 0 : ((E.crystal.kind == HALO_CRYSTAL_PRISM && !host_pool) ? 2 : 1);
      std::vector<ShapeDev> pool((deterministic && !cached) || host_pool ? shape_cnt : 0u);
      for (uint32_t k = 0; k < static_cast<uint32_t>(pool.size()); k++) {
        if (host_crystal) host::ToShapeDev(*rays->crystal, pool[k]);
        else host::MakeShapeDev(b->seed, E.crystal, deterministic ? 0 : (b->shape_count + k), pool[k]);
      }
      const uint64_t first_shape = b->shape_count;
      if (!deterministic) {
        b->shape_count += shape_cnt;
        b->sess_crystal_samples += shape_cnt;
      }
      if (rays == nullptr && (E.axis.azimuth.type != HALO_DIST_NONE || E.axis.latitude.type != HALO_DIST_NONE || E.axis.roll.type != HALO_DIST_NONE))
        b->sess_orient_samples += m;   // AxisDistribution::IsAxisDeterministic is false: one orientation per ray
      // ---- dispatch slot: tables + zeroed tallies go up in one copy from the pinned mirror ----
      const int k = b->ring_next;
      b->ring_next = (k + 1) % HaloBackend::kRing;
      harvest_slot(b, k);  // blocks only if the ring has wrapped onto a dispatch still in flight
      b->ring_use[k]++;
      DispatchSlot& hs = b->ring_host[k];
      if (ce && !cached) ce->cur ^= 1;   // an upload takes the entry's other slot
      DispatchSlot* ds = ce ? ce_dev + ce->cur : b->ring_dev.ptr + k;
      bool entry_fast = false, hex_regular = false, has_fast = fast_host != nullptr;
      if (cached) {
        entry_fast = ce->entry_fast, hex_regular = ce->hex_regular, has_fast = ce->has_fast;
      } else {
        if (P.lat_path == kLatLut) {
          const host::LatLut& lut = cached_lut(b, E.axis.latitude);
          std::copy(lut.theta.begin(), lut.theta.end(), hs.lut);
          std::copy(lut.cdf.begin(), lut.cdf.end(), hs.lut + kLutNodes);
          std::copy(lut.flip.begin(), lut.flip.end(), hs.lut + 2 * kLutNodes);
        }
        std::copy(b->wl_pool_host.begin(), b->wl_pool_host.end(), hs.wl);
        if (deterministic) {
          hs.shape = pool[0];
          entry_fast = b->entry_fast && host::BuildEntryFast(pool[0], hs.efast);
          hex_regular = entry_fast && hs.efast.hex_regular;
        }
        if (use_filter) hs.filter = fd;
        if (use_color) hs.color = cd;
        if (fast_host) std::memcpy(&hs.fast, fast_host, sizeof(FastTables));
        if (P.source == kSrcTransit) std::copy(b->cont_seg, b->cont_seg + kContShards + 1, hs.seg);
        // (the fast filter tables are the slot's last member and travel only with the dispatches that use them; the pinned mirror is this
        // ring slot's, so it stays as it is until the copy has run — harvest_slot above)
        // A cache slot is a fixed device address: a kernel of an earlier, queued session (async: nobody harvested it) may still be staging its
        // tables out of it on a trace stream.  The upload goes behind the last launch that read this slot (no host wait; ring slots are
        // protected by harvest_slot above).  Without this the earlier kernel's late workgroups could pick up THIS session's wavelength pool or
        // shape (tests/test_gpu_production_routes.py::test_async_sessions_alternating_wavelengths).
        if (ce && ce->read_ring[ce->cur] >= 0) {
          const int rk = ce->read_ring[ce->cur];
          if (b->ring_busy[rk] && b->ring_use[rk] == ce->read_use[ce->cur]) HIPCHK(b, hipStreamWaitEvent(b->stream, b->ring_done[rk], 0));
        }
        HIPCHK(b, hipMemcpyAsync(ds, &hs, fast_host ? sizeof(DispatchSlot) : offsetof(DispatchSlot, fast), hipMemcpyHostToDevice, b->stream));
        if (ce) {   // the next dispatch of this entry — the layer's next chunk, the next equal session — finds the tables in place
          ce->valid = true, ce->mode = mode, ce->use_filter = use_filter, ce->use_color = use_color;
          ce->entry_fast = entry_fast, ce->hex_regular = hex_regular, ce->has_fast = has_fast;
          cached = true;
        }
      }
      P.lut = ds->lut;
      P.wl_pool = ds->wl;
      P.filter = use_filter ? &ds->filter : nullptr;
      P.color = use_color ? &ds->color : nullptr;
      P.fast = has_fast ? &ds->fast : nullptr;
      P.lanes = b->lanes.ptr;
      P.lane_stride = static_cast<uint32_t>(b->acc_w) * static_cast<uint32_t>(b->acc_h);
      P.cont_in_seg = ds->seg;
      P.entry_fast = entry_fast ? &ds->efast : nullptr;
      int shape_set = 0;
      if (deterministic) {
        P.shapes = &ds->shape;
      } else {
        // Two pools, taken in turn by the launches that may run beside a neighbour: the records of launch k+1 are written while launch k still
        // traces out of the other one.
        const bool turn = b->overlap && (m <= (1ull << b->alt_log2) || b->gen_ahead);   // (a launch that fills the chip runs alone on stream 0: pool 0)
        shape_set = turn ? b->shapes_next : 0;
        if (turn) b->shapes_next ^= 1;
        DevBuf<ShapeDev>& shapes = b->shapes_s[shape_set];
        if (int rc = reserve_idle(b, shapes, shape_cnt)) return rc;   // sized for ShapeDev records; prism pools use the front third
        if (turn)   // (its twin with it, for the same reason as the log's second buffer set)
          if (int rc = reserve_idle(b, b->shapes_s[shape_set ^ 1], shape_cnt)) return rc;
        if (host_pool) {
          if (b->shapes_used[shape_set]) HIPCHK(b, hipStreamWaitEvent(b->stream, b->ev_shapes_free[shape_set], 0));
          HIPCHK(b, hipMemcpyAsync(shapes.ptr, pool.data(), pool.size() * sizeof(ShapeDev), hipMemcpyHostToDevice, b->stream));
        }
        // (device generator: queued below on the launch's trace stream, in front of the trace kernel)
        P.shapes = shapes.ptr;
      }
      P.shape_cnt = shape_cnt;
      P.n_rays = static_cast<uint32_t>(m);
      P.ci_start = static_cast<uint32_t>(ci_start + off);
      auto split = [](uint64_t v, uint32_t& lo, uint32_t& hi) {  // SplitPcgRayBase trace_backend.hpp:184-190
        lo = static_cast<uint32_t>(v & 0xFFFFFFFFull);
        hi = static_cast<uint32_t>(v >> 32);
      };
      split(b->gen_count, P.gen_lo, P.gen_hi);
      split(b->gate_count, P.gate_lo, P.gate_hi);
      split(b->transit_count, P.transit_lo, P.transit_hi);
      if (rays) {
        P.host_d = b->host_f.ptr + off * 3;
        P.host_p = b->host_f.ptr + n * 3 + off * 3;
        P.host_w = b->host_f.ptr + n * 6 + off;
        P.host_tf = b->host_u.ptr + off;
      }
      const int blocks = blocks_of(m, !deterministic);
      // binned accumulation for big one-plane launches (see halo_trace.inl: HitBuffer)
      // 16384-slot tiles over the session's planes taken as one array (1 plane, or the per-entry planes of a small image)
      const uint64_t bin_slots = (static_cast<uint64_t>(kMonoRows) << b->mono_s_log2) * (b->mono_by_wl ? b->plane_cnt : 1u);
      const uint32_t bin_tiles = static_cast<uint32_t>(bin_slots >> 14);
      const bool bin_geom_ok = deterministic || E.crystal.kind == HALO_CRYSTAL_PRISM;
      // up to 512 tiles: one level (interleaved tiles).  More (per-wavelength planes of a large image): two levels — the
      // trace kernel fills coarse lists of `fan` consecutive tiles each, a split pass deals them out to the tiles.
      uint32_t fan_log2 = 0u;
      while (((bin_tiles + (1u << fan_log2) - 1u) >> fan_log2) > b->bin_l1) fan_log2++;
      const bool two_level = bin_tiles > 512u;
      const bool bin_shape_ok = two_level ? (fan_log2 <= 8u && (bin_slots & 16383ull) == 0ull)
                                          : (bin_tiles >= 8u && (bin_tiles & (bin_tiles - 1u)) == 0u);
      // The hit log over the scalar planes of a per-entry-plane session (small images: the reference's 512x256 D65 scenes): every plane gets
      // 2^t interleaved tiles of <= 16 Ki slots, and the split pass feeds at most 512 lists — 64 planes up to 128 Ki pixels, 32 up to 256 Ki.
      const uint32_t wl_t_log2 = b->mono_s_log2 >= 4u ? b->mono_s_log2 - 4u : 0u;
      const bool log_planes_ok = b->mono_by_wl && (static_cast<uint64_t>(b->plane_cnt) << wl_t_log2) <= 512ull &&
                                 (static_cast<uint64_t>(b->plane_cnt) << (b->mono_s_log2 + 10u)) <= (1ull << 31) && b->hit_log != 0 && b->bin <= 0;
      const bool use_bin = b->mono_session && b->aggregate == 1 && !b->capture && bin_shape_ok && bin_slots <= (1ull << 31) &&
                           // own choice: only where the hit log cannot go (one plane per pool entry on a larger image) — the log beats the binned route
                           // on every launch measured (tools/bin_vs_log_probe.py: dual fisheye 50 M rays 5.13 -> 4.42 ms)
                           (b->bin < 0 ? (b->mono_by_wl && !log_planes_ok && bin_geom_ok && b->render.visible == HALO_VISIBLE_FULL && m >= (2ull << 20)) : b->bin != 0);
      const uint32_t lists1 = two_level ? ((bin_tiles + (1u << fan_log2) - 1u) >> fan_log2) : bin_tiles;
      uint32_t cap2 = 0u;
      // Hit log (halo_trace.inl log_hit): global fp32 atomics retire memory-side at 21 G/s on this part, which bounded every big
      // launch once its trace was fast enough (configs[1]: 2.4 ms of trace, 2.8 ms of atomics).  A production-mode launch on one
      // scalar plane, or on the X/Y/Z planes of an illuminant session, appends its cache misses to one log region per workgroup
      // instead; a split pass and a per-tile pass add them to the plane(s) afterwards.
      // Tiles: as many as the split pass feeds whatever the image size — the per-tile pass has one workgroup per tile — of at least
      // 256 and at most 16 Ki slots (X/Y/Z: 4 Ki) of ONE plane: 512 for X/Y/Z; for a scalar plane 128 where that keeps them within
      // 16 Ki slots (longer runs in the split pass: 0.23 vs 0.25 ms at configs[1]).
      const uint32_t log_t_log2 = log_planes_ok ? wl_t_log2 : b->xyz_log ? std::min<uint32_t>(9u, b->mono_s_log2 + 2u) : std::max<uint32_t>(b->mono_s_log2 >= 4u ? b->mono_s_log2 - 4u : 0u, std::min<uint32_t>(static_cast<uint32_t>(m < (4ull << 20) ? std::max(b->log_tiles_log2, 8) : b->log_tiles_log2), b->mono_s_log2 + 2u));   // (a small launch takes 256 tiles: its per-tile pass is one workgroup per tile, and 128 of them leave half the chip idle — config 4d 2.85 -> 3.18 G rays/s; big launches keep 128 for the split's longer runs)
      const uint32_t log_planes = log_planes_ok ? b->plane_cnt : 1u;
      const uint32_t log_tiles = log_planes << log_t_log2;   // lists the split pass feeds
      const bool log_layout_ok = b->xyz_log ? (b->mono_s_log2 <= 11u) : (b->mono_session && (log_planes_ok || (!b->mono_by_wl && b->mono_s_log2 <= 12u)));
      bool use_log = !use_bin && log_layout_ok && b->aggregate == 1 && fast_mode &&
                           (P.prob < 1.0f || P.final_layer) &&   // a layer whose every exit continues puts nothing on the image
                           (b->hit_log < 0 ? m >= ((b->mono_session && !b->mono_by_wl) ? log_min_rays(b->render.visible) : (2ull << 20)) : b->hit_log != 0);
      bool use_log_xyz = use_log && b->xyz_log;
      uint64_t log_cap = 0;
      // Launches of <= 2^21 rays alternate between the two trace streams (next_trace_stream); only those can meet a neighbour that still reads
      // or writes a buffer set, so only those take the sets in turn — a launch that fills the chip keeps stream 0 and set 0 (a second set
      // allocated in the middle of a run is a host sync and a multi-gigabyte hipMalloc: 1.2 s in configs[4]'s first timed step)
      const bool alternate = m <= (1ull << b->alt_log2);
      const int ls = (b->overlap && alternate) ? b->log_set : 0;   // the buffer set of this launch
      if (use_log) {
        // a region takes 4 records per ray of its workgroup (configs[1]: 1.2 logged per ray), 8 for full-sky renders (5-6 per ray),
        // and a tile list twice its even share of that; what runs over falls back to direct atomics
        const uint64_t per_wg = (m + static_cast<uint64_t>(blocks) - 1u) / static_cast<uint64_t>(blocks);
        uint64_t cap = std::max<uint64_t>(4ull * per_wg, 4096ull);
        cap = std::min<uint64_t>(cap, (6ull << 30) / (8ull * static_cast<uint64_t>(blocks)));
        if (use_log_xyz || b->render.visible == HALO_VISIBLE_FULL) cap = std::min<uint64_t>(std::max<uint64_t>(8ull * per_wg, 4096ull), (8ull << 30) / (8ull * static_cast<uint64_t>(blocks)));
        // (per-launch cost of the slack, measured: configs[1] split 234 / 238 / 241 us at 2x / 4x / 8x; configs[4]'s 512 X/Y/Z lists 554 / 574 / 635 us —
        // so 8x for scalar planes, where a hot pixel meets near-constant addends, 4x for X/Y/Z, whose addends vary with the wavelength)
        const uint64_t kListSlack = use_log_xyz ? 4ull : 8ull;
        // (8x the even share, within the 8 GB below: what runs over a tile list is added with fp32 atomics, and on a pixel that holds 4e5 the
        // rounding of a near-constant addend is a bias, not noise — tools/route_fuzz.py seed 969: the sun's tile of a 1024 x 512 render took five
        // even shares and its pixel read 5.6e-4 high with 2x)
        uint64_t c2 = std::max<uint64_t>(kListSlack * ((use_log_xyz || b->render.visible == HALO_VISIBLE_FULL) ? 8ull : 4ull) * m / log_tiles, 1ull << 14);
        c2 = std::min<uint64_t>(c2, (8ull << 30) / (8ull * log_tiles));
        if (b->hit_log_cap) {   // tests: run both overflow fallbacks
          cap = b->hit_log_cap;
          c2 = std::max<uint64_t>(cap * static_cast<uint64_t>(blocks) / (2ull * log_tiles), 64ull);
        }
        // The log's capacity is a speed matter only (what runs over is added directly), so it gives way to the memory there is: the
        // regions and lists still to be allocated take at most half of what is free (the continuation and shape pools of later layers
        // want theirs), and a reserve that fails all the same sends this launch down the direct route instead of failing the trace.
        {
          size_t free_b = 0, total_b = 0;
          const uint64_t have = (b->bin_list_s[ls].cap + b->bin_list2_s[ls].cap) * sizeof(HitRec);
          uint64_t need = (cap * static_cast<uint64_t>(blocks) + c2 * log_tiles) * sizeof(HitRec);
          if (need > have && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need - have > free_b / 2u) {
            const double shrink = static_cast<double>(have + free_b / 2u) / static_cast<double>(need);
            cap = std::max<uint64_t>(static_cast<uint64_t>(static_cast<double>(cap) * shrink), 1024ull);
            c2 = std::max<uint64_t>(static_cast<uint64_t>(static_cast<double>(c2) * shrink), 1024ull);
          }
        }
        cap2 = static_cast<uint32_t>(c2 & ~15ull);   // whole 128-byte lines per tile list
        log_cap = cap;
        hipError_t e1 = hipSuccess, e2 = hipSuccess;
        if (int rc = reserve_idle(b, b->bin_list_s[ls], cap * static_cast<uint64_t>(blocks), &e1)) return rc;
        if (e1 == hipSuccess)
          if (int rc = reserve_idle(b, b->bin_list2_s[ls], c2 * log_tiles, &e2)) return rc;
        if (b->overlap && alternate && e1 == hipSuccess && e2 == hipSuccess) {
          // the other set with it: the next launch would otherwise stop the queue (a host sync and a multi-gigabyte hipMalloc) for it.  A speed
          // matter only — if there is no room, the launch that finds the set short shrinks its log as above.
          hipError_t e3 = hipSuccess, e4 = hipSuccess;
          if (int rc = reserve_idle(b, b->bin_list_s[ls ^ 1], cap * static_cast<uint64_t>(blocks), &e3)) return rc;
          if (e3 == hipSuccess)
            if (int rc = reserve_idle(b, b->bin_list2_s[ls ^ 1], c2 * log_tiles, &e4)) return rc;
          if (e3 != hipSuccess || e4 != hipSuccess) (void)hipGetLastError();
        }
        if (e1 != hipSuccess || e2 != hipSuccess) {
          (void)hipGetLastError();   // out of memory: not this launch's route
          use_log = use_log_xyz = false;
        }
      }
      DevBuf<HitRec>&bin_list = b->bin_list_s[ls], &bin_list2 = b->bin_list2_s[ls];
      DevBuf<uint32_t>&bin_cnt = b->bin_cnt_s[ls], &bin_cnt2 = b->bin_cnt2_s[ls];
      if (use_bin)   // the staged-list route shares one buffer set and resets its counters on the main stream: one launch at a time
        if (int rc = join_aux(b)) return rc;
      if (use_log) {
        const uint64_t cap = log_cap;
        if (int rc = reserve_idle(b, bin_cnt, std::max<size_t>(static_cast<size_t>(blocks), static_cast<size_t>(512) * 16u))) return rc;
        if (int rc = reserve_idle(b, bin_cnt2, static_cast<size_t>(512) * 16u)) return rc;
        if (b->set_used[ls]) HIPCHK(b, hipStreamWaitEvent(b->stream, b->ev_set_free[ls], 0));   // the passes of the launch that last wrote this set
        P.bin_list = bin_list.ptr;
        P.bin_cap = static_cast<uint32_t>(cap);
        P.bin_tiles = log_tiles;
        P.bin_shift = 0u;
        P.bin_cnt = bin_cnt.ptr;   // one fill count per region, written by the trace kernel
        P.bin_log = 1u;
        P.mono_copy_mask = 0u;   // logged slots and their fallbacks address copy 0
        P.log_xyz = use_log_xyz ? 1u : 0u;
        P.log_plane_stride = static_cast<uint32_t>((static_cast<uint64_t>(kMonoRows) << b->mono_s_log2) * b->plane_copies);
      } else
      if (use_bin) {
        // lists sized for ~6 hits per ray spread 4x unevenly; an overflowing list falls back to direct atomics
        // (the coarse and tile lists of the two-level route are slot ranges interleaved over the image — rows p % 1024 — and
        // stay balanced whatever the image shows: 2x there)
        const uint64_t slack = two_level ? 2ull : 4ull;
        uint64_t cap = std::max<uint64_t>(slack * 6ull * m / lists1, 1ull << 16);
        cap = std::min<uint64_t>(cap, (8ull << 30) / (8ull * lists1));
        if (int rc = reserve_idle(b, bin_cnt, static_cast<size_t>(512) * 16u)) return rc;
        HIPCHK(b, hipMemsetAsync(bin_cnt.ptr, 0, static_cast<size_t>(two_level ? 512u : bin_tiles) * 16u * sizeof(uint32_t), b->stream));
        if (int rc = reserve_idle(b, bin_list, cap * lists1)) return rc;
        P.bin_list = bin_list.ptr;
        P.bin_cap = static_cast<uint32_t>(cap);
        P.bin_tiles = two_level ? b->bin_l1 : bin_tiles;
        P.bin_shift = two_level ? 14u + fan_log2 : 0u;
        P.bin_cnt = bin_cnt.ptr;
        P.bin_log = 0u;
        P.log_xyz = 0u;
        P.mono_by_wl = b->mono_by_wl ? 1u : 0u;
        P.mono_copy_mask = 0u;   // staged hits and their fallbacks address copy 0
        if (two_level) {
          uint64_t c2 = std::max<uint64_t>(slack * 6ull * m / bin_tiles, 1ull << 12);
          c2 = std::min<uint64_t>(c2, (8ull << 30) / (8ull * bin_tiles));
          cap2 = static_cast<uint32_t>(c2 & ~15ull);   // whole 128-byte lines per tile list
          if (int rc = reserve_idle(b, bin_cnt2, static_cast<size_t>(bin_tiles) * 16u)) return rc;
          HIPCHK(b, hipMemsetAsync(bin_cnt2.ptr, 0, static_cast<size_t>(bin_tiles) * 16u * sizeof(uint32_t), b->stream));
          if (int rc = reserve_idle(b, bin_list2, c2 * bin_tiles)) return rc;
        }
      } else {
        P.bin_list = nullptr;
        P.bin_log = 0u;
        P.log_xyz = 0u;
        P.mono_by_wl = b->mono_by_wl ? 1u : 0u;
        P.mono_copy_mask = b->plane_copies - 1u;
      }
      P.no_land = (P.prob >= 1.0f && !P.final_layer && fast_mode && b->aggregate == 1) ? 1u : 0u;
      // The launch's trace stream (alternating for small launches), behind whatever the auxiliary stream still does to the planes (a closing
      // fold): the launch adds to them — from the trace kernel itself (direct atomics, staged lists) or in its passes, which follow it on
      // the same stream.
      hipStream_t ts = nullptr;
      int ts_i = 0;
      // (Round 6: a LOGGED trace kernel writes records, its region counts and the twin's other half — nothing the closing fold of the session
      //  before reads or zeroes — so on a SMALL launch it starts under that fold, and only its passes, which add to the planes, wait for it.  A
      //  discrete spectrum of many short sessions (bench.py --config 4d: 31 x 0.8 M rays) was one serial chain of generator, trace, split, sums
      //  and fold, 325 us per session; the trace kernel and generator of session k + 1 now run beside the passes and the fold of session k.
      //  Launches that fill the chip keep the old order: measured in round 5, nothing to gain there and the kernel's own span stretches.)
      const bool under_fold = use_log && alternate;
      if (int rc = next_trace_stream(b, !under_fold, alternate, &ts, &ts_i)) return rc;
      hipEvent_t before_sums = nullptr;   // rule (1), for the routes whose sums write the planes plainly
#ifndef HALO_NO_PLANE_GATES   // (defined only to see tests/test_gpu_production_routes.py's two-stream test fail without the gates)
      if (b->overlap && !use_log && b->rmw_pending[ts_i ^ 1])   // this trace kernel adds to the planes itself: behind the other stream's sums
        HIPCHK(b, hipStreamWaitEvent(ts, b->ev_rmw[ts_i ^ 1], 0));
      if (b->overlap && (use_log || (use_bin && two_level)) && b->cs_pending[ts_i ^ 1]) {
        HIPCHK(b, hipEventRecord(b->ev_gate, b->cs[ts_i ^ 1]));
        before_sums = b->ev_gate;
      }
#endif
      if (!deterministic && !host_pool) {
        // Device generator: one team per sampled crystal, in front of the trace kernel on its stream.  Option "gen_ahead" puts the one of a
        // launch that fills the chip on trace stream 1 instead, beside the previous launch's trace kernel and passes (it needs nothing of
        // them, only its pool back from the launch before that one).  Measured (25 M rays per launch): prism pools 4.43 -> 4.55 ms per launch,
        // pyramid pools 7.78 -> 7.70: both kernels are VALU-bound, so sharing the chip conserves the work.  Off by default.
        const bool ahead = b->overlap && b->gen_ahead && !alternate;
        hipStream_t gs = ahead ? b->cs[1] : ts;
        if (b->shapes_used[shape_set]) HIPCHK(b, hipStreamWaitEvent(gs, b->ev_shapes_free[shape_set], 0));
        hipError_t ge = launch_shapegen(const_cast<ShapeDev*>(P.shapes), geom == 2, shape_cnt, b->seed, host::MakeRecipe(E.crystal), first_shape, gs, b->gen_serial != 0);
        if (ge != hipSuccess) return hip_fail(b, ge, "halo_shapegen_kernel launch");
        if (ahead) {
          HIPCHK(b, hipEventRecord(b->ev_gen, gs));
          HIPCHK(b, hipStreamWaitEvent(ts, b->ev_gen, 0));
          b->cs_pending[1] = true;
        }
      }
      HIPCHK(b, hipEventRecord(b->ring_ev0[k], ts));  // HIP events on the launch stream bracket the kernel alone
      b->ring_final[k] = P.final_layer != 0u;
      // a one-shape dispatch of a regular hexagonal prism takes the literal-normal instantiation (kGeomOneHex = 3)
      const int launch_geom = (geom == 0 && entry_fast && b->hex_fast && hex_regular) ? 3 : geom;
      hipError_t le = launch_trace(P, blocks, ts, mode, launch_geom, b->mono_session);
      if (ce) {   // the slot's last reader (see TableCacheEntry): this launch, whose ring_done event is recorded below
        ce->read_ring[ce->cur] = k;
        ce->read_use[ce->cur] = b->ring_use[k];
      }
      b->mono_dirty = true;
      b->route.launches++;
      b->route.mode_mask |= 1u << mode;
      // the masks name the instantiation launch_mode / launch_mono pick (halo_trace.inl), not the request: the regular-prism search exists for
      // the production-shaped kernels off the binned route, the no-accumulation kernels for their one-shape dispatches
      const bool ran_hex = launch_geom == 3 && fast_mode && (P.bin_list == nullptr || P.bin_log != 0u || P.no_land != 0u);
      const bool ran_none = P.no_land != 0u && geom == 0;
      b->route.geom_mask |= 1u << (launch_geom == 3 && !ran_hex ? 0 : launch_geom);
      b->route.accum_mask |= ran_none ? 64u : use_log ? (use_log_xyz ? 32u : 16u) : (use_bin ? (two_level ? 8u : 4u) : (b->mono_session ? 2u : 1u));
      // specialisations (plain kernels only; launch_lens / launch_vis / launch_mono): bit 0 last-layer kernel (no continuation code), 1 lens as a
      // constant, 2 visible range as a constant, 3 closed gate as a constant
      {
        const bool one = geom == 0, lens_known = P.proj.proj_type == HALO_LENS_LINEAR || P.proj.proj_type == HALO_LENS_FISHEYE_EQUAL_AREA ||
                                                 P.proj.proj_type == HALO_LENS_DUAL_FISHEYE_EQUAL_AREA || P.proj.proj_type == HALO_LENS_RECTANGULAR;
        const bool vis_known = P.proj.visible_range == HALO_VISIBLE_UPPER || P.proj.visible_range == HALO_VISIBLE_FULL;
        uint32_t spec = 0u;
        if (fast_mode && use_log && one && !ran_none && b->mono_session && P.final_layer) {
          spec |= 1u;
          if ((mode == 0 || mode == 1) && P.prob <= 0.0f && lens_known && vis_known) spec |= 2u | 4u | 8u;
        }
        if (fast_mode && use_log && one && !ran_none && b->mono_session && !P.final_layer && (mode == 0 || mode == 1) && lens_known && vis_known)
          spec |= 2u | 4u;   // (round 6: the logging kernels of the layers before the last know their lens and visible range too; their gate is open)
        if (mode == 0 && use_log && !one && !b->mono_session && P.prob <= 0.0f && P.proj.visible_range == HALO_VISIBLE_FULL) spec |= 4u | 8u;
        if (mode == 0 && use_log && !one && P.prob <= 0.0f && P.proj.visible_range == HALO_VISIBLE_FULL && P.proj.proj_type == HALO_LENS_RECTANGULAR)
          spec |= 2u | 4u | 8u;   // (shape-pool kernels, either plane layout: bench_config_stoch.json's render as constants)
        b->route.spec_mask |= spec;
        if (spec == 0u) b->route.generic_launches++;
      }
      b->route.source_mask |= 1u << P.source;
      if (le != hipSuccess) return hip_fail(b, le, "halo_trace_kernel launch");
      if (!deterministic) {
        HIPCHK(b, hipEventRecord(b->ev_shapes_free[shape_set], ts));
        b->shapes_used[shape_set] = true;
      }
      // fixed-point scale of the per-tile sums (halo_kernels.hip FixQ): no slot of this launch can sum to more than 4 x max weight x rays
      const uint32_t frac_bits = fix_frac_bits(b->sess_max_w, m);
      if (use_log) {
        // the accumulation passes of this launch: behind the trace kernel on its stream.  (Round 5 tried them on the auxiliary stream, under the
        // next launch's trace kernel: configs[1] + 1.2 %, the trace kernel itself 5 % slower for the CUs the 134 KB split workgroups take, the
        // passes' spans stretched sevenfold — profiles that no longer say what a kernel costs, for one percent.  Not kept.)
        HIPCHK(b, hipEventRecord(b->ring_ev1[k], ts));          // the trace kernel alone: ev0 .. ev1
        hipStream_t ps = ts;
        if (under_fold) {   // the passes add to the planes: behind the fold that last touched THEIR planes
          if (b->mono_two) {   // two plane sets: that is the fold of two sessions ago (it zeroed this set), not the one the trace kernel started under
            if (b->set_folded_pending[b->mono_set]) HIPCHK(b, hipStreamWaitEvent(ps, b->ev_set_folded[b->mono_set], 0));
          } else if (int rc = wait_for_aux(b, ts_i)) return rc;
        }
        HIPCHK(b, hipEventRecord(b->ring_ev2[k], ps));
        HIPCHK(b, hipMemsetAsync(bin_cnt2.ptr, 0, static_cast<size_t>(log_tiles) * 16u * sizeof(uint32_t), ps));
        hipError_t be = use_log_xyz ? launch_log_route_xyz(P.mono, P.log_plane_stride, bin_list.ptr, P.bin_cap, bin_cnt.ptr, static_cast<uint32_t>(blocks),
                                                           bin_list2.ptr, cap2, bin_cnt2.ptr, log_tiles, b->mono_s_log2, P.wl_pool, P.wl_pool_size, frac_bits, P.ovf, P.ovf_flag, P.ovf_copies_log2, ps, before_sums)
                                    : launch_log_route(P.mono, bin_list.ptr, P.bin_cap, bin_cnt.ptr, static_cast<uint32_t>(blocks), bin_list2.ptr, cap2,
                                                       bin_cnt2.ptr, 1u << log_t_log2, log_planes, b->mono_s_log2, b->render.visible == HALO_VISIBLE_FULL, frac_bits,
                                                       P.ovf, P.ovf_flag, P.ovf_copies_log2, ps, before_sums);
        if (be != hipSuccess) return hip_fail(b, be, "halo_split_kernel launch");
        if (b->overlap) {
          HIPCHK(b, hipEventRecord(b->ev_rmw[ts_i], ps));
          b->rmw_pending[ts_i] = true;
        }
        HIPCHK(b, hipEventRecord(b->ring_ev3[k], ps));
        HIPCHK(b, hipEventRecord(b->ev_set_free[ls], ps));
        b->set_used[ls] = true;
        if (b->overlap && alternate) b->log_set ^= 1;
        b->ring_posts[k] = true;
        HIPCHK(b, hipEventRecord(b->ring_done[k], ps));
      } else {
      if (use_bin) {
        hipError_t be = two_level ? launch_bin_two_level(P.mono, bin_list.ptr, P.bin_cap, bin_cnt.ptr, lists1, bin_list2.ptr, cap2, bin_cnt2.ptr,
                                                         bin_tiles, fan_log2, frac_bits, P.ovf, P.ovf_flag, ts, before_sums)
                                  : launch_bin_accumulate(P.mono, bin_list.ptr, P.bin_cap, bin_cnt.ptr, bin_tiles, frac_bits, ts);
        if (be != hipSuccess) return hip_fail(b, be, "halo_bin_accumulate_kernel launch");
        if (b->overlap && two_level) {
          HIPCHK(b, hipEventRecord(b->ev_rmw[ts_i], ts));
          b->rmw_pending[ts_i] = true;
        }
      }
      HIPCHK(b, hipEventRecord(b->ring_ev1[k], ts));
      b->ring_posts[k] = false;
      HIPCHK(b, hipEventRecord(b->ring_done[k], ts));
      }
      b->tally_unread = true;
      b->ring_busy[k] = true;
      if (host_pool) HIPCHK(b, hipStreamSynchronize(b->stream));  // the pageable shape pool must outlive its copy (the copy is on the main stream)
      if (P.source == kSrcGen) b->gen_count += m;
      if (P.source == kSrcTransit) b->transit_count += m;
      b->gate_count += m;
      off += m;
    }
    ci_start += n_ci;
  }
  if (defer) {  // queued, not waited for: the tallies arrive through halo_collect_stats
    b->layer_acc.root_count += n;
    if (stats) {
      std::memset(stats, 0, sizeof(*stats));
      stats->root_count = n;
    }
    b->cont_in_n = 0;
    return HALO_OK;
  }
  if (int rc = join_aux(b)) return rc;   // the copies below (and the caller) want this layer's kernels done: the main stream waits for the trace streams
  uint32_t cnt[kCntNum] = {0, 0, 0, 0};
  std::vector<uint32_t> fill(final_layer ? 0 : kContShards * kContCntStride);
  HIPCHK(b, hipMemcpyAsync(cnt, b->counters.ptr, sizeof(cnt), hipMemcpyDeviceToHost, b->stream));
  if (!final_layer) HIPCHK(b, hipMemcpyAsync(fill.data(), b->cont_cnt.ptr, fill.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
  if (b->tally_unread) {
    if (int rc = pull_tally(b)) return rc;   // (one more small copy and THE stream sync of this call: the two copies above are complete behind it)
  } else {
    HIPCHK(b, hipStreamSynchronize(b->stream));
  }
  harvest_all(b);
  uint64_t n_cont = 0;
  if (!final_layer) {  // the next layer addresses the pool through the prefix of the shard fill counts
    for (int sI = 0; sI < kContShards; sI++) {
      const uint32_t c = fill[static_cast<size_t>(sI) * kContCntStride];
      if (c > out_cap) return fail(b, HALO_FATAL, "continuation pool overflow");
      b->cont_seg[sI] = static_cast<uint32_t>(n_cont);
      n_cont += c;
    }
    b->cont_seg[kContShards] = static_cast<uint32_t>(n_cont);
  }
  b->exits_pending = std::min<uint64_t>(cnt[kCntExit], b->exits.cap);
  b->layer_acc.root_count = n;
  b->layer_acc.continuation_count = n_cont;
  if (stats) *stats = b->layer_acc;
  add_stats(b->pending, b->layer_acc);
  b->layer_acc = HaloLayerStats{};
  b->cont_in_n = n_cont;  // becomes the next layer's input at Recombine
  return HALO_OK;
}

int halo_generate_shapes(halo_handle_t b, const HaloCrystal* crystal, uint64_t first_index, uint32_t n, int on_device, HaloGeomTables* out) {
  if (!b || !crystal || (!out && n)) return HALO_FATAL;
  if (crystal->kind != HALO_CRYSTAL_PRISM && crystal->kind != HALO_CRYSTAL_PYRAMID) return fail(b, HALO_FATAL, "unknown crystal kind");
  std::vector<ShapeDev> pool(n);
  if (on_device == 2 && crystal->kind == HALO_CRYSTAL_PRISM) {   // the prism-pool generator (1360-byte ShapePrism records, one team of 16 lanes per crystal)
    HIPCHK(b, hipSetDevice(b->device));
    DevBuf<ShapePrism> dev;
    HIPCHK(b, dev.reserve(n));
    std::vector<ShapePrism> recs(n);
    hipError_t ge = launch_shapegen(dev.ptr, true, n, b->seed, host::MakeRecipe(*crystal), first_index, b->stream, b->gen_serial != 0);
    if (ge == hipSuccess) ge = hipMemcpyAsync(recs.data(), dev.ptr, static_cast<size_t>(n) * sizeof(ShapePrism), hipMemcpyDeviceToHost, b->stream);
    if (ge == hipSuccess) ge = hipStreamSynchronize(b->stream);
    dev.release();
    if (ge != hipSuccess) return hip_fail(b, ge, "halo_prismgen_team_kernel");
    for (uint32_t k = 0; k < n; k++) {   // same members, shorter rows: widen to the general record
      const ShapePrism& r = recs[k];
      ShapeDev& d = pool[k];
      std::memset(&d, 0, sizeof(d));
      d.face_cnt = r.face_cnt, d.tri_cnt = r.tri_cnt, d.slab_cnt = r.slab_cnt, d.single_cnt = r.single_cnt;
      std::memcpy(d.face, r.face, sizeof(r.face));
      std::memcpy(d.slab, r.slab, sizeof(r.slab));
      std::memcpy(d.tri_v, r.tri_v, sizeof(r.tri_v));
      std::memcpy(d.tri_na, r.tri_na, sizeof(r.tri_na));
      std::memcpy(d.tri_face, r.tri_face, sizeof(r.tri_face));
      std::memcpy(d.face_number, r.face_number, sizeof(r.face_number));
      std::memcpy(d.single, r.single, sizeof(r.single));
    }
  } else if (on_device) {
    HIPCHK(b, hipSetDevice(b->device));
    DevBuf<ShapeDev> dev;
    HIPCHK(b, dev.reserve(n));
    hipError_t ge = launch_shapegen(dev.ptr, false, n, b->seed, host::MakeRecipe(*crystal), first_index, b->stream, b->gen_serial != 0);
    if (ge == hipSuccess) ge = hipMemcpyAsync(pool.data(), dev.ptr, static_cast<size_t>(n) * sizeof(ShapeDev), hipMemcpyDeviceToHost, b->stream);
    if (ge == hipSuccess) ge = hipStreamSynchronize(b->stream);
    dev.release();
    if (ge != hipSuccess) return hip_fail(b, ge, "halo_shapegen_kernel");
  } else {
    for (uint32_t k = 0; k < n; k++) host::MakeShapeDev(b->seed, *crystal, first_index + k, pool[k]);
  }
  for (uint32_t k = 0; k < n; k++) host::FromShapeDev(pool[k], out[k]);
  return HALO_OK;
}

int halo_set_color(halo_handle_t b, const HaloColorSet* sets, int n_sets, const HaloColorClass* classes, int n_classes) {
  if (!b || n_sets < 0 || n_classes < 0 || (n_sets && !sets) || (n_classes && !classes)) return HALO_FATAL;
  if (b->in_session) return fail(b, HALO_FATAL, "halo_set_color inside a session");
  if (n_classes > HALO_COLOR_MAX_CLASSES) return fail(b, HALO_FATAL, "more than HALO_COLOR_MAX_CLASSES colour classes");
  for (int i = 0; i < n_sets; i++) {
    if (sets[i].term_count < 0 || sets[i].term_count > HALO_COLOR_MAX_TERMS) return fail(b, HALO_FATAL, "colour set term_count out of range");
    for (int k = 0; k < sets[i].term_count; k++)
      if (sets[i].terms[k].bit < 0 || sets[i].terms[k].bit > 63) return fail(b, HALO_FATAL, "colour bit outside 0..63");
  }
  const bool same = static_cast<size_t>(n_sets) == b->color_sets.size() && static_cast<size_t>(n_classes) == b->color_classes.size() &&
                    (n_sets == 0 || std::memcmp(b->color_sets.data(), sets, sizeof(HaloColorSet) * static_cast<size_t>(n_sets)) == 0) &&
                    (n_classes == 0 || std::memcmp(b->color_classes.data(), classes, sizeof(HaloColorClass) * static_cast<size_t>(n_classes)) == 0);
  if (same) return HALO_OK;   // (handed over again unchanged, as the Lumice glue does at every BeginSession: the table cache stays)
  b->color_sets.assign(sets, sets + n_sets);
  b->color_classes.assign(classes, classes + n_classes);
  drop_table_cache(b);
  return HALO_OK;
}

int halo_readback_class_lanes(halo_handle_t b, float* lanes, int width, int height, int class_count) {
  if (!b || !lanes) return HALO_FATAL;
  if (class_count != static_cast<int>(b->color_classes.size()) || width != b->lanes_w || height != b->lanes_h || !b->lanes.ptr)
    return fail(b, HALO_FATAL, "class lanes: size does not match the session (classes x width x height)");
  HIPCHK(b, hipSetDevice(b->device));
  if (int rc = join_aux(b)) return rc;   // (kernels of queued sessions may still be running on the trace / auxiliary streams)
  const size_t n = static_cast<size_t>(class_count) * width * height;
  // the lanes are summed in fp64 on the device (hot pixels), the seam hands out floats (trace_backend.hpp:471-493): narrowed and drained by
  // one kernel into the staging buffer, so half the bytes cross the bus and no host pass follows
  HIPCHK(b, b->lanes_stage.reserve(n));
  hipError_t e = launch_lanes_drain(b->lanes.ptr, b->lanes_stage.ptr, n, b->cu_count * 8, b->stream);
  if (e != hipSuccess) return hip_fail(b, e, "halo_lanes_drain_kernel launch");
  HIPCHK(b, hipMemcpyAsync(lanes, b->lanes_stage.ptr, n * sizeof(float), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(b, hipStreamSynchronize(b->stream));
  return HALO_OK;
}

int halo_last_sample_counts(halo_handle_t b, uint64_t* crystal_samples, uint64_t* orientation_samples) {
  if (!b) return HALO_FATAL;
  if (crystal_samples) *crystal_samples = b->sess_crystal_samples;
  if (orientation_samples) *orientation_samples = b->sess_orient_samples;
  return HALO_OK;
}

int halo_last_route(halo_handle_t b, HaloRouteInfo* out) {
  if (!b || !out) return HALO_FATAL;
  *out = b->route;
  return HALO_OK;
}

int halo_reduce_accumulator(halo_handle_t b, void* nccl_comm, int root, int this_rank) {
  if (!b || !nccl_comm) return HALO_FATAL;
  if (!b->acc || b->acc_w <= 0) return fail(b, HALO_FATAL, "reduce_accumulator before any session");
  // ncclReduce(sendbuff, recvbuff, count, ncclFloat32 = 7, ncclSum = 0, root, comm, stream) — rccl.h:550
  typedef int (*nccl_reduce_fn)(const void*, void*, size_t, int, int, int, void*, hipStream_t);
  static nccl_reduce_fn reduce = nullptr;   // looked up once, whichever GPU's thread gets here first (one thread per GPU is a supported use)
  static std::once_flag reduce_once;
  std::call_once(reduce_once, [] {
    void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (lib) reduce = reinterpret_cast<nccl_reduce_fn>(dlsym(lib, "ncclReduce"));
  });
  if (!reduce) return fail(b, HALO_UNAVAILABLE, "RCCL (librccl.so: ncclReduce) cannot be loaded");
  HIPCHK(b, hipSetDevice(b->device));
  int rc = fold_if_dirty(b);   // an unfinished session still owes its planes to the accumulator
  if (rc != HALO_OK) return rc;
  const size_t n = static_cast<size_t>(b->acc_w) * static_cast<size_t>(b->acc_h) * 3 + 4;
  if (n > b->acc_floats) return fail(b, HALO_FATAL, "reduce_accumulator: the bound accumulator is smaller than width*height*3+4 floats");
  const int nr = reduce(b->acc, b->acc, n, 7 /* ncclFloat32 */, 0 /* ncclSum */, root, nccl_comm, b->stream);
  if (nr != 0) return fail(b, HALO_FATAL, "ncclReduce failed with code " + std::to_string(nr));
  if (this_rank != root) HIPCHK(b, hipMemsetAsync(b->acc, 0, n * sizeof(float), b->stream));   // drained, in stream order
  return HALO_OK;
}

int halo_collect_stats(halo_handle_t b, HaloLayerStats* out) {
  if (!b || !out) return HALO_FATAL;
  HIPCHK(b, hipSetDevice(b->device));
  if (int rc = join_aux(b)) return rc;   // (kernels of queued sessions may still be running on the trace / auxiliary streams)
  HIPCHK(b, hipStreamSynchronize(b->stream));
  harvest_all(b);
  if (int rc = pull_tally(b)) return rc;
  add_stats(b->pending, b->layer_acc);
  b->layer_acc = HaloLayerStats{};
  *out = b->pending;
  b->pending = HaloLayerStats{};
  return HALO_OK;
}

int halo_collect_timing(halo_handle_t b, double* trace_ms, double* post_ms, uint64_t* launches) {
  if (!b) return HALO_FATAL;
  HIPCHK(b, hipSetDevice(b->device));
  if (int rc = sync_all(b)) return rc;
  harvest_all(b);   // (adds the kernel times to the layer tallies as well: each ring slot is harvested once, whoever asks first)
  if (trace_ms) *trace_ms = b->time_trace_ms;
  if (post_ms) *post_ms = b->time_post_ms;
  if (launches) *launches = b->time_launches;
  b->time_trace_ms = b->time_post_ms = 0.0;
  b->time_launches = 0;
  return HALO_OK;
}

int halo_recombine(halo_handle_t b, int shuffle, uint64_t* continuation_count) {
  if (!b) return HALO_FATAL;
  if (!b->in_session) return fail(b, HALO_FATAL, "Recombine outside a session");
  // No data moves: the pools swap roles and the Feistel permutation (shuffle_cont_kernel, cu:1633-1657) is
  // applied as a gather index by the next layer's kernel.
  b->cont_out_slot ^= 1;
  b->cont_shuffle = shuffle ? 1 : 0;
  b->layer_idx++;
  if (continuation_count) *continuation_count = b->cont_in_n;
  return HALO_OK;
}

int halo_drain_exits(halo_handle_t b, HaloExitRecord* out, uint64_t cap, uint64_t* count) {
  if (!b) return HALO_FATAL;
  HIPCHK(b, hipSetDevice(b->device));
  if (int rc = join_aux(b)) return rc;   // (kernels of queued sessions may still be running on the trace / auxiliary streams)
  HIPCHK(b, hipStreamSynchronize(b->stream));
  // Piecewise drain (trace_backend.hpp:430-448): a call takes at most `cap` records from the front and the rest stays
  // pending; *count = records copied by THIS call.  out == NULL reports how many are pending without consuming any.
  const uint64_t n = b->exits_pending;
  if (!out) {
    if (count) *count = n;
    return HALO_OK;
  }
  const uint64_t take = std::min(n, cap);
  if (count) *count = take;
  if (take) HIPCHK(b, hipMemcpy(out, b->exits.ptr, take * sizeof(HaloExitRecord), hipMemcpyDeviceToHost));
  const uint64_t left = n - take;
  if (left && take) {  // move the tail to the front (ranges may overlap: go through a scratch copy)
    DevBuf<HaloExitRecord> tmp;
    HIPCHK(b, tmp.reserve(left));
    HIPCHK(b, hipMemcpy(tmp.ptr, b->exits.ptr + take, left * sizeof(HaloExitRecord), hipMemcpyDeviceToDevice));
    HIPCHK(b, hipMemcpy(b->exits.ptr, tmp.ptr, left * sizeof(HaloExitRecord), hipMemcpyDeviceToDevice));
    tmp.release();
  }
  b->exits_pending = left;
  const uint32_t left32 = static_cast<uint32_t>(left);  // the device append counter continues behind the kept tail
  HIPCHK(b, hipMemcpyAsync(b->counters.ptr + kCntExit, &left32, sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
  HIPCHK(b, hipStreamSynchronize(b->stream));
  return HALO_OK;
}

int halo_flush(halo_handle_t b) {
  if (!b) return HALO_FATAL;
  if (b->in_session) return fail(b, HALO_FATAL, "halo_flush inside a session");
  HIPCHK(b, hipSetDevice(b->device));
  return fold_if_dirty(b);   // queued; the backend's stream waits for it, the host does not
}

int halo_sync(halo_handle_t b) {
  if (!b) return HALO_FATAL;
  HIPCHK(b, hipSetDevice(b->device));
  // with option defer_fold a caller-bound accumulator is brought up to date HERE (and by every reader): the closing folds of the ended sessions
  if (b->defer_fold && !b->in_session)
    if (int rc = fold_if_dirty(b)) return rc;
  return sync_all(b);
}

int halo_take_landed(halo_handle_t b, double* landed) {
  if (!b || !landed) return HALO_FATAL;
  HIPCHK(b, hipSetDevice(b->device));
  {
    int rc = fold_if_dirty(b);
    if (rc != HALO_OK) return rc;
  }
  return take_landed_delta(b, landed);
}

int halo_readback_xyz64(halo_handle_t b, float* xyz, int width, int height, double* landed) {
  if (!b || !xyz) return HALO_FATAL;
  if (!b->acc || width != b->acc_w || height != b->acc_h) return fail(b, HALO_FATAL, "readback size does not match the session render");
  HIPCHK(b, hipSetDevice(b->device));
  {
    int rc = fold_if_dirty(b);
    if (rc != HALO_OK) return rc;
  }
  const size_t n = static_cast<size_t>(width) * height * 3;
  HIPCHK(b, hipMemcpyAsync(xyz, b->acc, n * sizeof(float), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(b, hipMemsetAsync(b->acc, 0, (n + 4) * sizeof(float), b->stream));  // readback ZEROES the accumulator (cu:4832-4846)
  double l = 0.0;
  if (int rc = take_landed_delta(b, &l)) return rc;   // (syncs the stream: the image copy above is complete behind it)
  if (landed) *landed = l;
  return HALO_OK;
}

int halo_readback_xyz(halo_handle_t b, float* xyz, int width, int height, float* landed_weight) {
  double l = 0.0;
  int rc = halo_readback_xyz64(b, xyz, width, height, &l);
  if (rc == HALO_OK && landed_weight) *landed_weight += static_cast<float>(l);  // ADDS (trace_backend.hpp:461-469)
  return rc;
}

// ---- consumer on device ------------------------------------------------------------------------------------
int halo_consumer_reset(halo_handle_t b) {
  if (!b) return HALO_FATAL;
  HIPCHK(b, hipSetDevice(b->device));
  if (int rc = join_aux(b)) return rc;   // (kernels of queued sessions may still be running on the trace / auxiliary streams)
  if (b->cons_sum.ptr) {
    const size_t n = static_cast<size_t>(b->cons_w) * b->cons_h * 3;
    HIPCHK(b, hipMemsetAsync(b->cons_sum.ptr, 0, n * sizeof(float), b->stream));
    HIPCHK(b, hipMemsetAsync(b->cons_comp.ptr, 0, n * sizeof(float), b->stream));
  }
  if (b->lanes.ptr) HIPCHK(b, hipMemsetAsync(b->lanes.ptr, 0, b->lanes.cap * sizeof(double), b->stream));   // RenderConsumer::Reset clears the class lanes too (render.cpp:598-612)
  b->total_intensity = 0.0;
  return HALO_OK;
}

int halo_consumer_fold(halo_handle_t b) {
  if (!b) return HALO_FATAL;
  if (!b->acc || b->acc_w <= 0) return fail(b, HALO_FATAL, "consumer_fold before any session");
  HIPCHK(b, hipSetDevice(b->device));
  int rc = fold_if_dirty(b);
  if (rc != HALO_OK) return rc;
  const size_t n = static_cast<size_t>(b->acc_w) * b->acc_h * 3;
  if (b->cons_w != b->acc_w || b->cons_h != b->acc_h || !b->cons_sum.ptr) {
    HIPCHK(b, b->cons_sum.reserve(n));
    HIPCHK(b, b->cons_comp.reserve(n));
    b->cons_w = b->acc_w;
    b->cons_h = b->acc_h;
    HIPCHK(b, hipMemsetAsync(b->cons_sum.ptr, 0, n * sizeof(float), b->stream));
    HIPCHK(b, hipMemsetAsync(b->cons_comp.ptr, 0, n * sizeof(float), b->stream));
    b->total_intensity = 0.0;
  }
  hipError_t e = launch_consumer_fold(b->acc, b->cons_sum.ptr, b->cons_comp.ptr, static_cast<uint32_t>(n), b->cu_count * 8, b->stream);
  if (e != hipSuccess) return hip_fail(b, e, "halo_consumer_fold_kernel launch");
  double landed = 0.0;
  if (int rc = take_landed_delta(b, &landed)) return rc;
  b->total_intensity += landed;  // total_intensity_ += xyz_landed_weight_ (render.cpp:149)
  return HALO_OK;
}

int halo_consumer_consume(halo_handle_t b, const float* xyz, int width, int height, float landed, const float* lanes, int class_count) {
  if (!b || !xyz) return HALO_FATAL;
  if (b->in_session) return fail(b, HALO_FATAL, "consumer_consume inside a session");
  if (width <= 0 || height <= 0) return fail(b, HALO_FATAL, "consumer_consume: empty image");
  if (static_cast<uint64_t>(width) * static_cast<uint64_t>(height) > (1ull << 25)) return fail(b, HALO_UNAVAILABLE, "more than 2^25 pixels");
  if (b->cons_sum.ptr && (b->cons_w != width || b->cons_h != height)) return fail(b, HALO_FATAL, "consumer_consume: image size differs from the consumer's (halo_consumer_reset first)");
  HIPCHK(b, hipSetDevice(b->device));
  if (int rc = join_aux(b)) return rc;   // (kernels of queued sessions may still be running on the trace / auxiliary streams)
  const size_t n = static_cast<size_t>(width) * height * 3;
  if (!b->cons_sum.ptr) {
    HIPCHK(b, b->cons_sum.reserve(n));
    HIPCHK(b, b->cons_comp.reserve(n));
    b->cons_w = width;
    b->cons_h = height;
    HIPCHK(b, hipMemsetAsync(b->cons_sum.ptr, 0, n * sizeof(float), b->stream));
    HIPCHK(b, hipMemsetAsync(b->cons_comp.ptr, 0, n * sizeof(float), b->stream));
    b->total_intensity = 0.0;
  }
  // the caller's image goes through the staging buffer and the same Neumaier fold kernel the device accumulator takes (the kernel zeroes
  // what it folds: the staging buffer here)
  HIPCHK(b, b->cons_stage.reserve(n));
  HIPCHK(b, hipMemcpyAsync(b->cons_stage.ptr, xyz, n * sizeof(float), hipMemcpyHostToDevice, b->stream));
  hipError_t e = launch_consumer_fold(b->cons_stage.ptr, b->cons_sum.ptr, b->cons_comp.ptr, static_cast<uint32_t>(n), b->cu_count * 8, b->stream);
  if (e != hipSuccess) return hip_fail(b, e, "halo_consumer_fold_kernel launch");
  if (lanes != nullptr && class_count > 0) {   // lane_pixel_data_: dst[p] += src[p] per class (render.cpp:176-185)
    if (class_count != static_cast<int>(b->color_classes.size())) return fail(b, HALO_FATAL, "consumer_consume: class count must equal halo_set_color's");
    const size_t nl = static_cast<size_t>(class_count) * width * height;
    if (!b->lanes.ptr || b->lanes_w != width || b->lanes_h != height) {
      HIPCHK(b, b->lanes.reserve(nl));
      HIPCHK(b, hipMemsetAsync(b->lanes.ptr, 0, b->lanes.cap * sizeof(double), b->stream));
      b->lanes_w = width;
      b->lanes_h = height;
    }
    HIPCHK(b, b->lanes_stage.reserve(nl));
    HIPCHK(b, hipMemcpyAsync(b->lanes_stage.ptr, lanes, nl * sizeof(float), hipMemcpyHostToDevice, b->stream));
    e = launch_lanes_add(b->lanes_stage.ptr, b->lanes.ptr, nl, b->cu_count * 8, b->stream);
    if (e != hipSuccess) return hip_fail(b, e, "halo_lanes_add_kernel launch");
  }
  HIPCHK(b, hipStreamSynchronize(b->stream));   // `xyz` / `lanes` are the caller's (pageable) memory
  b->total_intensity += static_cast<double>(landed);   // total_intensity_ += xyz_landed_weight_ (render.cpp:149)
  return HALO_OK;
}

int halo_consumer_snapshot(halo_handle_t b, const HaloDisplay* dsp, uint8_t* rgb_out, float* xyz_out, double* total_intensity) {
  if (!b || !dsp) return HALO_FATAL;
  if (!b->cons_sum.ptr) return fail(b, HALO_FATAL, "consumer_snapshot before consumer_fold");
  HIPCHK(b, hipSetDevice(b->device));
  const uint32_t npix = static_cast<uint32_t>(b->cons_w) * static_cast<uint32_t>(b->cons_h);
  const size_t n = static_cast<size_t>(npix) * 3;
  // ExposureScale (render.cpp:96-102), evaluated in float like the reference
  const float snapshot_intensity = static_cast<float>(b->total_intensity);
  const float scale = (npix == 0 || snapshot_intensity <= 0.0f) ? 0.0f : dsp->intensity_factor * 0.08f * static_cast<float>(npix) / snapshot_intensity;
  if (rgb_out) HIPCHK(b, b->cons_rgb.reserve(n));
  if (xyz_out) HIPCHK(b, b->cons_xyz_out.reserve(n));
  hipError_t e = launch_post_snapshot(b->cons_sum.ptr, b->cons_comp.ptr, rgb_out ? b->cons_rgb.ptr : nullptr,
                                      xyz_out ? b->cons_xyz_out.ptr : nullptr, npix, scale, dsp->ray_color, dsp->background,
                                      b->cu_count * 8, b->stream);
  if (e != hipSuccess) return hip_fail(b, e, "halo_post_snapshot_kernel launch");
  if (rgb_out) HIPCHK(b, hipMemcpyAsync(rgb_out, b->cons_rgb.ptr, n, hipMemcpyDeviceToHost, b->stream));
  if (xyz_out) HIPCHK(b, hipMemcpyAsync(xyz_out, b->cons_xyz_out.ptr, n * sizeof(float), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(b, hipStreamSynchronize(b->stream));
  if (rgb_out && scale == 0.0f) std::memset(rgb_out, 0, n);  // PostSnapshot early-out (render.cpp:515-518)
  if (total_intensity) *total_intensity = b->total_intensity;
  return HALO_OK;
}

// ---- display-side composite of the class lanes (server/component_compositor.cpp) -------------------------
int halo_host_parse_composite_mode(const char* mode) {   // ParseCompositeMode, component_compositor.cpp:118-134
  if (mode && std::strcmp(mode, "dominant") == 0) return HALO_COMPOSITE_DOMINANT;
  if (mode && std::strcmp(mode, "additive") == 0) return HALO_COMPOSITE_ADDITIVE;
  return HALO_COMPOSITE_PAINTER;
}

int halo_consumer_load_lanes(halo_handle_t b, const float* lanes, int width, int height, int class_count, double total_intensity) {
  if (!b || !lanes) return HALO_FATAL;
  if (b->in_session) return fail(b, HALO_FATAL, "consumer_load_lanes inside a session");
  if (width <= 0 || height <= 0 || class_count <= 0 || class_count != static_cast<int>(b->color_classes.size()))
    return fail(b, HALO_FATAL, "consumer_load_lanes: class count must equal halo_set_color's, image must not be empty");
  if (static_cast<uint64_t>(width) * static_cast<uint64_t>(height) > (1ull << 25)) return fail(b, HALO_UNAVAILABLE, "more than 2^25 pixels");
  HIPCHK(b, hipSetDevice(b->device));
  if (int rc = join_aux(b)) return rc;   // (kernels of queued sessions may still be running on the trace / auxiliary streams)
  const size_t n = static_cast<size_t>(class_count) * width * height;
  HIPCHK(b, b->lanes.reserve(n));
  b->lanes_w = width;
  b->lanes_h = height;
  HIPCHK(b, b->lanes_stage.reserve(n));
  HIPCHK(b, hipMemcpyAsync(b->lanes_stage.ptr, lanes, n * sizeof(float), hipMemcpyHostToDevice, b->stream));
  hipError_t e = launch_lanes_load(b->lanes_stage.ptr, b->lanes.ptr, n, b->cu_count * 8, b->stream);
  if (e != hipSuccess) return hip_fail(b, e, "halo_lanes_load_kernel launch");
  HIPCHK(b, hipStreamSynchronize(b->stream));   // `lanes` is the caller's (pageable) memory
  if (total_intensity >= 0.0) b->total_intensity = total_intensity;
  return HALO_OK;
}

int halo_consumer_composite(halo_handle_t b, const HaloComposite* spec, float* linear_rgb_out, uint8_t* srgb_out, float* participating_p99_y, int32_t* produced) {
  if (!b || !spec || !produced) return HALO_FATAL;
  *produced = 0;
  if (spec->mode < HALO_COMPOSITE_DOMINANT || spec->mode > HALO_COMPOSITE_PAINTER) return fail(b, HALO_FATAL, "consumer_composite: unknown mode");
  uint64_t referenced = 0;   // ColorClassTable::referenced_mask_ = OR of the classes' member bits (color_class_table.hpp:38-43)
  for (const HaloColorClass& c : b->color_classes) referenced |= c.bits;
  if (referenced == 0) return HALO_OK;   // component_compositor.cpp:183-185: nothing is touched
  if (spec->class_count != static_cast<int>(b->color_classes.size())) return fail(b, HALO_FATAL, "consumer_composite: class_count differs from halo_set_color's");
  if (!b->lanes.ptr || b->lanes_w <= 0 || b->lanes_h <= 0) return fail(b, HALO_FATAL, "consumer_composite before any colour session (no lanes)");
  HIPCHK(b, hipSetDevice(b->device));
  if (int rc = join_aux(b)) return rc;   // (kernels of queued sessions may still be running on the trace / auxiliary streams)
  const uint32_t npix = static_cast<uint32_t>(b->lanes_w) * static_cast<uint32_t>(b->lanes_h);
  const size_t n3 = static_cast<size_t>(npix) * 3u;
  // GatherActiveClasses (component_compositor.cpp:24-54): solo beats visible; stable sort by z_order; lane binding by class index
  bool any_solo = false;
  for (int c = 0; c < spec->class_count; c++) any_solo = any_solo || spec->classes[c].solo != 0;
  std::vector<int> order(static_cast<size_t>(spec->class_count));
  for (int c = 0; c < spec->class_count; c++) order[static_cast<size_t>(c)] = c;
  std::stable_sort(order.begin(), order.end(), [spec](int x, int y) { return spec->classes[x].z_order < spec->classes[y].z_order; });
  CompositeDev cd{};
  cd.mode = static_cast<uint32_t>(spec->mode);
  for (int c : order) {
    const HaloCompositeClass& k = spec->classes[c];
    if (!(any_solo ? k.solo != 0 : k.visible != 0)) continue;
    cd.lane[cd.n_active] = static_cast<uint32_t>(c);
    for (int j = 0; j < 3; j++) cd.color[cd.n_active][j] = k.color[j];
    cd.n_active++;
  }
  auto deliver_black = [&]() -> int {   // "nothing visible -> all-black, still a valid composite" (:201-206)
    if (linear_rgb_out) std::memset(linear_rgb_out, 0, n3 * sizeof(float));
    if (srgb_out) std::memset(srgb_out, 0, n3);
    if (participating_p99_y) *participating_p99_y = 0.0f;
    *produced = 1;
    return HALO_OK;
  };
  if (cd.n_active == 0) return deliver_black();
  // ComputeParticipatingP99Y (:138-163) as a radix select: 11 + 11 + 10 bits of the float pattern, one histogram pass each
  HIPCHK(b, b->comp_hist.reserve(2048));
  std::vector<uint32_t> hist(2048);
  const int blocks = b->cu_count * 8;
  float p99 = 0.0f;
  {
    uint32_t prefix = 0u;
    uint64_t rank = 0;   // 0-based rank wanted among the values that share `prefix`
    const uint32_t shifts[3] = {21u, 10u, 0u}, widths[3] = {11u, 11u, 10u};
    bool empty = false;
    for (int pass = 0; pass < 3 && !empty; pass++) {
      const uint32_t nb = 1u << widths[pass];
      HIPCHK(b, hipMemsetAsync(b->comp_hist.ptr, 0, nb * sizeof(uint32_t), b->stream));
      hipError_t e = launch_lane_hist(b->lanes.ptr, npix, cd, shifts[pass], widths[pass], prefix, b->comp_hist.ptr, blocks, b->stream);
      if (e != hipSuccess) return hip_fail(b, e, "halo_lane_hist_kernel launch");
      HIPCHK(b, hipMemcpyAsync(hist.data(), b->comp_hist.ptr, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
      HIPCHK(b, hipStreamSynchronize(b->stream));
      if (pass == 0) {
        uint64_t count = 0;
        for (uint32_t i = 0; i < nb; i++) count += hist[i];
        if (count == 0) {
          empty = true;
          break;
        }
        // idx = size * 0.99f in float, clamped (:156-159)
        uint64_t idx = static_cast<uint64_t>(static_cast<float>(count) * 0.99f);
        if (idx >= count) idx = count - 1;
        rank = idx;
      }
      uint32_t bin = 0;
      for (; bin < nb; bin++) {
        if (rank < hist[bin]) break;
        rank -= hist[bin];
      }
      if (bin >= nb) return fail(b, HALO_FATAL, "consumer_composite: lanes changed under the P99 select");
      prefix = (prefix << widths[pass]) | bin;
    }
    if (!empty) std::memcpy(&p99, &prefix, sizeof(float));
  }
  // ParticipatingExposureScale (render.cpp:120-135), in float like the reference
  const float snapshot_intensity = static_cast<float>(b->total_intensity);
  float A = 0.0f;
  if (p99 > 0.0f && snapshot_intensity > 0.0f) {
    const float target_srgb = 135.0f / 255.0f;
    const float target_linear = target_srgb <= 0.04045f ? target_srgb / 12.92f : std::pow((target_srgb + 0.055f) / 1.055f, 2.4f);
    A = spec->intensity_factor * target_linear / p99;
  }
  if (participating_p99_y) *participating_p99_y = p99;
  if (!(A > 0.0f)) {   // :209-214: P99 published, no composite; the linear buffer was already assigned zeros (:189)
    if (linear_rgb_out) std::memset(linear_rgb_out, 0, n3 * sizeof(float));
    return HALO_OK;
  }
  cd.a = A;
  cd.s = A * spec->display_exposure_scale;
  cd.display = spec->display_exposure_scale;
  if (linear_rgb_out) HIPCHK(b, b->comp_rgb.reserve(n3));
  if (srgb_out) HIPCHK(b, b->cons_rgb.reserve(n3));
  if (linear_rgb_out || srgb_out) {
    hipError_t e = launch_composite(b->lanes.ptr, npix, cd, linear_rgb_out ? b->comp_rgb.ptr : nullptr, srgb_out ? b->cons_rgb.ptr : nullptr, blocks, b->stream);
    if (e != hipSuccess) return hip_fail(b, e, "halo_composite_kernel launch");
    if (linear_rgb_out) HIPCHK(b, hipMemcpyAsync(linear_rgb_out, b->comp_rgb.ptr, n3 * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    if (srgb_out) HIPCHK(b, hipMemcpyAsync(srgb_out, b->cons_rgb.ptr, n3, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(b, hipStreamSynchronize(b->stream));
  }
  *produced = 1;
  return HALO_OK;
}

// ---- host-side pieces, exported for parity tests -------------------------------------------------------
int halo_host_prism_geometry(float h, const float dist[6], HaloGeomTables* out) {
  if (!out || !dist) return HALO_FATAL;
  host::BuildPrism(h, dist, *out);
  return HALO_OK;
}
int halo_host_pyramid_geometry(float wu, float wl, float h1, float h2, float h3, const float dist[6], HaloGeomTables* out) {
  if (!out || !dist) return HALO_FATAL;
  return host::BuildPyramid(wu, wl, h1, h2, h3, dist, *out) ? HALO_OK : HALO_UNAVAILABLE;
}
int halo_host_shape_scalars(const HaloCrystal* crystal, uint32_t seed, uint64_t shape_index, int via_plan, float out9[9]) {
  if (!crystal || !out9) return HALO_FATAL;
  const geom::CrystalRecipe rc = host::MakeRecipe(*crystal);
  if (via_plan) {
    for (int q = 0; q < 9; q++) out9[q] = geom::DrawShapeScalarOne(seed, rc, shape_index, q);
  } else {
    geom::DrawShapeScalars(seed, rc, shape_index, out9);
  }
  return HALO_OK;
}
int halo_host_build_lat_lut(const HaloDist* lat, float* theta, float* cdf, float* flip) {
  if (!lat || !theta || !cdf || !flip) return HALO_FATAL;
  host::LatLut l = host::BuildLatLut(*lat);
  std::copy(l.theta.begin(), l.theta.end(), theta);
  std::copy(l.cdf.begin(), l.cdf.end(), cdf);
  std::copy(l.flip.begin(), l.flip.end(), flip);
  return HALO_OK;
}
int halo_host_build_proj_params(const HaloRender* render, void* out76) {
  if (!render || !out76) return HALO_FATAL;
  ProjDev p = host::BuildProj(*render);
  static_assert(sizeof(ProjDev) == 76, "ProjDev mirrors lm_proj::ProjParams");
  std::memcpy(out76, &p, sizeof(p));
  return HALO_OK;
}
int halo_host_partition(const float* prop, int n, uint64_t ray_num, double* carry, uint64_t* out) {
  if (!prop || !carry || !out || n < 0) return HALO_FATAL;
  std::vector<uint64_t> v = host::Partition(prop, n, ray_num, carry);
  std::copy(v.begin(), v.end(), out);
  return HALO_OK;
}
int halo_host_reduce_raypath(const uint8_t* rp, int32_t n, int32_t symmetry, int32_t sigma_a, int32_t d_applicable, uint8_t* out) {
  if (!rp || !out || n < 0 || n > HALO_MAX_HITS) return HALO_FATAL;
  std::vector<uint8_t> v = host::ReduceRaypath(std::vector<uint8_t>(rp, rp + n), static_cast<uint8_t>(symmetry), sigma_a, d_applicable != 0);
  std::copy(v.begin(), v.end(), out);
  return HALO_OK;
}
int halo_host_filter_fast_check(const HaloFilter* f, const HaloAxis* axis, const uint8_t* path, int32_t n, const float dir[3], int32_t crystal_id, int32_t* pass) {
  if (!f || !axis || !pass || n < 0 || n > 16 || (n && !path) || !dir) return HALO_FATAL;
  // the tables of the (filter, axis) asked about last are kept: the tests sweep thousands of paths per filter
  struct Cache {
    HaloFilter f;
    HaloAxis axis;
    int32_t crystal_id = 0;
    bool valid = false, fits = false;
    FastTables tables;
  };
  static thread_local std::unique_ptr<Cache> cache;
  if (!cache) cache.reset(new Cache());
  if (!cache->valid || std::memcmp(&cache->f, f, sizeof(HaloFilter)) != 0 || std::memcmp(&cache->axis, axis, sizeof(HaloAxis)) != 0 || cache->crystal_id != crystal_id) {
    std::memset(&cache->tables, 0, sizeof(FastTables));
    cache->f = *f;
    cache->axis = *axis;
    cache->crystal_id = crystal_id;
    cache->fits = host::BuildFastTables(f, nullptr, *axis, static_cast<uint32_t>(crystal_id), cache->tables);
    cache->valid = true;
  }
  if (!cache->fits) return HALO_FATAL;
  *pass = host::FastFilterCheck(cache->tables, path, static_cast<uint32_t>(n), dir, static_cast<uint32_t>(crystal_id)) ? 1 : 0;
  return HALO_OK;
}
int halo_host_color_fast_mask(const HaloColorSet* colors, const HaloAxis* axis, const uint8_t* path, int32_t n, const float dir[3], int32_t crystal_id, uint64_t carried,
                              uint64_t* mask) {
  if (!colors || !axis || !mask || n < 0 || n > 16 || (n && !path) || !dir) return HALO_FATAL;
  std::unique_ptr<FastTables> tables(new FastTables());
  std::memset(tables.get(), 0, sizeof(FastTables));
  if (!host::BuildFastTables(nullptr, colors, *axis, static_cast<uint32_t>(crystal_id), *tables)) return HALO_FATAL;
  *mask = host::FastColorMask(*tables, carried, path, static_cast<uint32_t>(n), dir, static_cast<uint32_t>(crystal_id));
  return HALO_OK;
}
double halo_host_refractive_index(double wl) { return host::IceRefractiveIndex(wl); }
float halo_host_illuminant_spd(int illuminant, float wl) { return host::IlluminantSpd(illuminant, wl); }
int halo_host_wl_pool(const HaloWl* wl, float* entries5, int cap) {
  if (!wl || !entries5 || cap < 0) return 0;
  const std::vector<WlEntryDev> pool = host::BuildWlPool(*wl);
  for (size_t m = 0; m < pool.size() && static_cast<int>(m) < cap; m++) {
    float* e = entries5 + 5 * m;
    e[0] = pool[m].n_idx;
    e[1] = pool[m].spd_weight;
    e[2] = pool[m].cmf_x;
    e[3] = pool[m].cmf_y;
    e[4] = pool[m].cmf_z;
  }
  return static_cast<int>(pool.size());
}

// sizes of the boundary structs, for the Python layout check
uint64_t halo_abi_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(HaloScene);
    case 1: return sizeof(HaloRender);
    case 2: return sizeof(HaloWl);
    case 3: return sizeof(HaloExitRecord);
    case 4: return sizeof(HaloGeomTables);
    case 5: return sizeof(HaloLayerStats);
    case 6: return sizeof(HaloEntry);
    case 7: return sizeof(HaloColorSet);
    case 8: return sizeof(HaloColorClass);
    case 9: return sizeof(HaloFilter);
    case 10: return sizeof(HaloRouteInfo);
    case 11: return sizeof(HaloComposite);
    default: return 0;
  }
}

}  // extern "C"
