// hip_trace_backend.hpp — C++ adapter: the C ABI of include/halo_trace.h presented with the method set, state
// machine and error behaviour of `lumice::TraceBackend` (reference src/core/backend/trace_backend.hpp:367-641).
//
// Header-only, depends on nothing but <stdexcept>/<vector> and halo_trace.h, so it compiles in this repo (see
// tests/test_cpp_adapter.py) AND inside a Lumice checkout.  There, INTEGRATION.md shows the 30-line glue that
// derives from the real `lumice::TraceBackend`, converts SceneConfig/RenderConfig → HaloScene/HaloRender and
// forwards to this class; nothing here includes reference headers.
#ifndef HALO_HIP_TRACE_BACKEND_HPP_
#define HALO_HIP_TRACE_BACKEND_HPP_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/halo_trace.h"

namespace halo {

// Mirrors lumice::BackendUnavailableError (trace_backend.hpp:140-158): the ONE recoverable error; the caller drops
// the backend for the rest of Run() and re-runs the wavelength on the legacy CPU path (simulator.cpp:1049-1062).
class BackendUnavailableError : public std::runtime_error {
 public:
  using std::runtime_error::runtime_error;
};

struct XyzImageData {  // trace_backend.hpp:356-360
  float* data = nullptr;
  int width = 0;
  int height = 0;
};

struct LayerHandle {  // trace_backend.hpp:309-333 (value type here: the device state lives in the backend)
  HaloLayerStats stats{};
  size_t ContinuationCount() const { return static_cast<size_t>(stats.continuation_count); }
};

class HipTraceBackend {
 public:
  // CreateBackend(BackendKind::kHip): simulator.cpp:854-919.  seed = effective_seed_ (non-zero).
  explicit HipTraceBackend(int device_ordinal = 0, uint32_t seed = 1) {
    int rc = halo_create(device_ordinal, seed, &h_);
    if (rc == HALO_UNAVAILABLE) throw BackendUnavailableError("halo: no usable gfx950 device");
    if (rc != HALO_OK) throw std::runtime_error("halo_create failed");
  }
  ~HipTraceBackend() { halo_destroy(h_); }
  HipTraceBackend(const HipTraceBackend&) = delete;
  HipTraceBackend& operator=(const HipTraceBackend&) = delete;

  void SetOption(const char* key, int64_t value) { Check(halo_set_option(h_, key, value)); }

  // --- the seam ---------------------------------------------------------------------------------------------
  void BeginSession(const HaloScene& scene, const HaloRender& render, const HaloWl& wl, size_t ray_num = 0) {
    Check(halo_begin(h_, &scene, &render, &wl, ray_num));
    width_ = render.width;
    height_ = render.height;
  }
  // First call: host mode (count roots generated on device, or injected golden rays); later calls consume the
  // continuation returned by Recombine (trace_backend.hpp:380-389).
  LayerHandle TraceLayer(size_t count, const HaloHostRays* host = nullptr) {
    LayerHandle lh;
    Check(halo_trace_layer(h_, count, host, &lh.stats));
    return lh;
  }
  size_t Recombine(const LayerHandle&, bool shuffle = true) {
    uint64_t n = 0;
    Check(halo_recombine(h_, shuffle ? 1 : 0, &n));
    return static_cast<size_t>(n);
  }
  // TraceBackend::DrainExits (trace_backend.hpp:430-448): every record emitted since the previous drain, no clamp.  A
  // device-accumulating backend has none in production; with the "capture_exits" option (tests) the captured records come
  // back.  max_records > 0 drains in pieces: at most that many now, the rest stays pending for the next call.
  size_t DrainExits(std::vector<HaloExitRecord>& out, size_t max_records = 0) {
    uint64_t n = 0;
    Check(halo_drain_exits(h_, nullptr, 0, &n));  // pending count, nothing consumed
    if (max_records && n > max_records) n = max_records;
    out.resize(static_cast<size_t>(n));
    if (n) Check(halo_drain_exits(h_, out.data(), n, &n));
    out.resize(static_cast<size_t>(n));
    return out.size();
  }
  bool SupportsDeviceXyzAccum() const { return true; }
  bool SupportsThirdClockDrain() const { return true; }  // accumulator persists across sessions (cu:4801-4808)
  uint32_t WlPoolSize() const { return 64; }             // illuminant mode: per-ray pool (wl_pool.hpp:41)
  // Adds into landed_weight, copies W*H*3 floats, zeroes the device accumulator (trace_backend.hpp:461-469).
  void ReadbackXyzAccum(XyzImageData& xyz, float& landed_weight) {
    Check(halo_readback_xyz(h_, xyz.data, xyz.width, xyz.height, &landed_weight));
  }
  void EndSession() { Check(halo_end(h_)); }

  // --- beyond the reference's virtuals: tables the reference hands over inside SessionSpec, and the deferred tallies ---
  // physical filters referenced by HaloEntry::filter_id (reference: FilterSpec per scattering setting, filter_spec.hpp)
  void SetFilters(const std::vector<HaloFilter>& filters) {
    Check(halo_set_filters(h_, filters.empty() ? nullptr : filters.data(), static_cast<int32_t>(filters.size())));
  }
  // raypath colour (reference: SessionSpec::raypath_color → ColorGateTable / ColorClassTable)
  void SetColor(const std::vector<HaloColorSet>& sets, const std::vector<HaloColorClass>& classes) {
    Check(halo_set_color(h_, sets.empty() ? nullptr : sets.data(), static_cast<int>(sets.size()), classes.empty() ? nullptr : classes.data(),
                         static_cast<int>(classes.size())));
    class_count_ = classes.size();
  }
  // TraceBackend::ReadbackClassLanes (trace_backend.hpp:471-493): lane c at lane_data[c*W*H + py*W + px]; zeroes the device
  void ReadbackClassLanes(std::vector<float>& lane_data, size_t& class_count) {
    class_count = class_count_;
    lane_data.assign(class_count_ * static_cast<size_t>(width_) * static_cast<size_t>(height_), 0.0f);
    if (class_count_) Check(halo_readback_class_lanes(h_, lane_data.data(), width_, height_, static_cast<int>(class_count_)));
  }
  // multi-GPU drain from a C++ host: one ncclReduce of the accumulator onto `root`, the other ranks drained (halo_reduce_accumulator)
  void ReduceAccumulator(void* nccl_comm, int root, int this_rank) { Check(halo_reduce_accumulator(h_, nccl_comm, root, this_rank)); }
  double TakeLanded() {
    double l = 0.0;
    Check(halo_take_landed(h_, &l));
    return l;
  }
  // option "async" = 1: final-layer TraceLayer only queues; the summed LayerStats of everything traced since the last call
  HaloLayerStats CollectStats() {
    HaloLayerStats st{};
    Check(halo_collect_stats(h_, &st));
    return st;
  }
  // RenderConsumer on the device (server/render.cpp:138-330)
  void ConsumeDeviceFused() { Check(halo_consumer_fold(h_)); }
  // ... and the same for a drained image the caller holds (SimData::xyz_pixel_data_ / xyz_landed_weight_ / lane_pixel_data_ of another rank or backend)
  void ConsumeDeviceFused(const float* xyz, int width, int height, float landed, const float* lanes = nullptr, int class_count = 0) {
    Check(halo_consumer_consume(h_, xyz, width, height, landed, lanes, class_count));
  }
  // CompositeColorClassesLinear + LinearRgbToSrgbU8 (server/component_compositor.cpp:180-303) on the device lanes, which stay; returns what the
  // reference's function returns (false: nothing referenced, or no energy in the participating lanes); either output may be null
  bool CompositeColorClasses(const HaloComposite& spec, float* linear_rgb_out, uint8_t* srgb_out, float* participating_p99_y = nullptr) {
    int32_t produced = 0;
    Check(halo_consumer_composite(h_, &spec, linear_rgb_out, srgb_out, participating_p99_y, &produced));
    return produced != 0;
  }
  // lanes summed elsewhere (other ranks, a saved consumer) become this consumer's lanes; total_intensity < 0 keeps the consumer's own
  void LoadClassLanes(const float* lanes, int width, int height, int class_count, double total_intensity = -1.0) {
    Check(halo_consumer_load_lanes(h_, lanes, width, height, class_count, total_intensity));
  }
  double Snapshot(const HaloDisplay& display, uint8_t* rgb_out, float* xyz_out = nullptr) {
    double total = 0.0;
    Check(halo_consumer_snapshot(h_, &display, rgb_out, xyz_out, &total));
    return total;
  }
  // trace_backend.hpp:587,625 — real counts of the last session's stochastic draws (never stand-ins)
  size_t GetLastBatchStochasticCrystalSampleCount() const {
    uint64_t c = 0, o = 0;
    (void)halo_last_sample_counts(h_, &c, &o);
    return static_cast<size_t>(c);
  }
  size_t GetLastBatchStochasticOrientationSampleCount() const {
    uint64_t c = 0, o = 0;
    (void)halo_last_sample_counts(h_, &c, &o);
    return static_cast<size_t>(o);
  }
  bool IsCompatible(const HaloRender& render) const { return render.width > 0 && render.height > 0; }

 private:
  void Check(int rc) {
    if (rc == HALO_OK) return;
    std::string msg = halo_last_error(h_);
    if (rc == HALO_UNAVAILABLE) throw BackendUnavailableError(msg);
    throw std::runtime_error(msg);
  }
  halo_handle_t h_ = nullptr;
  int width_ = 0, height_ = 0;
  size_t class_count_ = 0;
};

}  // namespace halo

#endif
