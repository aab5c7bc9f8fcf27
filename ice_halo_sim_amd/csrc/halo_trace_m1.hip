// halo_trace_m1.hip — the kModeFilter instantiations of halo_trace_kernel (see halo_trace.inl): emit-gate filter in its fast form.
#include "halo_trace.inl"

namespace halo {
hipError_t launch_trace_m1(const DispatchParams& P, int blocks, hipStream_t stream, int geom, bool mono) {
  return launch_mode<kModeFilter>(P, blocks, stream, geom, mono);
}
}  // namespace halo
