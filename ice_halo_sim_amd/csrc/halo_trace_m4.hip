// halo_trace_m4.hip — the kModeColor instantiations of halo_trace_kernel (see halo_trace.inl): emit-gate filter + raypath colour,
// fast form.
#include "halo_trace.inl"

namespace halo {
hipError_t launch_trace_m4(const DispatchParams& P, int blocks, hipStream_t stream, int geom, bool mono) {
  return launch_mode<kModeColor>(P, blocks, stream, geom, mono);
}
}  // namespace halo
