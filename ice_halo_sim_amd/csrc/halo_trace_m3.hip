// halo_trace_m3.hip — the kModeGeneric instantiations of halo_trace_kernel (see halo_trace.inl): filters and raypath colour with
// paths of up to 64 faces and the symmetry reduction on the device; serves what the fast kernels (m1, m4) do not take.
#include "halo_trace.inl"

namespace halo {
hipError_t launch_trace_m3(const DispatchParams& P, int blocks, hipStream_t stream, int geom, bool mono) {
  return launch_mode<kModeGeneric>(P, blocks, stream, geom, mono);
}
}  // namespace halo
