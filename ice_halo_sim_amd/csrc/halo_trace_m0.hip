// halo_trace_m0.hip — the kModePlain instantiations of halo_trace_kernel (see halo_trace.inl).
#include "halo_trace.inl"

namespace halo {
hipError_t launch_trace_m0(const DispatchParams& P, int blocks, hipStream_t stream, int geom, bool mono) {
  return launch_mode<kModePlain>(P, blocks, stream, geom, mono);
}
}  // namespace halo

#ifdef HALO_PROBE
// probe builds only (tools/phase_probe.py): read and optionally zero the per-phase wave-cycle sums of the MODE 0 kernels
extern "C" int halo_probe_dump(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(halo::g_halo_probe), 16 * sizeof(unsigned long long)) != hipSuccess) return 2;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(halo::g_halo_probe), z, sizeof(z)) != hipSuccess) return 2;
  }
  return 0;
}
#endif
