// halo_shapegen.hip — device crystal generator: one thread samples one crystal instance and writes
// its kernel tables.  The geometry code is halo_geom.h, the same functions the host uses, compiled here with
// -ffp-contract=off so that identical scalars give bit-identical tables on both sides.
//
// Replaces, for stochastic crystals, the host loop MakeCrystal (simulator.cpp:448) → closed-form geometry
// (geo3d_closedform.cpp:124-302,1318-1407) → PopulateFromCfGeom (crystal.cpp:304-347) and the upload of its result;
// the reference's device counterpart is the per-ray geometry stage of its CUDA backend (cuda_trace_backend.cu:2577-2800).
// geom_clock consecutive rays share shape k (legacy semantics, simulator.cpp:1244-1275), so the pool has n/geom_clock
// entries and the trace kernel reads shape tid/geom_clock.
#include <hip/hip_runtime.h>

#include "halo_geom.h"

namespace halo {

constexpr int kGenBlock = 64;

// S = ShapeDev (any crystal; 4.1 KB records) or ShapePrism (prism pools; 1.4 KB records).  Every record gets its counts
// written (an invalid draw leaves them 0 = empty crystal) and rows beyond the counts are never read, so the pool needs no
// clearing.
#ifndef HALO_GEN_WAVES
#define HALO_GEN_WAVES 2
#endif
template <class S>
// (waves per SIMD the general builder is compiled for: measured below; the pyramid feasibility scans keep the 20 fp64 planes
// in registers)
__global__ void __launch_bounds__(kGenBlock, (sizeof(S) == sizeof(ShapeDev) ? HALO_GEN_WAVES : 1)) halo_shapegen_kernel(S* __restrict__ pool, uint32_t n, uint32_t seed, const geom::CrystalRecipe rc,
                                                                   uint64_t first_index) {
  const uint32_t k = blockIdx.x * kGenBlock + threadIdx.x;
  if (k >= n) return;
  geom::MakeShapeDev(seed, rc, first_index + k, pool[k]);
}

hipError_t launch_shapegen(void* pool, bool prism_records, uint32_t n, uint32_t seed, const geom::CrystalRecipe& rc, uint64_t first_index,
                           hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const dim3 grid((n + kGenBlock - 1) / kGenBlock), block(kGenBlock);
  if (prism_records) hipLaunchKernelGGL(halo_shapegen_kernel<ShapePrism>, grid, block, 0, stream, static_cast<ShapePrism*>(pool), n, seed, rc, first_index);
  else hipLaunchKernelGGL(halo_shapegen_kernel<ShapeDev>, grid, block, 0, stream, static_cast<ShapeDev*>(pool), n, seed, rc, first_index);
  return hipGetLastError();
}

}  // namespace halo
