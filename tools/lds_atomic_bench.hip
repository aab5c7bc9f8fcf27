// Micro-benchmark: LDS atomic throughput on gfx950 by flavour (u32 / f32, returning or not, random or few addresses).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t pcg(uint32_t x){ x = x*747796405u+2891336453u; x=((x>>((x>>28u)+4u))^x)*277803737u; return (x>>22u)^x; }
template<int MODE>
__global__ void __launch_bounds__(1024) k(float* out, uint32_t iters, uint32_t range){
  __shared__ float acc[16384];
  uint32_t* ai = reinterpret_cast<uint32_t*>(acc);
  for (uint32_t j = threadIdx.x; j < 16384; j += blockDim.x) acc[j] = 0.0f;
  __syncthreads();
  uint32_t s = pcg(blockIdx.x * 1024u + threadIdx.x);
  uint32_t sink = 0;
  for (uint32_t i = 0; i < iters; i++) {
    s = s * 747796405u + 2891336453u;
    const uint32_t a = (s >> 10) & (range - 1u);
    if (MODE == 0) unsafeAtomicAdd(&acc[a], 1.0f);                 // ds_add_f32
    if (MODE == 1) atomicAdd(&ai[a], 1u);                          // ds_add_u32 (no return)
    if (MODE == 2) sink += atomicAdd(&ai[a], 1u);                  // ds_add_rtn_u32
    if (MODE == 3) { float v = acc[a]; acc[a] = v + 1.0f; }         // non-atomic RMW (wrong under conflicts; speed reference)
    if (MODE == 5) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + (a & 8191u), 1ull);   // ds_add_u64
    if (MODE == 6) unsafeAtomicAdd(reinterpret_cast<double*>(acc) + (a & 8191u), 1.0);            // ds_add_f64
    if (MODE == 7) atomicMax(&ai[a], s);                                                        // ds_max_u32
    if (MODE == 4) sink += __float_as_uint(unsafeAtomicAdd(&acc[a], 1.0f));  // ds_add_rtn_f32
  }
  __syncthreads();
  float t = 0.0f;
  for (uint32_t j = threadIdx.x; j < 16384; j += blockDim.x) t += acc[j];
  if (t == 123.456f || sink == 0xFFFFFFFFu) out[blockIdx.x] = t;
}
template<int MODE> void run(const char* name, float* out, uint32_t range, int threads){
  const int blocks = 256 * 2 * 8; const uint32_t iters = 256;
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>,dim3(blocks),dim3(threads),0,0,out,8,range); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(k<MODE>,dim3(blocks),dim3(threads),0,0,out,iters,range); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b);
  double ops = (double)blocks*threads*iters;
  printf("%-28s range %6u threads %4d  %8.3f ms  %8.1f G ops/s\n",name,range,threads,ms,ops/ms/1e6);
}
int main(){
  float* out; hipMalloc(&out, 1<<20);
  for (int threads : {1024}) for (uint32_t range : {16384u, 512u}) {
    run<0>("ds_add_f32", out, range, threads);
    run<1>("ds_add_u32", out, range, threads);
    run<2>("ds_add_rtn_u32", out, range, threads);
    run<4>("ds_add_rtn_f32", out, range, threads);
    run<3>("ds_read+ds_write (no atomic)", out, range, threads);
    run<5>("ds_add_u64", out, range, threads);
    run<6>("ds_add_f64", out, range, threads);
  }
  return 0;
}
