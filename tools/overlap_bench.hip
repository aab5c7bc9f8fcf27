// Micro-benchmark: do scattered fp32 atomics overlap with VALU work on gfx950, and what staging helps?
//   MODE 0: ALU only   1: atomics only   2: ALU + 1 atomic per iteration (interleaved)
//   MODE 3: ALU + hits staged per wave in LDS, flushed every 16 iterations as back-to-back atomics
//   MODE 4: ALU + coalesced 8-byte hit-list stores (wave ballot slot reservation)
//   MODE 5: ALU + LDS atomic into a 16K-float LDS tile (no global)
//   MODE 6: ALU + scattered plain dword store (not an accumulate; speed reference)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t pcg(uint32_t x){ x = x*747796405u+2891336453u; x=((x>>((x>>28u)+4u))^x)*277803737u; return (x>>22u)^x; }

template<int MODE, int K>
__global__ void __launch_bounds__(256, 4) k(float* buf, uint32_t npix, uint32_t iters, uint2* list, uint32_t* list_cnt, float* sink){
  __shared__ float tile[16384];
  __shared__ uint32_t stage_pix[4][16 * 64];
  const uint32_t tid = blockIdx.x*blockDim.x+threadIdx.x;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (MODE == 5) { for (uint32_t i = threadIdx.x; i < 16384; i += 256) tile[i] = 0.f; __syncthreads(); }
  float a = tid * 1e-9f, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (uint32_t i = 0; i < iters; i++) {
    if (MODE != 1) {
#pragma unroll
      for (int j = 0; j < K; j++) { a = fmaf(a, b, c); c = fmaf(c, b, d); d = fmaf(d, b, a); b = fmaf(b, 0.99999f, 1e-6f); }
    }
    const uint32_t pix = pcg(tid*977u+i*0x9E3779B9u) % npix;
    const float v = 1.0f + a * 1e-30f;
    if (MODE == 1 || MODE == 2) unsafeAtomicAdd(buf + pix, v);
    if (MODE == 3) {
      stage_pix[wave][(i & 15) * 64 + lane] = pix;
      if ((i & 15) == 15) {
#pragma unroll
        for (int q = 0; q < 16; q++) unsafeAtomicAdd(buf + stage_pix[wave][q * 64 + lane], v);
      }
    }
    if (MODE == 4) {
      uint32_t base = 0; if (lane == 0) base = atomicAdd(list_cnt, 64u);
      base = __shfl(base, 0);
      list[base + lane] = make_uint2(pix, __float_as_uint(v));
    }
    if (MODE == 5) unsafeAtomicAdd(&tile[pix & 16383u], v);
    if (MODE == 6) buf[pix] = v;
  }
  if (MODE == 5) { __syncthreads(); float s = 0; for (uint32_t i = threadIdx.x; i < 16384; i += 256) s += tile[i]; a += s; }
  if (a + b + c + d == 12345.678f) sink[0] = a;
}
template<int MODE, int K> void run(const char* name, float* buf, uint32_t npix, uint2* list, uint32_t* cnt, float* sink){
  const int blocks=2048, threads=256; const uint32_t per=256;
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemset(cnt,0,4);
  hipLaunchKernelGGL((k<MODE,K>),dim3(blocks),dim3(threads),0,0,buf,npix,16u,list,cnt,sink); hipDeviceSynchronize();
  hipMemset(cnt,0,4);
  hipEventRecord(e0); hipLaunchKernelGGL((k<MODE,K>),dim3(blocks),dim3(threads),0,0,buf,npix,per,list,cnt,sink); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  double ops = (double)blocks*threads*per;
  printf("%-44s K=%3d  %8.3f ms  %7.2f G iter/s\n",name,K,ms,ops/ms/1e6);
}
template<int K> void suite(float* buf, uint32_t npix, uint2* list, uint32_t* cnt, float* sink){
  run<0,K>("ALU only",buf,npix,list,cnt,sink);
  run<1,K>("atomics only",buf,npix,list,cnt,sink);
  run<2,K>("ALU + atomic interleaved",buf,npix,list,cnt,sink);
  run<3,K>("ALU + LDS-staged burst of 16 atomics",buf,npix,list,cnt,sink);
  run<4,K>("ALU + coalesced hit-list store",buf,npix,list,cnt,sink);
  run<5,K>("ALU + LDS atomic",buf,npix,list,cnt,sink);
  run<6,K>("ALU + scattered plain store",buf,npix,list,cnt,sink);
}
int main(){
  const uint32_t npix = 1920u*1080u; float* buf; hipMalloc(&buf,(size_t)npix*4); hipMemset(buf,0,(size_t)npix*4);
  const size_t nlist = (size_t)2048*256*256 + 64; uint2* list; hipMalloc(&list, nlist*8);
  uint32_t* cnt; hipMalloc(&cnt,4); float* sink; hipMalloc(&sink,4);
  suite<8>(buf,npix,list,cnt,sink);
  suite<32>(buf,npix,list,cnt,sink);
  suite<128>(buf,npix,list,cnt,sink);
  return 0;
}
