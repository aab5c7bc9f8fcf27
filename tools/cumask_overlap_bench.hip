// Can a memory-bound pass hide behind a VALU-bound kernel when the two run on CU-masked streams?
// A: fp32 FMA chains (VALU-bound, no memory); B: a streaming read-modify-write over `bytes` (HBM-bound).
// Prints: A alone (all CUs), B alone (all CUs), A then B on one stream, and for K = 8..64 aux CUs: A on the other CUs and B on K CUs, together.
//   hipcc --offload-arch=gfx950 -O3 cumask_overlap_bench.hip -o cumask_overlap_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) valu_kernel(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; i++) {
    a = fmaf(a, b, c); c = fmaf(c, b, d); d = fmaf(d, b, a); b = fmaf(b, 0.99999f, 1e-6f);
  }
  if (a + c + d + b == 12345.0f) out[blockIdx.x] = a;
}
__global__ void __launch_bounds__(1024) stream_kernel(float4* buf, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = buf[i];
    v.x += 1.0f;
    buf[i] = v;
  }
}
static float timed(hipStream_t s0, hipStream_t s1, float* out, float4* buf, size_t n, int iters, int blocksA, int blocksB, bool runA, bool runB) {
  hipEvent_t e0, e1, ea;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&ea);
  hipDeviceSynchronize();
  hipEventRecord(e0, s0);
  if (s1 != s0) hipStreamWaitEvent(s1, e0, 0);
  if (runA) hipLaunchKernelGGL(valu_kernel, dim3(blocksA), dim3(256), 0, s0, out, iters);
  if (runB) hipLaunchKernelGGL(stream_kernel, dim3(blocksB), dim3(1024), 0, s1, buf, n);
  if (s1 != s0) { hipEventRecord(ea, s1); hipStreamWaitEvent(s0, ea, 0); }
  hipEventRecord(e1, s0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  const size_t bytes = 900ull << 20;   // read + write = 1.8 GB of traffic
  float4* buf; float* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(buf, 0, bytes));
  const size_t n = bytes / 16;
  hipStream_t full; CK(hipStreamCreateWithFlags(&full, hipStreamNonBlocking));
  const int iters = 60000, blocksA = 256 * 16;
  for (int r = 0; r < 2; r++) {
    printf("A alone %.3f ms | B alone %.3f ms | serial %.3f ms\n", timed(full, full, out, buf, n, iters, blocksA, 1024, true, false),
           timed(full, full, out, buf, n, iters, blocksA, 1024, false, true), timed(full, full, out, buf, n, iters, blocksA, 1024, true, true));
  }
  for (int K : {8, 16, 24, 32, 48, 64}) {
    uint32_t ma[8], mb[8];
    for (int w = 0; w < 8; w++) { ma[w] = 0xFFFFFFFFu; mb[w] = 0u; }
    for (int i = 0; i < K; i++) { ma[i >> 5] &= ~(1u << (i & 31)); mb[i >> 5] |= 1u << (i & 31); }
    hipStream_t sa, sb;
    CK(hipExtStreamCreateWithCUMask(&sa, 8, ma)); CK(hipExtStreamCreateWithCUMask(&sb, 8, mb));
    float a = 0, b = 0, ab = 0;
    for (int r = 0; r < 2; r++) {
      a = timed(sa, sa, out, buf, n, iters, blocksA, K * 2, true, false);
      b = timed(sb, sb, out, buf, n, iters, blocksA, K * 2, false, true);
      ab = timed(sa, sb, out, buf, n, iters, blocksA, K * 2, true, true);
    }
    printf("K=%2d aux CUs: A on the rest %.3f ms | B on K %.3f ms (%.2f TB/s) | together %.3f ms\n", K, a, b, 2.0 * bytes / b * 1e-9, ab);
    hipStreamDestroy(sa); hipStreamDestroy(sb);
  }
  return 0;
}
