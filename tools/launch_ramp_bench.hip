// Where does a SMALL launch's time go on MI355X?  N workgroups of 256 threads, each spinning for a fixed number of dependent FMAs (a stand-in
// for one pass of the trace loop), optionally touching `lds_bytes` of LDS and reading a 24 KB table that every workgroup shares (the dispatch
// slot).  Every workgroup stamps the constant 100 MHz clock (s_memrealtime) when it starts and when it ends, so the host can tell
//   * how long the first workgroup waited after the launch, and how the starts are spread (dispatch ramp),
//   * how long a workgroup runs (its own latency),
//   * what is left between the last workgroup's end and the kernel's end as the events see it.
// hipcc --offload-arch=gfx950 -O3 tools/launch_ramp_bench.hip -o tools/launch_ramp_bench.bin
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                 \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      std::printf("%s: %s\n", #x, hipGetErrorString(e));                       \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

template <int LDS>
__global__ void __launch_bounds__(256) spin_kernel(unsigned long long* stamps, const float* table, float* sink, int iters, int read_table) {
  __shared__ float lds[LDS / 4 > 0 ? LDS / 4 : 1];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float a = static_cast<float>(threadIdx.x) * 1e-3f, b = 1.0001f;
  if (read_table)
    for (int i = threadIdx.x; i < 6144; i += 256) lds[i % (LDS / 4 > 0 ? LDS / 4 : 1)] = table[i];
  __syncthreads();
  for (int i = 0; i < iters; i++) a = __builtin_fmaf(a, b, 1e-7f);
  if (LDS > 0) a += lds[threadIdx.x % (LDS / 4 > 0 ? LDS / 4 : 1)];
  if (a == 12345.678f) sink[0] = a;
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = t1;
  }
}

// the shader clock a small launch actually runs at: s_memtime ticks (clock64) per 100 MHz tick (wall_clock64) around a dependent FMA chain
__global__ void __launch_bounds__(256) clock_kernel(unsigned long long* out, float* sink, int iters) {
  float a = static_cast<float>(threadIdx.x) * 1e-3f, b = 1.0001f;
  const unsigned long long w0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) a = __builtin_fmaf(a, b, 1e-7f);
  if (a == 12345.678f) sink[0] = a;
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = w1 - w0;
    out[1] = c1 - c0;
  }
}
// instruction cache: ~32 KB of straight-line code (4096 dependent v_fma_f32 with a literal: 8 bytes each... the compiler picks the encoding)
// run twice inside ONE launch; pass 0 fetches the code, pass 1 finds it cached.  Stamps per pass from workgroup 0.
#define FMA8(a, b) a = __builtin_fmaf(a, b, 1e-7f); a = __builtin_fmaf(a, b, 2e-7f); a = __builtin_fmaf(a, b, 3e-7f); a = __builtin_fmaf(a, b, 4e-7f); \
                   a = __builtin_fmaf(a, b, 5e-7f); a = __builtin_fmaf(a, b, 6e-7f); a = __builtin_fmaf(a, b, 7e-7f); a = __builtin_fmaf(a, b, 8e-7f);
#define FMA64(a, b) FMA8(a, b) FMA8(a, b) FMA8(a, b) FMA8(a, b) FMA8(a, b) FMA8(a, b) FMA8(a, b) FMA8(a, b)
#define FMA512(a, b) FMA64(a, b) FMA64(a, b) FMA64(a, b) FMA64(a, b) FMA64(a, b) FMA64(a, b) FMA64(a, b) FMA64(a, b)
__global__ void __launch_bounds__(256) bigcode_kernel(unsigned long long* out, float* sink, int passes) {
  float a = static_cast<float>(threadIdx.x) * 1e-3f, b = 1.0001f;
  for (int p = 0; p < passes; p++) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    FMA512(a, b) FMA512(a, b) FMA512(a, b) FMA512(a, b) FMA512(a, b) FMA512(a, b) FMA512(a, b) FMA512(a, b)
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[p] = c1 - c0;
    __builtin_amdgcn_sched_barrier(0);
  }
  if (a == 12345.678f) sink[0] = a;
}
__global__ void __launch_bounds__(256) burn_kernel(float* sink, int iters) {
  float a = static_cast<float>(threadIdx.x) * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; i++) {
    a = __builtin_fmaf(a, b, 1e-7f);
    c = __builtin_fmaf(c, b, 1e-7f);
    d = __builtin_fmaf(d, b, 1e-7f);
  }
  if (a + c + d == 12345.678f) sink[0] = a;
}

template <int LDS>
static void run(int wgs, int iters, int read_table, int reps, hipStream_t st, unsigned long long* d_st, const float* d_tab, float* d_sink) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  std::vector<unsigned long long> h(2 * wgs);
  double ev_us = 0, first = 0, spread = 0, run_wg = 0, span = 0;
  for (int r = 0; r < reps + 2; r++) {
    CHK(hipEventRecord(e0, st));
    spin_kernel<LDS><<<wgs, 256, 0, st>>>(d_st, d_tab, d_sink, iters, read_table);
    CHK(hipEventRecord(e1, st));
    CHK(hipStreamSynchronize(st));
    if (r < 2) continue;
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long s_min = ~0ull, s_max = 0, e_max = 0;
    double rw = 0;
    for (int i = 0; i < wgs; i++) {
      s_min = std::min(s_min, h[2 * i]);
      s_max = std::max(s_max, h[2 * i]);
      e_max = std::max(e_max, h[2 * i + 1]);
      rw += static_cast<double>(h[2 * i + 1] - h[2 * i]);
    }
    ev_us += ms * 1e3;
    spread += (s_max - s_min) * 0.01;
    run_wg += rw / wgs * 0.01;
    span += (e_max - s_min) * 0.01;
  }
  std::printf("lds %6d B  wgs %5d  iters %6d  table %d : events %7.1f us | first-start..last-end %7.1f us | starts spread over %6.1f us | a workgroup runs %6.1f us\n", LDS, wgs,
              iters, read_table, ev_us / reps, span / reps, spread / reps, run_wg / reps);
  (void)first;
}

int main() {
  hipStream_t st;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned long long* d_st;
  float *d_tab, *d_sink;
  CHK(hipMalloc(&d_st, 2 * 65536 * 8));
  CHK(hipMalloc(&d_tab, 6144 * 4));
  CHK(hipMalloc(&d_sink, 4));
  CHK(hipMemset(d_tab, 0, 6144 * 4));
  for (int iters : {0, 2000, 8000})
    for (int wgs : {64, 128, 256, 512, 1024, 4096}) run<0>(wgs, iters, 0, 20, st, d_st, d_tab, d_sink);
  for (int wgs : {128, 256, 512, 1024}) run<31744>(wgs, 2000, 0, 20, st, d_st, d_tab, d_sink);
  for (int wgs : {128, 256, 512, 1024}) run<31744>(wgs, 2000, 1, 20, st, d_st, d_tab, d_sink);
  for (int wgs : {256, 1024}) run<65536>(wgs, 2000, 1, 20, st, d_st, d_tab, d_sink);
  // clocks: 256 workgroups x 8000 dependent FMAs — sporadic (10 ms apart), back to back, and right behind 50 ms of a full-chip burn
  auto clock_run = [&](const char* what, int gap_us, int burn) {
    unsigned long long h[2];
    double w = 0, c = 0;
    const int reps = 20;
    for (int r = 0; r < reps; r++) {
      if (burn) burn_kernel<<<256 * 32, 256, 0, st>>>(d_sink, burn);
      clock_kernel<<<256, 256, 0, st>>>(d_st, d_sink, 8000);
      CHK(hipStreamSynchronize(st));
      CHK(hipMemcpy(h, d_st, 16, hipMemcpyDeviceToHost));
      w += h[0] * 0.01;
      c += static_cast<double>(h[1]);
      if (gap_us) usleep(gap_us);
    }
    std::printf("%-40s: 8000 dependent FMAs take %7.1f us, s_memtime ticks %9.0f -> %6.1f ticks per us, %.2f ns per FMA\n", what, w / reps, c / reps, c / w, w / reps * 1e3 / 8000);
  };
  for (int launch = 0; launch < 3; launch++) {
    unsigned long long h[4] = {};
    bigcode_kernel<<<256, 256, 0, st>>>(d_st, d_sink, 3);
    CHK(hipStreamSynchronize(st));
    CHK(hipMemcpy(h, d_st, 24, hipMemcpyDeviceToHost));
    std::printf("4096 straight-line FMAs (32 KB of code), launch %d: pass 0 %llu ticks, pass 1 %llu, pass 2 %llu  (s_memtime ticks, 2.4 per ns)\n", launch, h[0], h[1], h[2]);
  }
  clock_run("sporadic (10 ms between launches)", 10000, 0);
  clock_run("back to back", 0, 0);
  clock_run("behind a 5 ms full-chip burn", 0, 200000);
  clock_run("behind a 50 ms full-chip burn", 0, 2000000);
  return 0;
}
