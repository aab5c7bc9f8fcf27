// Micro-benchmark: what does gfx950 sustain for the accumulation traffic of the trace kernel — fp32 global atomics on
// random slots of an image plane — and what do the alternatives cost (plain 8-byte log stores, returning atomics)?
// Variants: footprint of the plane (MB), active lanes per instruction (the kernel's miss path runs with ~10 of 64), copies
// addressed by blockIdx & 7 (the kernel's privatised planes), some ALU work between atomics.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_rate_bench.hip -o tools/atomic_rate_bench.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t pcg(uint32_t x) {
  x = x * 747796405u + 2891336453u;
  x = ((x >> ((x >> 28) + 4u)) ^ x) * 277803737u;
  return (x >> 22) ^ x;
}

// MODE 0: non-returning fp32 atomic; 1: plain 4-byte store to the slot; 2: 8-byte store to a per-wave contiguous log;
// 3: returning u32 atomic on a per-workgroup counter + 8-byte store (the shard-log scheme); 4: nothing (ALU only)
template <int MODE>
__global__ void __launch_bounds__(256, 5) k(float* plane, uint32_t slot_mask, uint32_t copy_shift, uint32_t iters, uint32_t lane_keep, uint32_t alu, uint2* log,
                                            uint32_t* cnt, float* sink) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  uint32_t s = pcg(t + 1u);
  float acc = 0.0f;
  const uint32_t copy = copy_shift < 31u ? (blockIdx.x & 7u) << copy_shift : 0u;
  const uint32_t wave = t >> 6;
  uint32_t cur = 0u;
  for (uint32_t i = 0; i < iters; i++) {
    s = pcg(s);
    float v = __uint_as_float(0x3f800000u | (s >> 9)) - 1.0f;
    for (uint32_t a = 0; a < alu; a++) v = fmaf(v, 0.999f, 1e-3f);   // stand-in for the trace between two hits
    acc += v;
    const bool on = (pcg(s ^ i) & 63u) < lane_keep;   // a random subset of the lanes takes the miss path
    if (on) {
      const uint32_t slot = (s & slot_mask) + copy;
      if (MODE == 0) {
        unsafeAtomicAdd(plane + slot, v);
      } else if (MODE == 1) {
        plane[slot] = v;
      } else if (MODE == 2) {
        const uint64_t m = __ballot(1);
        const uint32_t pos = cur + __popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull));
        log[static_cast<size_t>(wave) * iters * 64u + pos] = make_uint2(slot, __float_as_uint(v));
      } else if (MODE == 3) {
        const uint64_t m = __ballot(1);
        const uint32_t leader = __ffsll(static_cast<unsigned long long>(m)) - 1u;
        uint32_t base = 0u;
        if ((threadIdx.x & 63u) == leader) base = atomicAdd(&cnt[(blockIdx.x & 255u) * 16u], static_cast<uint32_t>(__popcll(m)));
        base = __shfl(base, static_cast<int>(leader));
        const uint32_t pos = base + __popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull));
        log[(static_cast<size_t>(blockIdx.x & 255u) << 21) + (pos & 0x1FFFFFu)] = make_uint2(slot, __float_as_uint(v));
      }
    }
    if (MODE == 2) cur += __popcll(__ballot(on));
  }
  if (acc == 12345.678f) sink[t] = acc;
}

template <int MODE>
static void run(const char* name, uint32_t foot_mb, uint32_t copies, uint32_t lane_keep, uint32_t alu, float* plane, uint2* log, uint32_t* cnt, float* sink) {
  const uint32_t blocks = 6104, iters = 512;
  const uint32_t slots = (foot_mb << 20) / 4u / copies;
  uint32_t shift = 0;
  while ((1u << shift) < slots) shift++;
  hipMemset(cnt, 0, 256 * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, plane, slots - 1u, copies > 1 ? shift : 31u, iters, lane_keep, alu, log, cnt, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = double(blocks) * 256 * iters * lane_keep / 64.0;
  std::printf("%-26s foot %4u MB copies %u lanes %2u/64 alu %3u : %7.3f ms  %6.2f G ops/s\n", name, foot_mb, copies, lane_keep, alu, ms, n / ms * 1e-6);
  std::fflush(stdout);
}

int main() {
  float *plane, *sink;
  uint2* log;
  uint32_t* cnt;
  hipMalloc(&plane, 1ull << 30);
  hipMemset(plane, 0, 1ull << 30);
  hipMalloc(&sink, 6104ull * 256 * 4);
  hipMalloc(&log, 6104ull * 4 * 512 * 64 * 8);   // 6.4 GB
  hipMalloc(&cnt, 256 * 64);
  for (uint32_t alu : {0u, 200u}) {
    run<4>("alu only", 8, 1, 64, alu, plane, log, cnt, sink);
    for (uint32_t lanes : {64u, 10u}) {
      run<0>("atomic_add_f32", 8, 1, lanes, alu, plane, log, cnt, sink);
      run<0>("atomic_add_f32", 64, 8, lanes, alu, plane, log, cnt, sink);
      run<0>("atomic_add_f32", 64, 1, lanes, alu, plane, log, cnt, sink);
      run<0>("atomic_add_f32", 1024, 1, lanes, alu, plane, log, cnt, sink);
      run<0>("atomic_add_f32", 1, 1, lanes, alu, plane, log, cnt, sink);
      run<1>("plain store", 64, 8, lanes, alu, plane, log, cnt, sink);
      run<2>("per-wave log store", 64, 8, lanes, alu, plane, log, cnt, sink);
      run<3>("shard log (ret. atomic)", 64, 8, lanes, alu, plane, log, cnt, sink);
    }
  }
  return 0;
}
