// Micro-benchmark: issue cost of the VALU instruction classes the trace kernel is made of, on gfx950 — so that per-phase
// instruction counts can be priced in cycles (a wave64 fp32 FMA holds its SIMD for 4 cycles; what do a 32-bit integer
// multiply, v_mad_u64_u32, a transcendental, a packed fp32 FMA cost?).
// Each kernel runs ITER x 64 instances of ONE instruction in 8 independent dependency chains per lane (inline asm, so the
// compiler can neither fuse nor drop them), 5 waves per SIMD resident like the trace kernel; cost = elapsed SIMD cycles per
// wave-instruction at the nominal 2.4 GHz, and relative to v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_bench.hip -o tools/valu_rate_bench.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256, 5) rate_kernel(uint32_t iters, uint32_t* sink) {
  uint32_t a[8];
  float2v p[8];
  uint64_t q[8];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    a[k] = t * 2654435761u + k * 40503u + 1u;
    p[k] = {1.0f + k * 0.001f + t * 1e-9f, 0.5f + k * 0.002f};
    q[k] = a[k];
  }
  const uint32_t c1 = 747796405u + (t & 1u) * 2u;  // in a VGPR: the multiplies take register operands like the kernel's
  const float cf = 0.999f + t * 1e-12f;
  uint64_t msk = 0x5555555555555555ull ^ blockIdx.x, m8[8] = {};
  uint32_t sg[8] = {};
  const uint32_t sgu = __builtin_amdgcn_readfirstlane(0x3f7fbe77u + (blockIdx.x & 3u));   // ~0.999f, provably wave-uniform
  const float sgf = __uint_as_float(sgu);
  asm volatile("s_mov_b64 vcc, %0" : : "s"(msk) : "vcc");
  for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (OP == 0) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(cf));
        REP8(X)
#undef X
      } else if (OP == 1) {
#define X(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c1));
        REP8(X)
#undef X
      } else if (OP == 2) {
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[k]) : "v"(a[k]), "v"(c1) : "vcc");
        REP8(X)
#undef X
      } else if (OP == 3) {
#define X(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[k]) : "v"(c1));
        REP8(X)
#undef X
      } else if (OP == 4) {
#define X(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[k]) : "v"(c1));
        REP8(X)
#undef X
      } else if (OP == 5) {
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
        REP8(X)
#undef X
      } else if (OP == 6) {
#define X(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
        REP8(X)
#undef X
      } else if (OP == 7) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[k]) : "v"(p[(k + 1) & 7]));
        REP8(X)
#undef X
      } else if (OP == 8) {
#define X(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(p[(k + 1) & 7]));
        REP8(X)
#undef X
      } else if (OP == 9) {
#define X(k) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c1));
        REP8(X)
#undef X
      } else if (OP == 18) {
#define X(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c1), "s"(msk));
        REP8(X)
#undef X
      } else if (OP == 19) {
#define X(k) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[k]) : "v"(a[(k + 1) & 7]));
        REP8(X)
#undef X
      } else if (OP == 20) {
#define X(k) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sg[k]) : "v"(a[k]));
        REP8(X)
#undef X
      } else if (OP == 21) {
#define X(k) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m8[k]) : "v"(a[k]), "v"(cf));
        REP8(X)
#undef X
      } else if (OP == 22) {
#define X(k) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[k]) : "v"(cf));
        REP8(X)
#undef X
      } else if (OP == 23) {
#define X(k) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[k]) : "v"(c1));
        REP8(X)
#undef X
      } else if (OP == 24) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "s"(sgu));
        REP8(X)
#undef X
      } else if (OP == 28) {
#define X(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "s"(sgu));
        REP8(X)
#undef X
      } else if (OP == 29) {
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[k]) : "v"(a[k]), "s"(sgu) : "vcc");
        REP8(X)
#undef X
      } else if (OP == 30) {
#define X(k) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7fbe77" : "+v"(a[k]) : "v"(cf));
        REP8(X)
#undef X
      } else if (OP == 31) {
#define X(k) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a[k]) : "s"(sgu));
        REP8(X)
#undef X
      } else if (OP == 32) {
#define X(k) asm volatile("v_xor_b32_e32 %0, 0x2c9277b5, %0" : "+v"(a[k]));
        REP8(X)
#undef X
      } else if (OP == 33) {
#define X(k) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[k]) : "s"(sgu));
        REP8(X)
#undef X
      } else if (OP == 34) {
#define X(k) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[k]) : "v"(cf));
        REP8(X)
#undef X
      } else if (OP == 25) {
#define X(k) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(cf), "v"(c1) : "vcc");
        X(0) X(1) X(2) X(3)
#undef X
      } else if (OP == 26) {
#define X(k) asm volatile("v_cvt_i32_f32_e32 %0, %0" : "+v"(a[k]));
        REP8(X)
#undef X
      } else if (OP == 27) {
#define X(k) asm volatile("v_writelane_b32 %0, %1, 7" : "+v"(a[k]) : "s"(sgu));
        REP8(X)
#undef X
      } else if (OP == 10) {
#define X(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(c1));
        REP8(X)
#undef X
      } else if (OP == 11) {
#define X(k) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[k]) : "v"(c1));
        REP8(X)
#undef X
      } else if (OP == 12) {
#define X(k) asm volatile("v_mad_u32_u16 %0, %0, %1, %0 op_sel:[0,1,0,0]" : "+v"(a[k]) : "v"(c1));
        REP8(X)
#undef X
      } else if (OP == 13) {
#define X(k) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(cf) : "vcc");
        REP8(X)
#undef X
      } else if (OP == 14) {
#define X(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c1));
        REP8(X)
#undef X
      } else if (OP == 15) {
#define X(k) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
        REP8(X)
#undef X
      } else if (OP == 16) {
#define X(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(p[(k + 1) & 7]));
        REP8(X)
#undef X
      } else if (OP == 17) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %0\n v_mul_lo_u32 %2, %2, %3" : "+v"(a[k]), "+v"(a[(k + 4) & 7]) : "v"(cf), "v"(c1));
        // (a mixed stream: does a quarter-rate multiply overlap with full-rate work of the same wave?  no — in order)
        X(0) X(1) X(2) X(3)
#undef X
      }
    }
  }
  uint32_t s = 0;
  for (int k = 0; k < 8; k++) s ^= sg[k] ^ static_cast<uint32_t>(m8[k]);
#pragma unroll
  for (int k = 0; k < 8; k++) s ^= a[k] ^ __float_as_uint(p[k].x) ^ __float_as_uint(p[k].y) ^ static_cast<uint32_t>(q[k]);
  if (s == 0x12345678u) sink[0] = s;
}

template <int OP>
static double run(const char* name, int per_iter, double base_cyc) {
  const uint32_t iters = 2000;
  const int blocks = 256 * 5;   // 5 workgroups of 4 waves per CU: 5 waves per SIMD
  uint32_t* sink;
  (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, 10u, sink);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, iters, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD = waves per SIMD x iters x per_iter
  const double insts_per_simd = 5.0 * iters * per_iter;
  const double cyc = ms * 1e-3 * 2.4e9 / insts_per_simd;
  printf("%-34s %8.3f ms  %6.2f cycles per wave64 instruction  (%.2fx v_fma_f32)\n", name, ms, cyc, base_cyc > 0 ? cyc / base_cyc : 1.0);
  (void)hipFree(sink);
  return cyc;
}

int main() {
  const double b = run<0>("v_fma_f32", 64, 0);
  run<1>("v_mul_lo_u32", 64, b);
  run<2>("v_mad_u64_u32", 64, b);
  run<14>("v_mul_hi_u32", 64, b);
  run<3>("v_mul_u32_u24", 64, b);
  run<4>("v_mad_u32_u24", 64, b);
  run<12>("v_mad_u32_u16 (op_sel)", 64, b);
  run<5>("v_rcp_f32", 64, b);
  run<6>("v_sqrt_f32", 64, b);
  run<15>("v_rsq_f32", 64, b);
  run<7>("v_pk_fma_f32 (2 FMAs per lane)", 64, b);
  run<8>("v_pk_mul_f32", 64, b);
  run<16>("v_pk_add_f32", 64, b);
  run<13>("v_cmp_gt_f32", 64, b);
  run<10>("v_xor_b32", 64, b);
  run<11>("v_lshrrev_b32", 64, b);
  run<17>("v_fma_f32 + v_mul_lo_u32 pairs", 64, b);
  run<9>("v_cndmask_b32_e32 (vcc)", 64, b);
  run<18>("v_cndmask_b32_e64 (sgpr mask)", 64, b);
  run<25>("v_cmp_e32 + v_cndmask_e32 pairs", 64, b);
  run<21>("v_cmp_gt_f32_e64 -> sgpr pair", 64, b);
  run<19>("v_mov_b32", 64, b);
  run<20>("v_readlane_b32", 64, b);
  run<27>("v_writelane_b32", 64, b);
  run<22>("v_max_f32", 64, b);
  run<23>("v_add_u32", 64, b);
  run<24>("v_fma_f32 with an SGPR operand", 64, b);
  run<26>("v_cvt_i32_f32", 64, b);
  run<34>("v_mul_f32 (VOP2, VGPR operands)", 64, b);
  run<33>("v_mul_f32 with an SGPR operand", 64, b);
  run<30>("v_fmaak_f32 (32-bit literal)", 64, b);
  run<28>("v_mul_lo_u32 with an SGPR operand", 64, b);
  run<29>("v_mad_u64_u32 with an SGPR operand", 64, b);
  run<31>("v_add_u32 with an SGPR operand", 64, b);
  run<32>("v_xor_b32 with a 32-bit literal", 64, b);
  return 0;
}
