// halo_kernels.hip — the kernels around the trace kernel (the hit log's split and per-tile passes, binned-accumulation passes,
// plane fold, device consumer) and the launch dispatcher.  The trace kernel itself is a template in halo_trace.inl, instantiated per MODE in
// halo_trace_m{0,1,2}.hip.
#include "halo_trace.inl"

namespace halo {

// Tile sums in 64-bit fixed point.  An fp64 LDS add is the slow kind here (tools/lds_atomic_bench.hip: ds_add_f64 1.5 T adds/s, ds_add_u64
// 2.9 T), and these passes are LDS-bound (the index unit ~70 % busy, profiles/r03_bench4_pmc_lds.txt) — so a record's weight becomes a
// 64-bit integer, weight x 2^F rounded to nearest, and integer sums do not depend on the order of the adds.  F comes with the launch
// (halo_backend.cpp fix_frac_bits): the largest F <= 32 for which 4 x (largest ray weight of the session) x (rays of the launch) x 2^F
// stays below 2^62 — no slot can sum to more than that bound (a ray's exits never outweigh the ray; x2 for the dual-fisheye overlap, x2
// for the CMF rows of the X/Y/Z pass), so the unsigned sums cannot wrap.  F = 32 (resolution 2.3e-10) for every launch of unit-weight
// rays up to 2^28; an illuminant session (spd weights ~100) of 2^26 rays runs F = 28.  (Round 3 had F fixed at 32 on the assumption
// "weight <= 1" and a signed read-out: 4e9 of weight in one slot wrapped negative — ADVICE r3.)  A NaN or negative weight adds nothing.
// A tile list is read once, by one workgroup, a moment after the split pass wrote it: loaded non-temporally (four dword loads the compiler merges
// into one `global_load_dwordx4 ... nt`) the per-tile pass of configs[1] takes 108 us instead of 137 — the same pass launched a second time on the
// same lists takes 107 (round 6, rocprofv3 per-kernel minima on one box; non-temporal STORES in the split pass cost it more than they save here).
// Short lists (a small session's: they sit in the caches whole) keep the ordinary load — `--config 4d` lost 0.8 % with the other; `once` is
// workgroup-uniform (the tile's record count).
constexpr uint32_t kListOnceMin = 1u << 16;
template <bool ONCE>   // (a run-time choice inside one loop comes out as ordinary loads on both sides: the loop exists twice instead)
__device__ __forceinline__ uint4 load_list_u4(const uint4* p) {
  if constexpr (!ONCE) return *p;
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
  return make_uint4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2), __builtin_nontemporal_load(q + 3));
}

struct FixQ {
  double to_fix, from_fix;
  __device__ __forceinline__ explicit FixQ(uint32_t frac_bits)
      : to_fix(__hiloint2double(static_cast<int>((1023u + frac_bits) << 20), 0)), from_fix(__hiloint2double(static_cast<int>((1023u - frac_bits) << 20), 0)) {}
  __device__ __forceinline__ unsigned long long fix(float v) const {   // round to nearest; v <= the launch's bound by construction
    return static_cast<unsigned long long>(static_cast<long long>(fma(static_cast<double>(fmaxf(v, 0.0f)), to_fix, 0.5)));
  }
  __device__ __forceinline__ float unfix(unsigned long long a) const { return static_cast<float>(static_cast<double>(a) * from_fix); }
};

// kBinSplit workgroups per image tile: each sums its share of the tile's hit list in a 64 KB LDS tile (8 independent
// loads in flight per thread — the loop is load-latency-bound otherwise) and adds the non-zero slots to the plane.
constexpr int kBinBlock = 1024;
constexpr uint32_t kBinSplit = 2u;
__global__ void __launch_bounds__(kBinBlock) halo_bin_accumulate_kernel(float* __restrict__ plane, const uint2* __restrict__ list, uint32_t cap,
                                                                         const uint32_t* __restrict__ cnt, uint32_t tiles_log2, uint32_t frac_bits) {
  const FixQ fq(frac_bits);
  // fp64 sums, and not for precision: on gfx950 ds_add_f32 retires ~0.3 lanes per clock per CU (200 G adds/s chip-wide) while
  // ds_add_f64 runs at 1500 G/s and ds_add_u32 at 4600 G/s (tools/lds_atomic_bench.hip) — the fp32 LDS atomic is the slow one,
  // and this pass is one LDS add per hit.  128 KB of the CU's 160 KB: one workgroup of 16 waves per CU.
  __shared__ unsigned long long acc[1u << kBinTileLog2];   // fixed point (FixQ)
  const uint32_t tile = blockIdx.x / kBinSplit, part = blockIdx.x % kBinSplit;
  const uint32_t n = min(cnt[tile * kBinCntStride], cap);
  const uint32_t lo = static_cast<uint32_t>(static_cast<uint64_t>(n) * part / kBinSplit);
  const uint32_t hi = static_cast<uint32_t>(static_cast<uint64_t>(n) * (part + 1u) / kBinSplit);
  if (hi <= lo) return;
  for (uint32_t j = threadIdx.x; j < (1u << kBinTileLog2); j += kBinBlock) acc[j] = 0ull;
  __syncthreads();
  const uint2* src = list + static_cast<size_t>(tile) * cap;
  constexpr uint32_t kU = 8u;
  uint32_t i = lo + threadIdx.x;
  for (; i + (kU - 1u) * kBinBlock < hi; i += kU * kBinBlock) {
    uint2 h[kU];
#pragma unroll
    for (uint32_t u = 0; u < kU; ++u) h[u] = src[i + u * kBinBlock];
#pragma unroll
    for (uint32_t u = 0; u < kU; ++u) atomicAdd(&acc[h[u].x >> tiles_log2], fq.fix(__uint_as_float(h[u].y)));
  }
  for (; i < hi; i += kBinBlock) {
    const uint2 h = src[i];
    atomicAdd(&acc[h.x >> tiles_log2], fq.fix(__uint_as_float(h.y)));
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < (1u << kBinTileLog2); j += kBinBlock) {
    const float v = fq.unfix(acc[j]);
    if (v != 0.0f) atomic_add_f32(plane + ((static_cast<size_t>(j) << tiles_log2) | tile), v);
  }
}

hipError_t launch_bin_accumulate(float* plane, const HitRec* list, uint32_t cap, uint32_t* cnt, uint32_t tiles, uint32_t frac_bits, hipStream_t stream) {
  hipLaunchKernelGGL(halo_bin_accumulate_kernel, dim3(tiles * kBinSplit), dim3(kBinBlock), 0, stream, plane, reinterpret_cast<const uint2*>(list), cap, cnt, static_cast<uint32_t>(__builtin_ctz(tiles)), frac_bits);
  return hipGetLastError();
}

// Interleaved tiles of one accumulation plane (the hit log's full-sky and X/Y/Z routes).  A slot is row << s | column (MonoSlot:
// row = pixel mod 1024 — for the usual image widths the image column mod 1024 — and column = a hash of pixel / 1024, i.e. of the
// image row).  Contiguous slot ranges are bands of image columns and run 6x uneven on a full-sky render, whose light sits
// around the sun's azimuth; the column alone is just as uneven (the sun's elevation).  So a tile takes both: with T = 2^t tiles
// and S = 2^s columns, tile = (column + row) mod T for T <= S, else (row mod T/S) * S + (column + row / (T/S)) mod S; `local` is
// what is left of the slot.  1024 * S / T slots per tile, a bijection (slot_of inverts it).
struct TileMap {
  uint32_t s, t;
  __device__ __forceinline__ void split(uint32_t slot, uint32_t& tile, uint32_t& local) const {
    const uint32_t row = slot >> s, col = slot & ((1u << s) - 1u);
    if (t <= s) {
      tile = (col + row) & ((1u << t) - 1u);
      local = (row << (s - t)) | (col >> t);
    } else {
      const uint32_t d = t - s;
      tile = ((row & ((1u << d) - 1u)) << s) | ((col + (row >> d)) & ((1u << s) - 1u));
      local = row >> d;
    }
  }
  __device__ __forceinline__ uint32_t slot_of(uint32_t tile, uint32_t local) const {
    if (t <= s) {
      const uint32_t row = local >> (s - t), chi = local & ((1u << (s - t)) - 1u);
      return (row << s) | (chi << t) | ((tile - row) & ((1u << t) - 1u));
    }
    const uint32_t d = t - s, r_hi = local;
    return ((((r_hi << d) | (tile >> s))) << s) | (((tile & ((1u << s) - 1u)) - r_hi) & ((1u << s) - 1u));
  }
};

// ---- the split pass: hit records of a source list dealt out to per-tile lists ----
// Two callers.  (1) Two-level binning, accumulators of more than 512 tiles (per-wavelength planes: 64 planes x 2 Mi slots = 8192
// tiles): the trace kernel can feed at most 512 lists from its 1536-record LDS buffer (fewer than ~3 records per list and flush
// and the appends stop coalescing), so its lists are COARSE — list l1 holds the hits of `fan` consecutive tiles (bin_shift = 14 +
// fan_log2) — and this kernel deals each coarse list out to its `fan` tile lists, kSplitParts workgroups per list.  (2) The hit
// log (halo_trace.inl log_hit): one UNSORTED region per trace workgroup, every region may hold every tile of an accumulator of
// at most kSplitFanMax tiles; one workgroup per region.
// A workgroup takes 4096 records per step.  Scattered 8-byte stores run at ~90 G/s on this part whatever their locality
// (tools/atomic_rate_bench.hip), so the records of a step are first SORTED by tile in LDS — rank from an LDS counter per tile,
// run offsets from a scan of the counters — and then written by consecutive lanes: a wave's store covers a few runs of
// consecutive addresses instead of 64 lists (the log split went from 0.67 to 0.37 ms per 68 M records with that).  One returning
// global atomic per tile and step reserves the run's place in the tile list; a tile list that overflows falls back to direct
// atomics on the plane.
constexpr uint32_t kSplitParts = 8u;
#ifndef HALO_SPLIT_THREADS
#define HALO_SPLIT_THREADS 1024
#endif
#ifndef HALO_SPLIT_PER
#define HALO_SPLIT_PER 16
#endif
// the hit log's split: records per step = threads x per.  Round 4 tried the shapes that put two workgroups on a CU (the kernel waits 61 % of its
// wave cycles, profiles/r04_ref_bench_light_single_ms_pmc_wait.txt): 512 x 16 and 1024 x 8 (8 Ki records, 70 KB) and 512 x 32 (16 Ki, eight waves)
// are 3-4 %, 4-5 % and 1.5-3 % SLOWER on the whole step of bench_light_single_ms / configs[1] / ms_multi_crystal — the length of the runs a step
// writes (32 records = 256 B per list at 512 lists) matters more than a second workgroup to hide the latencies behind
constexpr uint32_t kLogSplitThreads = HALO_SPLIT_THREADS, kLogSplitPer = HALO_SPLIT_PER;
// Step sizes (threads x records per thread).  The hit log has 128..256 destinations per step, and its stores only coalesce
// when a step is large: 16384 records (a region is read in one or two steps; 128 KB of LDS, one workgroup per CU) runs
// at 0.23 ms per 58 M records where 4096 takes 0.36 and 1024 takes 1.1.  The coarse lists of the two-level route have <= 32
// destinations per step, whose runs are long anyway, and ~1000 long chains that want many workgroups resident: 256 x 16.
// Tile of a record: ((slot & slot_mask) >> tile_log2) & (fan - 1) — contiguous 16 Ki-slot tiles of the plane array (tile_log2 14),
// or, X/Y/Z hit log, interleaved tiles of one plane (tile_log2 0, mix_log2 = log2 of the plane's columns; the wavelength-pool
// entry above the slot is cut off by slot_mask).  kSplitFanMax = destination tiles per source list.
struct SplitXyz {              // what a record that finds its tile list full is added to
  const WlEntryDev* pool;      // X/Y/Z hit log; nullptr = scalar planes: plane[slot] += w
  uint32_t pool_size;          // CMF codes: pool entries, then pool_size + c = "the weight is channel c already"
  uint32_t plane_stride;
  double* ovf;                 // the planes' fp64 twin (DispatchParams::ovf); nullptr = the fp32 plane itself
  uint32_t* ovf_flag;
  uint32_t plane_log2;         // log2 of a plane copy's floats, and of the privatised copies per plane: the twin has no copies (TwinOffset)
  uint32_t copies_log2;
};
__device__ __forceinline__ void split_overflow(float* plane, const SplitXyz& x, size_t off, float v) {
  if (x.ovf != nullptr) {
    atomicAdd(x.ovf + TwinOffset(off, x.plane_log2, x.copies_log2), static_cast<double>(v));
    *x.ovf_flag = 1u;
  } else {
    atomic_add_f32(plane + off, v);
  }
}
template <uint32_t kSplitBlock, uint32_t kSplitPer, uint32_t kSplitFanMax, bool kXyz, bool kMix>
__global__ void __launch_bounds__(kSplitBlock) halo_split_kernel(float* __restrict__ plane, const uint2* __restrict__ list1, uint32_t cap1,
                                                                 const uint32_t* __restrict__ cnt1, uint32_t cnt1_stride, uint32_t parts,
                                                                 uint2* __restrict__ list2, uint32_t cap2, uint32_t* __restrict__ cnt2, uint32_t fan_log2,
                                                                 uint32_t coarse, uint32_t slot_mask, uint32_t tile_log2, uint32_t mix_log2, SplitXyz xyz) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cnt[kSplitFanMax];
  __shared__ __attribute__((aligned(16))) uint32_t s_off[kSplitFanMax];
  __shared__ uint32_t s_base[kSplitFanMax];
  __shared__ uint2 s_rec[kSplitBlock * kSplitPer];
  static_assert(kSplitBlock >= kSplitFanMax && kSplitFanMax % 256u == 0u, "one thread per tile counter; the scan takes 4 per lane and round");
  const uint32_t l1 = blockIdx.x / parts, part = blockIdx.x % parts;
  const uint32_t n = min(cnt1[static_cast<size_t>(l1) * cnt1_stride], cap1);
  const uint32_t lo = static_cast<uint32_t>(static_cast<uint64_t>(n) * part / parts);
  const uint32_t hi = static_cast<uint32_t>(static_cast<uint64_t>(n) * (part + 1u) / parts);
  const uint32_t fmask = (1u << fan_log2) - 1u;
  // kMix: interleaved tiles (TileMap, s = mix_log2, t = fan_log2); else contiguous ranges of 2^tile_log2 slots
  auto tile_of = [&](uint32_t x) {
    const uint32_t sl = x & slot_mask;
    if (kMix) {   // interleaved tiles of ONE plane; the planes of a per-entry-plane session lie back to back, each with its own 2^fan_log2 tiles
      uint32_t tile, local;
      TileMap{mix_log2, fan_log2}.split(sl & ((1u << (mix_log2 + 10u)) - 1u), tile, local);
      return ((sl >> (mix_log2 + 10u)) << fan_log2) | tile;
    }
    return (sl >> tile_log2) & fmask;
  };
  const uint32_t tile0 = coarse ? (l1 << fan_log2) : 0u;   // the first destination tile of this source list
  const uint2* src = list1 + static_cast<size_t>(l1) * cap1;
  // The steps of a workgroup are a chain of dependent latencies (loads, LDS ranks, the reserving atomic, stores), so the next
  // step's records are loaded while this step's are written out, and the reserving atomic flies during the LDS scatter.
  uint2 h[kSplitPer];
  uint32_t rank[kSplitPer];
#pragma unroll
  for (uint32_t u = 0; u < kSplitPer; ++u) {
    const uint32_t i = lo + u * kSplitBlock + threadIdx.x;
    h[u] = i < hi ? src[i] : make_uint2(0xFFFFFFFFu, 0u);
  }
  if (threadIdx.x < kSplitFanMax) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  for (uint32_t b0 = lo; b0 < hi; b0 += kSplitBlock * kSplitPer) {   // workgroup-uniform trip count
#pragma unroll
    for (uint32_t u = 0; u < kSplitPer; ++u)
      rank[u] = h[u].x != 0xFFFFFFFFu ? atomicAdd(&s_cnt[tile_of(h[u].x)], 1u) : 0u;
    __syncthreads();
    if (threadIdx.x < 64u) {   // exclusive scan of the counters by one wave: 4 per lane and round
      uint32_t carry = 0u;
#pragma unroll
      for (uint32_t g = 0; g < kSplitFanMax; g += 256u) {
        const uint4 c = *reinterpret_cast<const uint4*>(&s_cnt[g + threadIdx.x * 4u]);
        const uint32_t own = c.x + c.y + c.z + c.w;
        uint32_t incl = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t up = __shfl_up(incl, d);
          if (static_cast<int>(threadIdx.x) >= d) incl += up;
        }
        const uint32_t ex = carry + incl - own;
        *reinterpret_cast<uint4*>(&s_off[g + threadIdx.x * 4u]) = make_uint4(ex, ex + c.x, ex + c.x + c.y, ex + c.x + c.y + c.z);
        carry += __shfl(incl, 63);
      }
    }
    const uint32_t c_own = threadIdx.x < kSplitFanMax ? s_cnt[threadIdx.x] : 0u;
    const uint32_t base = c_own ? atomicAdd(&cnt2[static_cast<size_t>(tile0 + threadIdx.x) * kBinCntStride], c_own) : 0u;
    __syncthreads();
    if (threadIdx.x < kSplitFanMax) s_cnt[threadIdx.x] = 0u;   // for the next step (ranked after the barrier that ends this one)
#pragma unroll
    for (uint32_t u = 0; u < kSplitPer; ++u)
      if (h[u].x != 0xFFFFFFFFu) s_rec[s_off[tile_of(h[u].x)] + rank[u]] = h[u];
    const uint32_t nb = b0 + kSplitBlock * kSplitPer;
#pragma unroll
    for (uint32_t u = 0; u < kSplitPer; ++u) {
      const uint32_t i = nb + u * kSplitBlock + threadIdx.x;
      h[u] = i < hi ? src[i] : make_uint2(0xFFFFFFFFu, 0u);
    }
    if (threadIdx.x < kSplitFanMax) s_base[threadIdx.x] = base;
    __syncthreads();
    const uint32_t n_step = min(kSplitBlock * kSplitPer, hi - b0);
#pragma unroll
    for (uint32_t u = 0; u < kSplitPer; ++u) {
      const uint32_t i = u * kSplitBlock + threadIdx.x;
      if (i >= n_step) continue;
      const uint2 r = s_rec[i];
      const uint32_t t = tile_of(r.x);
      const uint32_t pos = s_base[t] + (i - s_off[t]);
      if (pos < cap2) {
        list2[static_cast<size_t>(tile0 + t) * cap2 + pos] = r;
      } else if (kXyz) {
        const uint32_t code = r.x >> kLogWlShift;
        float c[3];
        if (code < xyz.pool_size) {
          const WlEntryDev e = xyz.pool[code];
          c[0] = e.cmf_x, c[1] = e.cmf_y, c[2] = e.cmf_z;
        } else {
          for (uint32_t k = 0; k < 3u; ++k) c[k] = code == xyz.pool_size + k ? 1.0f : 0.0f;
        }
        const size_t at = r.x & slot_mask;
        for (uint32_t k = 0; k < 3u; ++k)
          if (c[k] != 0.0f) split_overflow(plane, xyz, at + static_cast<size_t>(k) * xyz.plane_stride, c[k] * __uint_as_float(r.y));
      } else {
        split_overflow(plane, xyz, r.x, __uint_as_float(r.y));
      }
    }
    __syncthreads();
  }
}

// One workgroup per tile of 16 Ki CONSECUTIVE slots: sums the tile's list in LDS and adds the tile to the plane with plain
// coalesced float4 read-modify-writes — the workgroup is the only writer of those slots while this kernel runs (direct
// atomics of the trace / split kernels are ordered before it on the stream).
__global__ void __launch_bounds__(kBinBlock) halo_bin_accumulate_range_kernel(float* __restrict__ plane, const uint2* __restrict__ list, uint32_t cap,
                                                                               const uint32_t* __restrict__ cnt, uint32_t tile_log2, uint32_t frac_bits) {
  __shared__ __attribute__((aligned(16))) unsigned long long acc[1u << kBinTileLog2];   // fixed point (FixQ); tile_log2 <= kBinTileLog2
  const FixQ fq(frac_bits);
  const uint32_t tile = blockIdx.x;
  const uint32_t n = min(cnt[static_cast<size_t>(tile) * kBinCntStride], cap);
  if (n == 0u) return;
  const uint32_t slots = 1u << tile_log2, mask = slots - 1u;
  for (uint32_t j = threadIdx.x; j < slots; j += kBinBlock) acc[j] = 0ull;
  __syncthreads();
  const uint2* src = list + static_cast<size_t>(tile) * cap;
#ifndef HALO_ACC_U
#define HALO_ACC_U 4
#endif
  constexpr uint32_t kU = HALO_ACC_U;
  uint32_t i = threadIdx.x;
#ifndef HALO_ACC_NARROW
  // two records per 16-byte load (the lists start on whole 128-byte lines): configs[1]'s pass 164 -> 137 us; eight or sixteen 8-byte loads in
  // flight instead of four gave 152 / 149
  const uint4* src4 = reinterpret_cast<const uint4*>(src);
  const uint32_t n2 = n / 2u;
  auto stream = [&](auto once) {
    for (; i + (kU - 1u) * kBinBlock < n2; i += kU * kBinBlock) {
      uint4 h[kU];
#pragma unroll
      for (uint32_t u = 0; u < kU; ++u) h[u] = load_list_u4<decltype(once)::value>(&src4[i + u * kBinBlock]);
#pragma unroll
      for (uint32_t u = 0; u < kU; ++u) {
        atomicAdd(&acc[h[u].x & mask], fq.fix(__uint_as_float(h[u].y)));
        atomicAdd(&acc[h[u].z & mask], fq.fix(__uint_as_float(h[u].w)));
      }
    }
  };
  if (n >= kListOnceMin) stream(std::true_type{});
  else stream(std::false_type{});
  for (uint32_t r = 2u * i; r < n; r += 2u * kBinBlock) {   // what is left of this thread's pairs, and the odd last record
    for (uint32_t q = r; q < min(r + 2u, n); ++q) {
      const uint2 h = src[q];
      atomicAdd(&acc[h.x & mask], fq.fix(__uint_as_float(h.y)));
    }
  }
  i = n;
#else
  for (; i + (kU - 1u) * kBinBlock < n; i += kU * kBinBlock) {
    uint2 h[kU];
#pragma unroll
    for (uint32_t u = 0; u < kU; ++u) h[u] = src[i + u * kBinBlock];
#pragma unroll
    for (uint32_t u = 0; u < kU; ++u) atomicAdd(&acc[h[u].x & mask], fq.fix(__uint_as_float(h[u].y)));
  }
#endif
  for (; i < n; i += kBinBlock) {
    const uint2 h = src[i];
    atomicAdd(&acc[h.x & mask], fq.fix(__uint_as_float(h.y)));
  }
  __syncthreads();
  float4* dst = reinterpret_cast<float4*>(plane + (static_cast<size_t>(tile) << tile_log2));
  for (uint32_t j = threadIdx.x; j < slots / 4u; j += kBinBlock) {
    const float4 v = make_float4(fq.unfix(acc[4u * j]), fq.unfix(acc[4u * j + 1u]), fq.unfix(acc[4u * j + 2u]), fq.unfix(acc[4u * j + 3u]));
    if (v.x != 0.0f || v.y != 0.0f || v.z != 0.0f || v.w != 0.0f) {
      float4 q = dst[j];
      q.x += v.x;
      q.y += v.y;
      q.z += v.z;
      q.w += v.w;
      dst[j] = q;
    }
  }
}

// The hit log's per-tile pass.  CH = 1: one scalar plane (discrete wavelength), records {slot, w}, tiles of 16 Ki slots.
// CH = 3: an illuminant session on X, Y, Z planes, records {slot in ONE plane | CMF code << kLogWlShift, w}, tiles of 4 Ki slots:
// the code picks the CMF row (pool entries, then three unit rows for a cached pixel's X, Y, Z records; staged in LDS) and the
// record goes into X, Y and Z tiles with three fp64 LDS adds.  This is where an illuminant session's colour is made: no plane
// per pool entry (31 x 128 tiles at configs[4], a two-level split) and a fold over 3 planes instead of 31.
// Tiles are interleaved over the plane (TileMap) and as many as the split pass can feed (256 / 512), whatever the image size:
// the pass has one workgroup per tile, and a 512 x 256 image cut into 16 Ki-slot tiles would leave it 8 workgroups.  A tile's
// slots are scattered, so its write-out is plain 4-byte read-modify-writes (the workgroup is the only writer of its slots
// while this kernel runs).
template <uint32_t CH>
__global__ void __launch_bounds__(kBinBlock) halo_log_accumulate_kernel(float* __restrict__ planes, uint32_t plane_stride, const uint2* __restrict__ list, uint32_t cap,
                                                                        const uint32_t* __restrict__ cnt, const WlEntryDev* __restrict__ pool, uint32_t pool_size,
                                                                        uint32_t tiles_log2, uint32_t s_log2, uint32_t frac_bits) {
  constexpr uint32_t kTileLog2 = CH == 3u ? 12u : 14u;
  __shared__ __attribute__((aligned(16))) unsigned long long acc[CH][1u << kTileLog2];   // fixed point (FixQ)
  const FixQ fq(frac_bits);
  __shared__ __attribute__((aligned(16))) float s_cmf[CH == 3u ? HALO_WL_POOL_MAX + 3 : 1][4];   // rows of 16 bytes: one LDS read per record
  const uint32_t tile = blockIdx.x;   // plane << tiles_log2 | tile of that plane (CH = 1: the scalar planes of a per-entry-plane session lie back to back)
  const uint32_t n = min(cnt[static_cast<size_t>(tile) * kBinCntStride], cap);
  if (n == 0u) return;
  const TileMap map{s_log2, tiles_log2};
  const uint32_t tile_in = tile & ((1u << tiles_log2) - 1u), plane_of_tile = tile >> tiles_log2, in_plane = (1u << (s_log2 + 10u)) - 1u;
  const uint32_t slots = 1u << (s_log2 + 10u - tiles_log2);   // <= 1 << kTileLog2
  for (uint32_t c = 0; c < CH; ++c)
    for (uint32_t j = threadIdx.x; j < slots; j += kBinBlock) acc[c][j] = 0ull;
  if constexpr (CH == 3u) {
    for (uint32_t j = threadIdx.x; j < pool_size + 3u; j += kBinBlock) {
      const bool unit = j >= pool_size;
      s_cmf[j][0] = unit ? (j == pool_size ? 1.0f : 0.0f) : pool[j].cmf_x;
      s_cmf[j][1] = unit ? (j == pool_size + 1u ? 1.0f : 0.0f) : pool[j].cmf_y;
      s_cmf[j][2] = unit ? (j == pool_size + 2u ? 1.0f : 0.0f) : pool[j].cmf_z;
      s_cmf[j][3] = 0.0f;
    }
  }
  __syncthreads();
  const uint2* src = list + static_cast<size_t>(tile) * cap;
  constexpr uint32_t kU = CH == 3u ? 8u : 4u, kSlotMask = CH == 3u ? (1u << kLogWlShift) - 1u : 0xFFFFFFFFu;
  auto add = [&](uint2 h) {
    uint32_t t_unused, s;
    map.split(h.x & kSlotMask & in_plane, t_unused, s);
    const float w = __uint_as_float(h.y);
    if constexpr (CH == 3u) {
      const uint32_t code = h.x >> kLogWlShift;
      const float4 c4 = *reinterpret_cast<const float4*>(s_cmf[code]);
      if (c4.x != 0.0f) atomicAdd(&acc[0][s], fq.fix(c4.x * w));
      if (c4.y != 0.0f) atomicAdd(&acc[1][s], fq.fix(c4.y * w));
      if (c4.z != 0.0f) atomicAdd(&acc[2][s], fq.fix(c4.z * w));
    } else {
      atomicAdd(&acc[0][s], fq.fix(w));
    }
  };
  // Where the time goes (configs[4], 137 M records, 1.00 ms): without the adds the pass takes 0.27 ms, without the write-out 0.94 — the
  // three fp64 LDS adds per record are 0.67 ms, 41 % of the rate tools/lds_atomic_bench.hip reaches, with the LDS index unit busy 13 % of the
  // time (SQ_LDS_IDX_ACTIVE / SQ_BUSY_CYCLES, profiles/r03_bench4_pmc_lds.txt): a latency chain, not a throughput limit.  One workgroup per
  // CU (96 KB of tiles) is four waves per SIMD, and each record was its own chain: CMF row read from LDS -> wait -> three adds.  So a batch's
  // rows are all requested first, then its adds issued back to back.  (Requesting the next batch's global loads early changes nothing.)
  // (two records per 16-byte load, as in halo_bin_accumulate_range_kernel: the lists start on whole 128-byte lines, and the order in which
  // records reach the fixed-point sums changes nothing)
  const uint4* src4 = reinterpret_cast<const uint4*>(src);
  const uint32_t n2 = n / 2u;
  constexpr uint32_t kU2 = kU / 2u;
  uint32_t i = threadIdx.x;
  auto stream = [&](auto once) {
  for (; i + (kU2 - 1u) * kBinBlock < n2; i += kU2 * kBinBlock) {
    uint2 h[kU];
#pragma unroll
    for (uint32_t u = 0; u < kU2; ++u) {
      const uint4 q = load_list_u4<decltype(once)::value>(&src4[i + u * kBinBlock]);
      h[2u * u] = make_uint2(q.x, q.y);
      h[2u * u + 1u] = make_uint2(q.z, q.w);
    }
    if constexpr (CH == 3u) {
      float4 c4[kU];
      uint32_t sl[kU];
#pragma unroll
      for (uint32_t u = 0; u < kU; ++u) {
        uint32_t t_unused;
        map.split(h[u].x & kSlotMask & in_plane, t_unused, sl[u]);
        c4[u] = *reinterpret_cast<const float4*>(s_cmf[h[u].x >> kLogWlShift]);
      }
#pragma unroll
      for (uint32_t u = 0; u < kU; ++u) {
        const float w = __uint_as_float(h[u].y);
        if (c4[u].x != 0.0f) atomicAdd(&acc[0][sl[u]], fq.fix(c4[u].x * w));
        if (c4[u].y != 0.0f) atomicAdd(&acc[1][sl[u]], fq.fix(c4[u].y * w));
        if (c4[u].z != 0.0f) atomicAdd(&acc[2][sl[u]], fq.fix(c4[u].z * w));
      }
    } else {
#pragma unroll
      for (uint32_t u = 0; u < kU; ++u) add(h[u]);
    }
  }
  };
  if (n >= kListOnceMin) stream(std::true_type{});
  else stream(std::false_type{});
  for (uint32_t r = 2u * i; r < n; r += 2u * kBinBlock) {   // what is left of this thread's pairs, and the odd last record
    add(src[r]);
    if (r + 1u < n) add(src[r + 1u]);
  }
  __syncthreads();
  for (uint32_t c = 0; c < CH; ++c) {
    float* dst = planes + static_cast<size_t>(c) * plane_stride + (static_cast<size_t>(plane_of_tile) << (s_log2 + 10u));
    for (uint32_t j = threadIdx.x; j < slots; j += kBinBlock) {
      const float v = fq.unfix(acc[c][j]);
      if (v != 0.0f) dst[map.slot_of(tile_in, j)] += v;
    }
  }
}

// (`before_sums`, all three routes below: the per-tile sums end in PLAIN read-modify-writes of the planes, so they may not run beside another
// launch that adds to the same planes; the event, when given, is waited for between the split pass and the sums — see halo_backend.cpp.)
// hit-log route, one scalar plane: regions -> `tiles` (a power of two <= 256) tiles -> the plane.  Contiguous tiles (plain float4
// write-out, 3-5 % faster at configs[1] / [2]) where the lists have room for an uneven image — renders that cull most exits,
// ~1.2 records per ray against room for 8 — interleaved tiles (halo_log_accumulate_kernel) for full-sky renders.
// `planes` > 1: the scalar planes of a per-entry-plane illuminant session (back to back), `tiles` interleaved tiles EACH, planes x tiles <= 512.
hipError_t launch_log_route(float* plane, const HitRec* log, uint32_t cap1, const uint32_t* cnt1, uint32_t regions, HitRec* list2, uint32_t cap2, uint32_t* cnt2,
                            uint32_t tiles, uint32_t planes, uint32_t s_log2, bool interleaved, uint32_t frac_bits, double* ovf, uint32_t* ovf_flag, uint32_t copies_log2, hipStream_t stream,
                            hipEvent_t before_sums) {
  const uint32_t tiles_log2 = static_cast<uint32_t>(__builtin_ctz(tiles)), tile_log2 = s_log2 + 10u - tiles_log2;   // slots per tile <= 16 Ki
  // more than 256 lists — several planes, or ONE plane of an 8 Mi-pixel image cut into 512 tiles (an illuminant pool of one entry on 4096 x 2048:
  // found by tests/test_gpu_fuzz.py, the 256-list kernel below wrote past its counters) — take the 512-list split kernel
  if (planes > 1u || tiles > 256u) {
    hipLaunchKernelGGL((halo_split_kernel<kLogSplitThreads, kLogSplitPer, 512u, false, true>), dim3(regions), dim3(kLogSplitThreads), 0, stream, plane, reinterpret_cast<const uint2*>(log), cap1, cnt1, 1u,
                       1u, reinterpret_cast<uint2*>(list2), cap2, cnt2, tiles_log2, 0u, 0xFFFFFFFFu, 0u, s_log2, SplitXyz{nullptr, 0u, 0u, ovf, ovf_flag, s_log2 + 10u, copies_log2});
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (before_sums && (e = hipStreamWaitEvent(stream, before_sums, 0)) != hipSuccess) return e;
    hipLaunchKernelGGL((halo_log_accumulate_kernel<1u>), dim3(tiles * planes), dim3(kBinBlock), 0, stream, plane, 0u, reinterpret_cast<const uint2*>(list2), cap2, cnt2,
                       static_cast<const WlEntryDev*>(nullptr), 0u, tiles_log2, s_log2, frac_bits);
    return hipGetLastError();
  }
  if (interleaved)
    hipLaunchKernelGGL((halo_split_kernel<kLogSplitThreads, kLogSplitPer, 256u, false, true>), dim3(regions), dim3(kLogSplitThreads), 0, stream, plane, reinterpret_cast<const uint2*>(log), cap1, cnt1, 1u,
                       1u, reinterpret_cast<uint2*>(list2), cap2, cnt2, tiles_log2, 0u, 0xFFFFFFFFu, 0u, s_log2, SplitXyz{nullptr, 0u, 0u, ovf, ovf_flag, s_log2 + 10u, copies_log2});
  else
    hipLaunchKernelGGL((halo_split_kernel<kLogSplitThreads, kLogSplitPer, 256u, false, false>), dim3(regions), dim3(kLogSplitThreads), 0, stream, plane, reinterpret_cast<const uint2*>(log), cap1, cnt1, 1u,
                       1u, reinterpret_cast<uint2*>(list2), cap2, cnt2, tiles_log2, 0u, 0xFFFFFFFFu, tile_log2, 0u, SplitXyz{nullptr, 0u, 0u, ovf, ovf_flag, s_log2 + 10u, copies_log2});
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (before_sums && (e = hipStreamWaitEvent(stream, before_sums, 0)) != hipSuccess) return e;
  if (interleaved)
    hipLaunchKernelGGL((halo_log_accumulate_kernel<1u>), dim3(tiles), dim3(kBinBlock), 0, stream, plane, 0u, reinterpret_cast<const uint2*>(list2), cap2, cnt2,
                       static_cast<const WlEntryDev*>(nullptr), 0u, tiles_log2, s_log2, frac_bits);
  else hipLaunchKernelGGL(halo_bin_accumulate_range_kernel, dim3(tiles), dim3(kBinBlock), 0, stream, plane, reinterpret_cast<const uint2*>(list2), cap2, cnt2, tile_log2, frac_bits);
  return hipGetLastError();
}

// the same for an illuminant session on X, Y, Z planes: `tiles` (a power of two <= 512) interleaved tiles of <= 4 Ki slots of one plane
hipError_t launch_log_route_xyz(float* planes, uint32_t plane_stride, const HitRec* log, uint32_t cap1, const uint32_t* cnt1, uint32_t regions, HitRec* list2,
                                uint32_t cap2, uint32_t* cnt2, uint32_t tiles, uint32_t s_log2, const WlEntryDev* pool, uint32_t pool_size, uint32_t frac_bits,
                                double* ovf, uint32_t* ovf_flag, uint32_t copies_log2, hipStream_t stream, hipEvent_t before_sums) {
  const uint32_t tiles_log2 = static_cast<uint32_t>(__builtin_ctz(tiles));
  hipLaunchKernelGGL((halo_split_kernel<kLogSplitThreads, kLogSplitPer, 512u, true, true>), dim3(regions), dim3(kLogSplitThreads), 0, stream, planes, reinterpret_cast<const uint2*>(log), cap1, cnt1, 1u, 1u,
                     reinterpret_cast<uint2*>(list2), cap2, cnt2, tiles_log2, 0u, (1u << kLogWlShift) - 1u, 0u, s_log2, SplitXyz{pool, pool_size, plane_stride, ovf, ovf_flag, s_log2 + 10u, copies_log2});
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (before_sums && (e = hipStreamWaitEvent(stream, before_sums, 0)) != hipSuccess) return e;
  hipLaunchKernelGGL((halo_log_accumulate_kernel<3u>), dim3(tiles), dim3(kBinBlock), 0, stream, planes, plane_stride, reinterpret_cast<const uint2*>(list2), cap2, cnt2,
                     pool, pool_size, tiles_log2, s_log2, frac_bits);
  return hipGetLastError();
}

hipError_t launch_bin_two_level(float* plane, const HitRec* list1, uint32_t cap1, uint32_t* cnt1, uint32_t lists1, HitRec* list2, uint32_t cap2,
                                uint32_t* cnt2, uint32_t tiles, uint32_t fan_log2, uint32_t frac_bits, double* ovf, uint32_t* ovf_flag, hipStream_t stream, hipEvent_t before_sums) {
  hipLaunchKernelGGL((halo_split_kernel<256u, 16u, 256u, false, false>), dim3(lists1 * kSplitParts), dim3(256u), 0, stream, plane, reinterpret_cast<const uint2*>(list1), cap1,
                     cnt1, static_cast<uint32_t>(kBinCntStride), kSplitParts, reinterpret_cast<uint2*>(list2), cap2, cnt2, fan_log2, 1u, 0xFFFFFFFFu,
                     static_cast<uint32_t>(kBinTileLog2), 0u, SplitXyz{nullptr, 0u, 0u, ovf, ovf_flag, 0u, 0u});   // (per-entry planes have no privatised copies: the twin's offsets are the planes')
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (before_sums && (e = hipStreamWaitEvent(stream, before_sums, 0)) != hipSuccess) return e;
  hipLaunchKernelGGL(halo_bin_accumulate_range_kernel, dim3(tiles), dim3(kBinBlock), 0, stream, plane, reinterpret_cast<const uint2*>(list2), cap2, cnt2,
                     static_cast<uint32_t>(kBinTileLog2), frac_bits);
  return hipGetLastError();
}

// xyz[pix] += sum over planes of coef[plane] * (sum over copies of plane[MonoSlot(pix)]); the slots are zeroed — closes a
// session.  Planes: 1 (discrete wavelength, coef = CMF), 3 (X, Y, Z; unit coefs) or one per wavelength-pool entry
// (coef = that entry's CMF), at most kFoldGroup per launch.  Tiled transpose through LDS: a block reads a R-row x
// 64-column tile of every plane copy along the columns (coalesced), then walks it along the rows, where consecutive
// rows are consecutive pixels (coalesced xyz RMW).
// Round 4.  The first form (R = 64 only) ran 436 us on the reference's own benchmark scene (512 x 256, 64 planes): 32 workgroups on
// a 256-CU part, and per thread 16 rows x 64 planes of load -> compare -> store-zero round trips in sequence (the zeroing store may
// alias the next load as far as the compiler knows, so nothing overlapped).  Now (1) the tile height R is 64, 16 or 4 rows, whichever
// gives the launch >= 512 workgroups (the write runs stay >= 48 bytes, and an image that small is cache-resident anyway), and (2) a
// thread requests kFoldBatch plane values before it looks at any of them or zeroes anything.
constexpr uint32_t kFoldTile = 64u;
constexpr uint32_t kFoldBatch = 8u;
template <uint32_t R>
__global__ void __launch_bounds__(kBlock) halo_fold_kernel(float* __restrict__ xyz, float* __restrict__ planes, uint32_t n_pix,
                                                            uint32_t s_log2, uint32_t copies, uint32_t n_planes, const FoldCoef coef,
                                                            double* __restrict__ ovf, const uint32_t* __restrict__ ovf_flag) {
  __shared__ float tile[3][R][kFoldTile + 1];
  // the planes' fp64 twin (copy 0: what log regions and tile lists could not hold, DispatchParams::ovf) comes in only when something was written
  // to it — one uniform load — and `ovf` already points at this group's first plane
  const bool take_ovf = ovf != nullptr && *ovf_flag != 0u;
  const uint32_t tiles_c = (1u << s_log2) / kFoldTile;  // columns per row / tile width (s_log2 >= 6)
  const uint32_t row0 = (blockIdx.x / tiles_c) * R, col0 = (blockIdx.x % tiles_c) * kFoldTile;
  const size_t plane = static_cast<size_t>(kMonoRows) << s_log2;
  const uint32_t lo = threadIdx.x & (kFoldTile - 1u), hi = threadIdx.x / kFoldTile;  // hi in [0, 4)
  for (uint32_t r = hi; r < R; r += kBlock / kFoldTile) {
    float* q = planes + (static_cast<size_t>(row0 + r) << s_log2) + col0 + lo;
    float x = 0.0f, y = 0.0f, z = 0.0f;
    if (copies == 1u) {   // per-entry planes, or a session whose launches all went through the hit log: batches of planes
      for (uint32_t p0 = 0; p0 < n_planes; p0 += kFoldBatch) {
        float t[kFoldBatch];
#pragma unroll
        for (uint32_t u = 0; u < kFoldBatch; ++u) t[u] = p0 + u < n_planes ? q[static_cast<size_t>(p0 + u) * plane] : 0.0f;
        if (take_ovf) {
          double* qo = ovf + (q - planes);
#pragma unroll
          for (uint32_t u = 0; u < kFoldBatch; ++u) {
            if (p0 + u >= n_planes) continue;
            const double o = qo[static_cast<size_t>(p0 + u) * plane];
            if (o != 0.0) {
              qo[static_cast<size_t>(p0 + u) * plane] = 0.0;
              const float sum = static_cast<float>(static_cast<double>(t[u]) + o);
              t[u] = sum;
            }
          }
        }
#pragma unroll
        for (uint32_t u = 0; u < kFoldBatch; ++u) {
          if (t[u] == 0.0f) continue;
          q[static_cast<size_t>(p0 + u) * plane] = 0.0f;
          x += coef.c[p0 + u][0] * t[u];
          y += coef.c[p0 + u][1] * t[u];
          z += coef.c[p0 + u][2] * t[u];
        }
      }
    } else {              // privatised copies of few planes: batches of copies, the plane's coefficient applied to their sum
      for (uint32_t pl = 0; pl < n_planes; ++pl) {
        float v = 0.0f;
        float* qp = q + static_cast<size_t>(pl) * copies * plane;
        for (uint32_t c0 = 0; c0 < copies; c0 += kFoldBatch) {
          float t[kFoldBatch];
#pragma unroll
          for (uint32_t u = 0; u < kFoldBatch; ++u) t[u] = c0 + u < copies ? qp[static_cast<size_t>(c0 + u) * plane] : 0.0f;
#pragma unroll
          for (uint32_t u = 0; u < kFoldBatch; ++u) {
            if (t[u] == 0.0f) continue;
            qp[static_cast<size_t>(c0 + u) * plane] = 0.0f;
            v += t[u];
          }
        }
        if (take_ovf) {   // the twin shadows copy 0
          double* qo = ovf + (q - planes) + static_cast<size_t>(pl) * plane;   // the twin has no copies: plane pl's twin starts at pl * plane
          const double o = *qo;
          if (o != 0.0) {
            *qo = 0.0;
            v = static_cast<float>(static_cast<double>(v) + o);
          }
        }
        x += coef.c[pl][0] * v;
        y += coef.c[pl][1] * v;
        z += coef.c[pl][2] * v;
      }
    }
    tile[0][r][lo] = x;
    tile[1][r][lo] = y;
    tile[2][r][lo] = z;
  }
  __syncthreads();
  const uint32_t s_mask = (1u << s_log2) - 1u;
  const uint32_t wr = threadIdx.x % R;   // row of the tile = consecutive pixels
  for (uint32_t c = threadIdx.x / R; c < kFoldTile; c += kBlock / R) {
    const float x = tile[0][wr][c], y = tile[1][wr][c], z = tile[2][wr][c];
    if (x == 0.0f && y == 0.0f && z == 0.0f) continue;
    const uint32_t a = ((col0 + c) * kMonoMulInv) & s_mask;   // column hash inverted
    const uint32_t pix = a * kMonoRows + row0 + wr;
    if (pix < n_pix) {
      xyz[3u * pix + 0u] += x;
      xyz[3u * pix + 1u] += y;
      xyz[3u * pix + 2u] += z;
    }
  }
}

// Consumer fold: running image += drained accumulator, Neumaier-compensated (accum_shared.h:70-74), accumulator zeroed.
__global__ void __launch_bounds__(kBlock) halo_consumer_fold_kernel(float* __restrict__ acc, float* __restrict__ sum,
                                                                     float* __restrict__ comp, uint32_t n) {
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const float delta = acc[i];
    if (delta == 0.0f) continue;
    acc[i] = 0.0f;
    const float s = sum[i];
    const float ns = s + delta;
    comp[i] += (fabsf(delta) < fabsf(s)) ? ((s - ns) + delta) : ((delta - ns) + s);
    sum[i] = ns;
  }
}

struct DisplayDev {
  float scale;
  float ray_color[3];
  float background[3];
};

// PostSnapshot (render.cpp:508-578): per pixel scale → gamut clip (color_space.cpp GamutClipXyz) → XYZ→linear RGB →
// background + clamp → sRGB gamma → u8.  Also writes the raw snapshot (sum + comp) when xyz_out is set.
__global__ void __launch_bounds__(kBlock) halo_post_snapshot_kernel(const float* __restrict__ sum, const float* __restrict__ comp,
                                                                     uint8_t* __restrict__ rgb_out, float* __restrict__ xyz_out,
                                                                     uint32_t n_pix, DisplayDev dsp) {
  const float kWhite[3] = {0.95047f, 1.00000f, 1.08883f};                       // kWhitePointD65, util/color_data.hpp:6
  const float kM[9] = {3.2404542f, -1.5371385f, -0.4985314f, -0.9692660f, 1.8760108f, 0.0415560f,
                       0.0556434f, -0.2040259f, 1.0572252f};                    // kXyzToRgb, util/color_data.hpp:8-12
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_pix; i += stride) {
    float xyz[3], rgb[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const float raw = sum[3u * i + j] + comp[3u * i + j];
      if (xyz_out) xyz_out[3u * i + j] = raw;
      xyz[j] = raw * dsp.scale;
    }
    if (rgb_out == nullptr) continue;
    if (dsp.ray_color[0] < 0.0f) {
      float gray[3], diff[3];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        gray[j] = kWhite[j] * xyz[1];
        diff[j] = xyz[j] - gray[j];
      }
      float s = 1.0f;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          a += -gray[k] * kM[j * 3 + k];
          b += diff[k] * kM[j * 3 + k];
        }
        if (a * b > 0.0f && a / b < s) s = a / b;
      }
#pragma unroll
      for (int j = 0; j < 3; j++) {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) v += (diff[k] * s + gray[k]) * kM[j * 3 + k];
        rgb[j] = fminf(fmaxf(v, 0.0f), 1.0f);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 3; j++) {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) v += (kWhite[k] * xyz[1]) * kM[j * 3 + k];
        rgb[j] = v * dsp.ray_color[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float v = rgb[j] + dsp.background[j];
      v = fminf(fmaxf(v, 0.0f), 1.0f);
      v = (v < 0.0031308f) ? v * 12.92f : 1.055f * powf(v, 1.0f / 2.4f) - 0.055f;
      rgb_out[3u * i + j] = static_cast<uint8_t>(v * 255.0f);
    }
  }
}

// ---- display-side composite of the raypath-colour class lanes (server/component_compositor.cpp) --------------------------------
// The lanes are fp64 on the device and are handed out as floats (halo_readback_class_lanes); the composite reads them the same way.

// One pass of the radix select behind ComputeParticipatingP99Y (component_compositor.cpp:138-163): a histogram of `bits` bits at `shift`
// of the float bit patterns of the positive lane values whose higher bits equal `prefix` (positive floats order like their patterns).
// shift + bits == 32 on the first pass (no prefix).
__global__ void __launch_bounds__(kBlock) halo_lane_hist_kernel(const double* __restrict__ lanes, uint32_t n_pix, CompositeDev cd, uint32_t shift, uint32_t bits,
                                                                 uint32_t prefix, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[2048];
  const uint32_t nb = 1u << bits;
  for (uint32_t i = threadIdx.x; i < nb; i += kBlock) h[i] = 0u;
  __syncthreads();
  const uint64_t total = static_cast<uint64_t>(n_pix) * cd.n_active;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
  const uint32_t hi = shift + bits;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += stride) {
    const uint32_t k = static_cast<uint32_t>(i / n_pix), p = static_cast<uint32_t>(i - static_cast<uint64_t>(k) * n_pix);
    const float v = static_cast<float>(lanes[static_cast<size_t>(cd.lane[k]) * n_pix + p]);
    if (!(v > 0.0f)) continue;
    const uint32_t u = __float_as_uint(v);
    if (hi < 32u && (u >> hi) != prefix) continue;
    atomicAdd(&h[(u >> shift) & (nb - 1u)], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < nb; i += kBlock)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

// CompositeDominantPixel / CompositeAdditivePixel / CompositePainterPixel (component_compositor.cpp:56-114) + the painter's post-multiply and
// clamp (:268-274) + LinearRgbToSrgbU8 (:292-300).  Every product and sum is rounded on its own (mul_rn / add_rn: the backend fuses plain a * b + c whatever the source says), like the
// reference's host build.
__global__ void __launch_bounds__(kBlock) halo_composite_kernel(const double* __restrict__ lanes, uint32_t n_pix, CompositeDev cd, float* __restrict__ rgb_out,
                                                                 uint8_t* __restrict__ srgb_out) {
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n_pix; p += stride) {
    float out[3] = {0.0f, 0.0f, 0.0f};
    if (cd.mode == HALO_COMPOSITE_DOMINANT) {
      int best = -1;
      float best_ey = 0.0f;
      for (uint32_t k = 0; k < cd.n_active; k++) {
        const float ey = mul_rn(static_cast<float>(lanes[static_cast<size_t>(cd.lane[k]) * n_pix + p]), cd.s);
        if (ey > best_ey) {
          best_ey = ey;
          best = static_cast<int>(k);
        }
      }
      if (best >= 0)
        for (int j = 0; j < 3; j++) out[j] = mul_rn(cd.color[best][j], best_ey);
    } else if (cd.mode == HALO_COMPOSITE_ADDITIVE) {
      for (uint32_t k = 0; k < cd.n_active; k++) {
        const float ey = mul_rn(static_cast<float>(lanes[static_cast<size_t>(cd.lane[k]) * n_pix + p]), cd.s);
        if (!(ey > 0.0f)) continue;
        for (int j = 0; j < 3; j++) out[j] = add_rn(out[j], mul_rn(cd.color[k][j], ey));
      }
      for (int j = 0; j < 3; j++) out[j] = fminf(fmaxf(out[j], 0.0f), 1.0f);
    } else {
      float T = 1.0f;
      for (uint32_t k = 0; k < cd.n_active && T > 0.0f; k++) {
        const float alpha = fminf(mul_rn(static_cast<float>(lanes[static_cast<size_t>(cd.lane[k]) * n_pix + p]), cd.a), 1.0f);
        if (!(alpha > 0.0f)) continue;
        const float ta = mul_rn(T, alpha);
        for (int j = 0; j < 3; j++) out[j] = add_rn(out[j], mul_rn(ta, cd.color[k][j]));
        T = mul_rn(T, add_rn(1.0f, -alpha));
      }
      for (int j = 0; j < 3; j++) out[j] = fminf(fmaxf(mul_rn(out[j], cd.display), 0.0f), 1.0f);
    }
    if (rgb_out)
      for (int j = 0; j < 3; j++) rgb_out[3u * p + j] = out[j];
    if (srgb_out)
      for (int j = 0; j < 3; j++) {
        const float v = fminf(fmaxf(out[j], 0.0f), 1.0f);
        const float g = (v < 0.0031308f) ? mul_rn(v, 12.92f) : add_rn(mul_rn(1.055f, powf(v, 1.0f / 2.4f)), -0.055f);   // LinearToSrgb color_space.cpp:47-52
        srgb_out[3u * p + j] = static_cast<uint8_t>(mul_rn(g, 255.0f));
      }
  }
}

__global__ void __launch_bounds__(kBlock) halo_lanes_load_kernel(const float* __restrict__ src, double* __restrict__ lanes, uint64_t n) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) lanes[i] = static_cast<double>(src[i]);
}

// ConsumeDeviceFused's lane half: lanes drained elsewhere are ADDED (render.cpp:176-185)
__global__ void __launch_bounds__(kBlock) halo_lanes_add_kernel(const float* __restrict__ src, double* __restrict__ lanes, uint64_t n) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) lanes[i] += static_cast<double>(src[i]);
}
// the inverse at the seam (TraceBackend::ReadbackClassLanes hands out floats and drains the device lanes): narrow into the staging buffer, zero the lane
__global__ void __launch_bounds__(kBlock) halo_lanes_drain_kernel(double* __restrict__ lanes, float* __restrict__ dst, uint64_t n) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    dst[i] = static_cast<float>(lanes[i]);
    lanes[i] = 0.0;
  }
}

hipError_t launch_lane_hist(const double* lanes, uint32_t n_pix, const CompositeDev& cd, uint32_t shift, uint32_t bits, uint32_t prefix, uint32_t* hist, int blocks,
                            hipStream_t stream) {
  hipLaunchKernelGGL(halo_lane_hist_kernel, dim3(blocks), dim3(kBlock), 0, stream, lanes, n_pix, cd, shift, bits, prefix, hist);
  return hipGetLastError();
}
hipError_t launch_composite(const double* lanes, uint32_t n_pix, const CompositeDev& cd, float* rgb_out, uint8_t* srgb_out, int blocks, hipStream_t stream) {
  hipLaunchKernelGGL(halo_composite_kernel, dim3(blocks), dim3(kBlock), 0, stream, lanes, n_pix, cd, rgb_out, srgb_out);
  return hipGetLastError();
}
hipError_t launch_lanes_load(const float* src, double* lanes, uint64_t n, int blocks, hipStream_t stream) {
  hipLaunchKernelGGL(halo_lanes_load_kernel, dim3(blocks), dim3(kBlock), 0, stream, src, lanes, n);
  return hipGetLastError();
}

hipError_t launch_lanes_add(const float* src, double* lanes, uint64_t n, int blocks, hipStream_t stream) {
  hipLaunchKernelGGL(halo_lanes_add_kernel, dim3(blocks), dim3(kBlock), 0, stream, src, lanes, n);
  return hipGetLastError();
}
hipError_t launch_lanes_drain(double* lanes, float* dst, uint64_t n, int blocks, hipStream_t stream) {
  hipLaunchKernelGGL(halo_lanes_drain_kernel, dim3(blocks), dim3(kBlock), 0, stream, lanes, dst, n);
  return hipGetLastError();
}

hipError_t launch_consumer_fold(float* acc, float* sum, float* comp, uint32_t n, int blocks, hipStream_t stream) {
  hipLaunchKernelGGL(halo_consumer_fold_kernel, dim3(blocks), dim3(kBlock), 0, stream, acc, sum, comp, n);
  return hipGetLastError();
}

hipError_t launch_post_snapshot(const float* sum, const float* comp, uint8_t* rgb_out, float* xyz_out, uint32_t n_pix, float scale,
                                const float ray_color[3], const float background[3], int blocks, hipStream_t stream) {
  DisplayDev d;
  d.scale = scale;
  for (int j = 0; j < 3; j++) {
    d.ray_color[j] = ray_color[j];
    d.background[j] = background[j];
  }
  hipLaunchKernelGGL(halo_post_snapshot_kernel, dim3(blocks), dim3(kBlock), 0, stream, sum, comp, rgb_out, xyz_out, n_pix, d);
  return hipGetLastError();
}

// `planes` points at the first plane of this group; coef holds n_planes (<= kFoldGroup) coefficient triples
hipError_t launch_fold(float* xyz, float* planes, uint32_t n_pix, uint32_t s_log2, uint32_t copies, uint32_t n_planes, const FoldCoef& coef,
                       double* ovf, const uint32_t* ovf_flag, hipStream_t stream) {
  const uint32_t tiles_c = (1u << s_log2) / kFoldTile;
  if (tiles_c * (kMonoRows / 64u) >= 512u)
    hipLaunchKernelGGL(halo_fold_kernel<64u>, dim3(tiles_c * (kMonoRows / 64u)), dim3(kBlock), 0, stream, xyz, planes, n_pix, s_log2, copies, n_planes, coef, ovf, ovf_flag);
  else if (tiles_c * (kMonoRows / 16u) >= 512u)
    hipLaunchKernelGGL(halo_fold_kernel<16u>, dim3(tiles_c * (kMonoRows / 16u)), dim3(kBlock), 0, stream, xyz, planes, n_pix, s_log2, copies, n_planes, coef, ovf, ovf_flag);
  else
    hipLaunchKernelGGL(halo_fold_kernel<4u>, dim3(tiles_c * (kMonoRows / 4u)), dim3(kBlock), 0, stream, xyz, planes, n_pix, s_log2, copies, n_planes, coef, ovf, ovf_flag);
  return hipGetLastError();
}

hipError_t launch_trace_m0(const DispatchParams& P, int blocks, hipStream_t stream, int geom, bool mono);
hipError_t launch_trace_m1(const DispatchParams& P, int blocks, hipStream_t stream, int geom, bool mono);
hipError_t launch_trace_m2(const DispatchParams& P, int blocks, hipStream_t stream, int geom, bool mono);
hipError_t launch_trace_m3(const DispatchParams& P, int blocks, hipStream_t stream, int geom, bool mono);
hipError_t launch_trace_m4(const DispatchParams& P, int blocks, hipStream_t stream, int geom, bool mono);

// mode: halo_trace.inl kMode* — 0 plain, 1 filter (fast form), 2 capture (tests), 3 generic filter / colour, 4 filter + colour (fast form)
// geom: 0 = one shape per dispatch, 1 = shape pool, 2 = shape pool of prisms (compact LDS slots), 3 = one regular hexagonal prism
hipError_t launch_trace(const DispatchParams& P, int blocks, hipStream_t stream, int mode, int geom, bool mono) {
  switch (mode) {
    case 0: return launch_trace_m0(P, blocks, stream, geom, mono);
    case 1: return launch_trace_m1(P, blocks, stream, geom, mono);
    case 2: return launch_trace_m2(P, blocks, stream, geom, mono);
    case 3: return launch_trace_m3(P, blocks, stream, geom, mono);
    case 4: return launch_trace_m4(P, blocks, stream, geom, mono);
  }
  return hipErrorInvalidValue;
}

}  // namespace halo
