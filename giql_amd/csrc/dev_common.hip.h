// giql_amd/csrc/dev_common.hip.h -- device helpers shared by every kernel.
// gfx950 (MI355X) only: 64-wide wavefronts are assumed throughout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace giql {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

constexpr int WAVE = 64;
constexpr u32 U32_MAX = 0xFFFFFFFFu;

// Device-side bookkeeping shared by the kernels of one call.  A pinned host
// mirror is copied back once per call (the only host sync of a join).
struct DevMeta {
  u64 total_span;   // linearised coordinate span (sum of per-chrom spans)
  u64 n_out;        // regular-path output count (pairs), written by the scan
  u64 n_out_irr;    // irregular-path output count
  u64 n_out_c1;     // class-1 share of n_out (first outputs of the pair arrays)
  u32 sentinel;     // key given to irregular rows: sorts after every real key
  u32 irr_a;        // rows of A with canonical end <= start
  u32 irr_b;
  int status;       // 0 ok, else a GIQL_ERR_* code
  u32 aux0, aux1;   // per-operator scratch (e.g. compaction counter)
  // canonical length (end - start) range over the well-formed rows of each side;
  // min == max means "uniform length" (fixed-length reads): see k_range_count
  int len_min_a, len_max_a, len_min_b, len_max_b;
  // 1 when k_chrom_offsets laid the chromosomes out on 2^24-aligned bases (the
  // histogram-in-the-span-pass form, see k_chrom_minmax<true>); 0 = tight packing
  u32 aligned_ok;
  // 1 as soon as a row of the side is found out of (chrom id, start) order or irregular (k_chrom_minmax): a side
  // that stays 0 arrives sorted on the linear axis and needs no sort (coordinate-sorted BED / BAM-derived tables)
  u32 unsorted_a, unsorted_b;
  // 1 as soon as a row of the side has its canonical end BELOW its canonical start (k_chrom_minmax): NEAREST rejects
  // such a table (zero-length rows are fine there)
  u32 inverted_a, inverted_b;
};

// XCD-aware block -> tile map.  Workgroups are dealt round-robin to the 8 XCDs, each
// with its own L2: inside groups of 8 G blocks, block 8j + x takes tile
// ((j / G) * 8 + x) * G + j % G, so G consecutive tiles run on one XCD at about the
// same time and what neighbouring tiles share (adjacent output runs, overlapping input
// windows) meets in one L2.  A bijection on [0, n_blocks); the tail keeps its order.
constexpr u32 XCD_GROUPS = 8;
constexpr u32 XCD_GROUP = 8;
template <u32 G = XCD_GROUP>
__device__ __forceinline__ u32 xcd_tile_of_block(u32 b, u32 n_blocks) {
  constexpr u32 SPAN = XCD_GROUPS * G;
  if (b >= (n_blocks / SPAN) * SPAN) return b;
  const u32 x = b % XCD_GROUPS, j = b / XCD_GROUPS;
  return ((j / G) * XCD_GROUPS + x) * G + j % G;
}

// A load of bytes that are read ONCE (a column streamed through a kernel): non-temporal, so the line is not
// kept in L2 / the Infinity Cache at the expense of what the kernel writes or re-reads.  Measured on MI355X with the
// library's own probe (giql_hip_stream_probe_dev, profiles/r04a_stream_probe.log): a read-only sweep runs at
// 6.2-6.4 TB/s with default-policy loads and 7.0-7.15 TB/s with nt loads; a copy at 5.3 vs 5.5-5.65.
// -DGIQL_NO_NT_LOADS=1 builds the default-policy variant (A/B aid).
template <typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
#if defined(GIQL_NO_NT_LOADS)
  return *p;
#else
  return __builtin_nontemporal_load(p);
#endif
}

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ u32 wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }


template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
  return v;
}

__device__ __forceinline__ u32 wave_reduce_max_u32(u32 v) {
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    u32 t = __shfl_xor(v, d, WAVE);
    v = t > v ? t : v;
  }
  return v;
}

// DPP controls (GCN wave64): shift right inside a 16-lane row, and the two
// cross-row broadcasts that finish a 64-lane scan.  bound_ctrl = true makes lanes
// with no source read 0, the identity of unsigned max and of +.
#define GIQL_DPP_ROW_SHR(n) (0x110 + (n))
#define GIQL_DPP_ROW_BCAST15 0x142
#define GIQL_DPP_ROW_BCAST31 0x143

// 64-lane inclusive max-scan in 6 VALU+DPP instructions (a __shfl_up scan costs
// 6 dependent ds_bpermute round trips through the LDS crossbar).
__device__ __forceinline__ u32 wave_incl_scan_max_u32(u32 v) {
  u32 t;
  t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, GIQL_DPP_ROW_SHR(1), 0xF, 0xF, true);
  v = t > v ? t : v;
  t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, GIQL_DPP_ROW_SHR(2), 0xF, 0xF, true);
  v = t > v ? t : v;
  t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, GIQL_DPP_ROW_SHR(4), 0xF, 0xF, true);
  v = t > v ? t : v;
  t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, GIQL_DPP_ROW_SHR(8), 0xF, 0xF, true);
  v = t > v ? t : v;
  // lane 15 of row r -> all lanes of row r+1 (rows 1 and 3), then lane 31 -> rows 2,3
  t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, GIQL_DPP_ROW_BCAST15, 0xA, 0xF, false);
  v = t > v ? t : v;
  t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, GIQL_DPP_ROW_BCAST31, 0xC, 0xF, false);
  v = t > v ? t : v;
  return v;
}

// Same structure for + on u32.
__device__ __forceinline__ u32 wave_incl_scan_add_u32(u32 v) {
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, GIQL_DPP_ROW_SHR(1), 0xF, 0xF, true);
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, GIQL_DPP_ROW_SHR(2), 0xF, 0xF, true);
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, GIQL_DPP_ROW_SHR(4), 0xF, 0xF, true);
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, GIQL_DPP_ROW_SHR(8), 0xF, 0xF, true);
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, GIQL_DPP_ROW_BCAST15, 0xA, 0xF, false);
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, GIQL_DPP_ROW_BCAST31, 0xC, 0xF, false);
  return v;
}

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    T t = __shfl_up(v, d, WAVE);
    if ((int)lane_id() >= d) v += t;
  }
  return v;
}
template <>
__device__ __forceinline__ u32 wave_incl_scan<u32>(u32 v) {
  return wave_incl_scan_add_u32(v);
}

// Exclusive scan over the NT threads of a block.  `lds` holds NT/64 + 1 items.
// Returns the exclusive prefix of v; `total` receives the block sum.
template <typename T, int NT>
__device__ __forceinline__ T block_excl_scan(T v, T* lds, T& total) {
  constexpr int NW = NT / WAVE;
  T incl = wave_incl_scan(v);
  if (lane_id() == WAVE - 1) lds[wave_id()] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    T run = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      T t = lds[w];
      lds[w] = run;
      run += t;
    }
    lds[NW] = run;
  }
  __syncthreads();
  T base = lds[wave_id()];
  total = lds[NW];
  __syncthreads();  // lds may be reused by the caller
  return base + incl - v;
}

// first index in [lo, hi) with v[idx] >= x   (v ascending)
__device__ __forceinline__ u32 lower_bound_u32(const u32* __restrict__ v, u32 lo, u32 hi, u32 x) {
  while (lo < hi) {
    u32 mid = lo + ((hi - lo) >> 1);
    if (v[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// first index in [lo, hi) with v[idx] > x   (v ascending)
__device__ __forceinline__ u32 upper_bound_u32(const u32* __restrict__ v, u32 lo, u32 hi, u32 x) {
  while (lo < hi) {
    u32 mid = lo + ((hi - lo) >> 1);
    if (v[mid] <= x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

__device__ __forceinline__ u64 upper_bound_u64(const u64* __restrict__ v, u64 lo, u64 hi, u64 x) {
  while (lo < hi) {
    u64 mid = lo + ((hi - lo) >> 1);
    if (v[mid] <= x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// Cooperative lower_bound by one full wave: 64-ary search, every lane returns
// the same result.  ~log64(n) rounds instead of log2(n) dependent loads.
__device__ __forceinline__ u32 wave_lower_bound_u32(const u32* __restrict__ v, u32 lo, u32 hi, u32 x) {
  const u32 lane = lane_id();
  while (hi - lo > (u32)WAVE) {
    const u32 len = hi - lo;
    const u32 step = (len + WAVE - 1) / WAVE;  // >= 2
    const u64 pos64 = (u64)lo + (u64)lane * step;
    const bool in = pos64 < hi;
    const u32 val = in ? v[(u32)pos64] : U32_MAX;
    const u64 m = __ballot(in && val < x);  // prefix-true because v is sorted
    const u32 k = __popcll(m);              // probes strictly below x
    // answer lies in (probe[k-1], probe[k]]
    const u32 nlo = k == 0 ? lo : lo + (k - 1) * step + 1;
    u64 nhi64 = (u64)lo + (u64)k * step;
    if (k == WAVE || nhi64 > hi) nhi64 = hi;
    lo = nlo;
    hi = (u32)nhi64;
  }
  const u32 pos = lo + lane;
  const bool below = pos < hi && v[pos] < x;
  return lo + __popcll(__ballot(below));
}

}  // namespace giql
