// giql_amd/csrc/scan.hip.h -- device-wide exclusive scan of u32 counts
// (reduce -> spine -> downsweep).  HBM-bound: reads the input twice, writes the
// output once.  Used for (a) radix tile-histogram offsets (u32 out) and (b) pair
// offsets (u64 out: P can exceed 2^32).
#pragma once

#include "dev_common.hip.h"

namespace giql {

constexpr int SCAN_NT = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_NT * SCAN_ITEMS;  // 4096 inputs per block

// Per-thread blocked load of SCAN_ITEMS consecutive u32 (4 x dwordx4).
__device__ __forceinline__ void scan_load(const u32* __restrict__ in, u64 n, u64 base, u32 (&x)[SCAN_ITEMS]) {
  if (base + SCAN_ITEMS <= n && ((base & 3) == 0)) {
    const uint4* p = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS / 4; k++) {
      uint4 t = p[k];
      x[4 * k + 0] = t.x;
      x[4 * k + 1] = t.y;
      x[4 * k + 2] = t.z;
      x[4 * k + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) x[k] = (base + k < n) ? in[base + k] : 0u;
  }
}

__global__ __launch_bounds__(SCAN_NT) void k_scan_reduce(const u32* __restrict__ in, u64 n,
                                                          u64* __restrict__ bsums) {
  __shared__ u64 lds[SCAN_NT / WAVE + 1];
  const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  u32 x[SCAN_ITEMS];
  scan_load(in, n, base, x);
  u64 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) s += x[k];
  s = wave_reduce_sum(s);
  if (lane_id() == 0) lds[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
#pragma unroll
    for (int w = 0; w < SCAN_NT / WAVE; w++) t += lds[w];
    bsums[blockIdx.x] = t;
  }
}

// Single block: in-place exclusive scan of the block sums; total -> *total_out
// (and, when given, into a DevMeta-style u64 slot).  Each thread owns SPINE_ITEMS consecutive
// sums per round (8192 per round: the 48K block totals of a 100M-row class-1 count are 6 rounds,
// 73 -> ~12 us against one sum per thread per round).
constexpr int SPINE_ITEMS = 8;
__global__ __launch_bounds__(1024) void k_scan_spine(u64* __restrict__ bsums, u32 nb,
                                                      u64* __restrict__ total_out,
                                                      u64* __restrict__ total_out2 = nullptr) {
  __shared__ u64 lds[1024 / WAVE + 1];
  u64 carry = 0;  // block-uniform: every thread adds the same round totals
  for (u32 base = 0; base < nb; base += 1024 * SPINE_ITEMS) {
    const u32 i0 = base + threadIdx.x * SPINE_ITEMS;
    u64 v[SPINE_ITEMS];
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < SPINE_ITEMS; k++) {
      v[k] = i0 + k < nb ? bsums[i0 + k] : 0;
      mine += v[k];
    }
    u64 total;
    u64 run = carry + block_excl_scan<u64, 1024>(mine, lds, total);
#pragma unroll
    for (int k = 0; k < SPINE_ITEMS; k++) {
      if (i0 + k < nb) bsums[i0 + k] = run;
      run += v[k];
    }
    carry += total;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
  if (threadIdx.x == 0 && total_out2) *total_out2 = carry;  // e.g. DevMeta::n_out: no device-to-device copy after the scan
}

// The same scan over counts given as TWO bounds per row (the fused range count of bucket_sort.hip.h writes
// `lo` and `hi` from different blocks): cnt = hi - lo for the rows below *n_total - *n_irr (the regular
// prefix of the sorted queries), 0 past it.  The down-sweep leaves cnt in the hi array (kept for
// giql_hip_inner_plan_export_dev) and zeroes lo past the prefix.
__device__ __forceinline__ void scan_load_diff(const u32* __restrict__ hi, const u32* __restrict__ lo, u64 n,
                                               u64 n_reg, u64 base, u32 (&x)[SCAN_ITEMS]) {
  u32 l[SCAN_ITEMS];
  scan_load(hi, n, base, x);
  scan_load(lo, n, base, l);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) x[k] = (base + k < n_reg) ? x[k] - l[k] : 0u;
}

__global__ __launch_bounds__(SCAN_NT) void k_scan_reduce_diff(const u32* __restrict__ hi,
                                                               const u32* __restrict__ lo, u64 n,
                                                               const u32* __restrict__ n_irr,
                                                               u64* __restrict__ bsums) {
  __shared__ u64 lds[SCAN_NT / WAVE + 1];
  const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  u32 x[SCAN_ITEMS];
  scan_load_diff(hi, lo, n, n - *n_irr, base, x);
  u64 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) s += x[k];
  s = wave_reduce_sum(s);
  if (lane_id() == 0) lds[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
#pragma unroll
    for (int w = 0; w < SCAN_NT / WAVE; w++) t += lds[w];
    bsums[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(SCAN_NT) void k_scan_down_diff(u32* __restrict__ hi, u32* __restrict__ lo, u64 n,
                                                             const u32* __restrict__ n_irr,
                                                             const u64* __restrict__ bsums,
                                                             u64* __restrict__ out) {
  __shared__ u64 lds[SCAN_NT / WAVE + 1];
  const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  const u64 n_reg = n - *n_irr;
  u32 x[SCAN_ITEMS];
  scan_load_diff(hi, lo, n, n_reg, base, x);
  u64 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) s += x[k];
  u64 total;
  u64 run = bsums[blockIdx.x] + block_excl_scan<u64, SCAN_NT>(s, lds, total);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    if (base + k < n) {
      out[base + k] = run;
      hi[base + k] = x[k];
      if (base + k >= n_reg) lo[base + k] = 0u;
    }
    run += x[k];
  }
}

template <typename TOut>
__global__ __launch_bounds__(SCAN_NT) void k_scan_down(const u32* __restrict__ in, u64 n,
                                                        const u64* __restrict__ bsums,
                                                        TOut* __restrict__ out) {
  __shared__ u64 lds[SCAN_NT / WAVE + 1];
  const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  u32 x[SCAN_ITEMS];
  scan_load(in, n, base, x);
  u64 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) s += x[k];
  u64 total;
  u64 run = bsums[blockIdx.x] + block_excl_scan<u64, SCAN_NT>(s, lds, total);
  TOut o[SCAN_ITEMS];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    o[k] = (TOut)run;
    run += x[k];
  }
  if (base + SCAN_ITEMS <= n) {
    // 16-byte stores (the arena keeps `out` 256-byte aligned, base % 16 == 0)
    constexpr int NV = (int)(sizeof(TOut) * SCAN_ITEMS / 16);
    uint4* dst = reinterpret_cast<uint4*>(out + base);
    const uint4* src = reinterpret_cast<const uint4*>(o);
#pragma unroll
    for (int k = 0; k < NV; k++) dst[k] = src[k];
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
      if (base + k < n) out[base + k] = o[k];
  }
}

}  // namespace giql
