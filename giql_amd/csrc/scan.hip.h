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
template <int ITEMS>
__device__ __forceinline__ void scan_load(const u32* __restrict__ in, u64 n, u64 base, u32 (&x)[ITEMS]) {
  if (base + ITEMS <= n && ((base & 3) == 0)) {
    const uint4* p = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
    for (int k = 0; k < ITEMS / 4; k++) {
      uint4 t = p[k];
      x[4 * k + 0] = t.x;
      x[4 * k + 1] = t.y;
      x[4 * k + 2] = t.z;
      x[4 * k + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < ITEMS; k++) x[k] = (base + k < n) ? in[base + k] : 0u;
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
template <int ITEMS>
__device__ __forceinline__ void scan_load_diff(const u32* __restrict__ hi, const u32* __restrict__ lo, u64 n,
                                               u64 n_reg, u64 base, u32 (&x)[ITEMS]) {
  u32 l[ITEMS];
  scan_load(hi, n, base, x);
  scan_load(lo, n, base, l);
#pragma unroll
  for (int k = 0; k < ITEMS; k++) x[k] = (base + k < n_reg) ? x[k] - l[k] : 0u;
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

// ---- one launch instead of three: chained scan over two bounds per row, with the fill's merge-path
// partition written on the way (round 3).  Tiles are taken from an atomic ticket, so a tile's
// predecessors have all started before it: the decoupled look-back cannot wait for a block that is
// not running.  One 64-bit status word per tile {flag:2 | value:62} (single-word hand-off: relaxed
// agent-scope store / loads, no fence), zeroed together with the ticket by whoever runs before
// (k_bucket_bounds_fused).  Wave 0 looks back 64 tiles per round.
//   off[q]  = exclusive sum of cnt = hi - lo (rows below the regular prefix, 0 past it); off[n] = total
//   part[t] = the row whose outputs cover output t * tile (k_partition's answer), t * tile < total,
//             t < part_cap; part[ceil(total / tile)] = n - 1      (part == nullptr: not written)
constexpr u64 SC_FLAG_AGG = 1ull << 62;
constexpr u64 SC_FLAG_PREFIX = 2ull << 62;
constexpr u64 SC_VALUE_MASK = (1ull << 62) - 1ull;
// Every tile of a grid that starts at once publishes an aggregate at about the same time, and the prefixes
// then advance one look-back window per round trip: a 64-tile window took the 2442 tiles of a 10M-row scan
// 38 round trips (75 us).  So each lane polls SC_LB_GROUPS words per round (a 256-tile window: 10 round trips).
constexpr int SC_NT = 1024;
constexpr int SC_ITEMS = 8;
constexpr int SC_LB_GROUPS = 2;
constexpr int SC_TILE = SC_NT * SC_ITEMS;

// The rows of one thread whose output ranges contain the first output of a fill tile (a rolled loop that reads
// its counts again: one thread in ~25 comes here, and the hot path keeps its registers).
__device__ __forceinline__ void scan_emit_part(const u32* __restrict__ hi, const u32* __restrict__ lo, u64 n, u64 n_reg,
                                            u64 base, u64 o, u32* __restrict__ part, u32 tile_log2, u32 part_cap) {
  const u64 tmask = (1ull << tile_log2) - 1ull;
#pragma unroll 1
  for (int k = 0; k < SC_ITEMS && base + k < n; k++) {
    const u32 xk = base + k < n_reg ? hi[base + k] - lo[base + k] : 0u;
    if (xk != 0u) {
      u64 t0 = (o + tmask) >> tile_log2;
      const u64 end = o + xk;
      while ((t0 << tile_log2) < end) {
        if (t0 < (u64)part_cap) part[t0] = (u32)(base + k);
        t0++;
      }
    }
    o += xk;
  }
}

__global__ __launch_bounds__(SC_NT) void k_scan_chain_diff(const u32* __restrict__ hi, const u32* __restrict__ lo,
                                                            u32 n, const u32* __restrict__ n_irr,
                                                            u64* __restrict__ status, u32* __restrict__ ticket,
                                                            u64* __restrict__ off, u64* __restrict__ total_out2,
                                                            u32* __restrict__ part, u32 tile_log2, u32 part_cap) {
  __shared__ u64 lds[SC_NT / WAVE + 1];
  __shared__ u32 s_tile;
  __shared__ u64 s_excl;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  const u32 tile = s_tile;
  const u32 n_tiles = (n + SC_TILE - 1) / SC_TILE;
  if (tile >= n_tiles) return;  // block-uniform
  const u64 base = (u64)tile * SC_TILE + (u64)threadIdx.x * SC_ITEMS;
  const u64 n_reg = (u64)n - *n_irr;
  u32 x[SC_ITEMS];
  scan_load_diff(hi, lo, n, n_reg, base, x);
  u64 s = 0;
#pragma unroll
  for (int k = 0; k < SC_ITEMS; k++) s += x[k];
  u64 total;
  const u64 in_block = block_excl_scan<u64, SC_NT>(s, lds, total);
  if (threadIdx.x == 0) {
    __hip_atomic_store(status + tile, (tile == 0 ? SC_FLAG_PREFIX : SC_FLAG_AGG) | total, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    if (tile == 0) s_excl = 0;
  }
  if (tile != 0 && wave_id() == 0) {
    const u32 lane = lane_id();
    u64 excl = 0;
    u32 t = tile;  // predecessors t-1, t-2, ...
    bool done = false;
    while (!done) {
      u64 v[SC_LB_GROUPS];
#pragma unroll
      for (int g = 0; g < SC_LB_GROUPS; g++) {
        const u32 back = (u32)g * WAVE + lane;  // predecessor t - 1 - back
        v[g] = SC_FLAG_PREFIX;                  // past tile 0: ends the walk, adds nothing
        if (t > back) v[g] = __hip_atomic_load(status + (t - 1u - back), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      u32 consumed = 0;
#pragma unroll
      for (int g = 0; g < SC_LB_GROUPS; g++) {
        if (done || consumed != (u32)g * WAVE) continue;  // wave-uniform: a group counts only behind full ones
        const u32 f = (u32)(v[g] >> 62);
        const u64 ready = __ballot(f != 0u);
        const u64 pref = __ballot(f == 2u);
        // usable lanes: the contiguous ready ones from lane 0, up to and including the first PREFIX
        const u32 n_ready = ~ready == 0ull ? 64u : (u32)__builtin_ctzll(~ready);
        const u32 first_pref = pref == 0ull ? 64u : (u32)__builtin_ctzll(pref);
        const u32 use = first_pref < n_ready ? first_pref + 1u : n_ready;
        u64 add = lane < use ? (v[g] & SC_VALUE_MASK) : 0ull;
        excl += wave_reduce_sum(add);
        consumed += use;
        if (first_pref < n_ready) done = true;
      }
      t -= consumed;
      if (!done && consumed == 0u) __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) {
      s_excl = excl;
      __hip_atomic_store(status + tile, SC_FLAG_PREFIX | ((excl + total) & SC_VALUE_MASK), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  const u64 run0 = s_excl + in_block;
  u64 run = run0;
  const u64 tmask = (1ull << tile_log2) - 1ull;
  if (base + SC_ITEMS <= n) {
    // 16-byte stores, two offsets each (the arena keeps `off` 256-byte aligned, base % 16 == 0); no array of
    // sixteen 64-bit offsets is kept (32 VGPRs: half the occupancy)
    uint4* dst = reinterpret_cast<uint4*>(off + base);
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k += 2) {
      const u64 a = run;
      const u64 b2 = run + x[k];
      run = b2 + x[k + 1];
      dst[k / 2] = make_uint4((u32)a, (u32)(a >> 32), (u32)b2, (u32)(b2 >> 32));
    }
  } else {
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
      if (base + k < n) off[base + k] = run;
      run += x[k];
    }
  }
  if (part) {
    // fill tiles whose first output lies in this thread's range [run0, run): rare (one tile per ~25 threads at
    // 40 outputs per row), so the per-row walk only runs for the threads that hold one
    const u64 t_first = (run0 + tmask) >> tile_log2;
    if ((t_first << tile_log2) < run) scan_emit_part(hi, lo, n, n_reg, base, run0, part, tile_log2, part_cap);
  }
  if (tile == n_tiles - 1 && threadIdx.x == SC_NT - 1) {  // this thread's `run` is the grand total
    off[n] = run;
    if (total_out2) *total_out2 = run;
    if (part) {
      const u64 nt = (run + tmask) >> tile_log2;
      if (nt < (u64)part_cap) part[nt] = n > 0 ? n - 1u : 0u;
    }
  }
}

// The down-sweep of a scan of 0 / 1 flags with the compaction folded in: rows_out[exclusive offset] = row for the
// flagged rows (ascending row ids) -- SEMI / ANTI's last two launches (offsets written, then read by a compact
// kernel) in one, the offsets never stored.
__global__ __launch_bounds__(SCAN_NT) void k_scan_down_compact(const u32* __restrict__ flag, u64 n,
                                                                const u64* __restrict__ bsums,
                                                                int32_t* __restrict__ rows_out) {
  __shared__ u64 lds[SCAN_NT / WAVE + 1];
  const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  u32 x[SCAN_ITEMS];
  scan_load(flag, n, base, x);
  u64 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) s += x[k];
  u64 total;
  u64 run = bsums[blockIdx.x] + block_excl_scan<u64, SCAN_NT>(s, lds, total);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    if (x[k]) rows_out[run] = (int32_t)(base + k);
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
