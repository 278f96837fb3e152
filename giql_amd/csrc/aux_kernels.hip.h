// giql_amd/csrc/aux_kernels.hip.h -- SEMI / ANTI existence, per-row COUNT and
// NEAREST k=1 on top of the sorted linearised arrays.
//
//   EXISTS b: b.start < a.end AND b.end > a.start
//        <=>  max{ b.end : b.start < a.end } > a.start
// so one prefix-max over B (sorted by start) plus one binary search per A row
// decides SEMI / ANTI with no pair materialisation -- exact for every row,
// including zero-length / inverted ones (no start<end assumption is used).
#pragma once

#include "dev_common.hip.h"
#include "join_kernels.hip.h"

namespace giql {

// ------------------------------------------------------ inclusive prefix max
constexpr int PM_NT = 256;
constexpr int PM_ITEMS = 16;
constexpr int PM_TILE = PM_NT * PM_ITEMS;

__device__ __forceinline__ u32 umax(u32 a, u32 b) { return a > b ? a : b; }

__device__ __forceinline__ u32 wave_incl_scan_max(u32 v) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    u32 t = __shfl_up(v, d, WAVE);
    if ((int)lane_id() >= d) v = umax(v, t);
  }
  return v;
}

// Exclusive running max across the block's threads (identity 0); total = max.
template <int NT>
__device__ __forceinline__ u32 block_excl_scan_max(u32 v, u32* lds, u32& total) {
  constexpr int NW = NT / WAVE;
  const u32 incl = wave_incl_scan_max(v);
  if (lane_id() == WAVE - 1) lds[wave_id()] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 run = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const u32 t = lds[w];
      lds[w] = run;
      run = umax(run, t);
    }
    lds[NW] = run;
  }
  __syncthreads();
  const u32 base = lds[wave_id()];
  total = lds[NW];
  u32 prev = __shfl_up(incl, 1, WAVE);
  if (lane_id() == 0) prev = 0;
  __syncthreads();
  return umax(base, prev);
}

// A thread's PM_ITEMS (16) consecutive values as four 16-byte loads / stores (the scalar
// form issued 16 dword loads 64 B apart per lane: 16 partial requests per cache line).
__device__ __forceinline__ void pm_load16(const u32* __restrict__ in, u32 base, u32 n, u32 (&x)[PM_ITEMS]) {
  if (base + PM_ITEMS <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
    for (int q = 0; q < PM_ITEMS / 4; q++) {
      const uint4 v = p[q];
      x[4 * q] = v.x;
      x[4 * q + 1] = v.y;
      x[4 * q + 2] = v.z;
      x[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < PM_ITEMS; k++) x[k] = (base + k < n) ? in[base + k] : 0u;
  }
}

__global__ __launch_bounds__(PM_NT) void k_pmax_reduce(const u32* __restrict__ in, u32 n,
                                                        u32* __restrict__ bmax) {
  __shared__ u32 lds[PM_NT / WAVE];
  const u32 base = blockIdx.x * PM_TILE + threadIdx.x * PM_ITEMS;
  u32 x[PM_ITEMS];
  pm_load16(in, base, n, x);
  u32 m = 0;
#pragma unroll
  for (int k = 0; k < PM_ITEMS; k++) m = umax(m, x[k]);
  m = wave_reduce_max_u32(m);
  if (lane_id() == 0) lds[wave_id()] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 t = 0;
#pragma unroll
    for (int w = 0; w < PM_NT / WAVE; w++) t = umax(t, lds[w]);
    bmax[blockIdx.x] = t;
  }
}

// Single block: bmax[i] <- max(bmax[0..i))  (exclusive running max)
__global__ __launch_bounds__(1024) void k_pmax_spine(u32* __restrict__ bmax, u32 nb) {
  __shared__ u32 lds[1024 / WAVE + 1];
  __shared__ u32 carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (u32 base = 0; base < nb; base += 1024) {
    const u32 i = base + threadIdx.x;
    const u32 v = i < nb ? bmax[i] : 0u;
    u32 total;
    const u32 ex = block_excl_scan_max<1024>(v, lds, total);
    const u32 carry = carry_s;
    if (i < nb) bmax[i] = umax(carry, ex);
    __syncthreads();
    if (threadIdx.x == 0) carry_s = umax(carry, total);
    __syncthreads();
  }
}

__global__ __launch_bounds__(PM_NT) void k_pmax_down(const u32* __restrict__ in, u32 n,
                                                      const u32* __restrict__ bmax,
                                                      u32* __restrict__ out) {
  __shared__ u32 lds[PM_NT / WAVE + 1];
  const u32 base = blockIdx.x * PM_TILE + threadIdx.x * PM_ITEMS;
  u32 x[PM_ITEMS];
  pm_load16(in, base, n, x);
  u32 m = 0;
#pragma unroll
  for (int k = 0; k < PM_ITEMS; k++) m = umax(m, x[k]);
  u32 total;
  u32 run = umax(bmax[blockIdx.x], block_excl_scan_max<PM_NT>(m, lds, total));
#pragma unroll
  for (int k = 0; k < PM_ITEMS; k++) {
    run = umax(run, x[k]);
    x[k] = run;
  }
  if (base + PM_ITEMS <= n) {
    uint4* p = reinterpret_cast<uint4*>(out + base);
#pragma unroll
    for (int q = 0; q < PM_ITEMS / 4; q++) p[q] = make_uint4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
  } else {
#pragma unroll
    for (int k = 0; k < PM_ITEMS; k++)
      if (base + k < n) out[base + k] = x[k];
  }
}

// ------------------------------------------------------------- SEMI / ANTI
// Queries are the A rows SORTED by linearised start (coherent binary searches:
// neighbouring lanes walk the same path), results are scattered back by row id.
// flag[rid] = 1 when the A row qualifies (SEMI: has an overlapping B row; ANTI: has
// none).  B is sorted by linearised start with ALL its rows (no sentinels).
__global__ __launch_bounds__(256) void k_semi_flags(const u32* __restrict__ a_keys,
                                                     const u32* __restrict__ a_ends,
                                                     const u32* __restrict__ a_rids, u32 n_a,
                                                     const DevMeta* __restrict__ meta,
                                                     const u32* __restrict__ b_keys,
                                                     const u32* __restrict__ b_pmax, u32 n_b,
                                                     int anti, u32* __restrict__ flag) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_a) return;
  const u32 qs = a_keys[i], qe = a_ends[i];
  bool hit = false;
  if (n_b > 0 && qs < meta->sentinel) {          // rows with a bad chrom id carry the sentinel
    const u32 j = lower_bound_u32(b_keys, 0, n_b, qe);  // rows with b.start < a.end
    hit = j > 0 && b_pmax[j - 1] > qs;
  }
  flag[a_rids[i]] = (hit != (anti != 0)) ? 1u : 0u;
}

// Fixed-length B (every row regular with canonical length L > 0): b.end = b.start + L, so
//   a.start < b.end AND a.end > b.start  <=>  b.start in (a.start - L, a.end)
// for ANY A row -- the literal predicate rewritten, no prefix max and no `end` payload on B:
// an A row qualifies iff the sorted B starts hold a key in [a.start - L + 1, a.end).  Keys of
// another chromosome cannot fall in that range (a B row ends inside its chromosome's span).
__global__ __launch_bounds__(256) void k_semi_flags_uniform(const u32* __restrict__ a_keys,
                                                             const u32* __restrict__ a_ends,
                                                             const u32* __restrict__ a_rids, u32 n_a,
                                                             const DevMeta* __restrict__ meta,
                                                             const u32* __restrict__ b_keys, u32 n_b,
                                                             i64 uni_len, int anti, u32* __restrict__ flag) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_a) return;
  const u32 qs = a_keys[i], qe = a_ends[i];
  bool hit = false;
  if (n_b > 0 && qs < meta->sentinel) {
    const i64 lo64 = (i64)qs - uni_len + 1;
    const u32 lo_key = lo64 < 0 ? 0u : (u32)lo64;
    if (qe > lo_key) {
      const u32 lo = lower_bound_u32(b_keys, 0, n_b, lo_key);
      // the range is short (a.len + L - 1 positions): the upper bound lies a few rows on
      hit = lo < n_b && b_keys[lo] < qe;
    }
  }
  flag[a_rids[i]] = (hit != (anti != 0)) ? 1u : 0u;
}

// ---- fixed-length B sorted COARSELY (round 3) ----
// Every A row asks ONE question of the fixed-length B's sorted starts, and a 10M-row B costs four global passes
// (163 us of a 347 us SEMI at 1M x 10M) to be sorted on every bit.  Sorted WITHOUT its lowest digit (three passes)
// its rows are ordered by key >> 8, and the rows that share those 24 bits -- not even one on average at that
// density -- are looked at one by one behind a binary search on the masked keys.  The host takes this form while
// such a group is short (the density of the context's previous call; exact at any density).  (Leaving TWO digits
// unsorted -- ~200 rows per group -- was tried with the groups' boundaries tabulated: the row-by-row look through a
// group took 149 us where the two passes saved 82.)
constexpr int COARSE_SHIFT = 8;
// first row whose masked key is >= x's
__device__ __forceinline__ u32 coarse_lower(const u32* __restrict__ keys, u32 n, u32 x) {
  const u32 xm = x >> COARSE_SHIFT;
  u32 lo = 0, hi = n;
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if ((keys[mid] >> COARSE_SHIFT) < xm)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}
// rank of x = rows with key < x
__device__ __forceinline__ u32 coarse_rank(const u32* __restrict__ keys, u32 n, u32 x) {
  u32 j = coarse_lower(keys, n, x);
  const u32 xm = x >> COARSE_SHIFT;
  u32 c = j;
  for (; j < n && (keys[j] >> COARSE_SHIFT) == xm; j++) c += (u32)(keys[j] < x);
  return c;
}

__global__ __launch_bounds__(256) void k_semi_flags_uniform_coarse(const u32* __restrict__ a_keys,
                                                                    const u32* __restrict__ a_ends,
                                                                    const u32* __restrict__ a_rids, u32 n_a,
                                                                    const DevMeta* __restrict__ meta,
                                                                    const u32* __restrict__ b_keys, u32 n_b,
                                                                    i64 uni_len, int anti, u32* __restrict__ flag) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_a) return;
  const u32 qs = a_keys[i], qe = a_ends[i];
  bool hit = false;
  if (n_b > 0 && qs < meta->sentinel) {
    const i64 lo64 = (i64)qs - uni_len + 1;
    const u32 lo_key = lo64 < 0 ? 0u : (u32)lo64;
    if (qe > lo_key) {  // a B start in [lo_key, qe)?  The groups from lo_key's to (qe - 1)'s, row by row
      const u32 last = (qe - 1u) >> COARSE_SHIFT;
      for (u32 j = coarse_lower(b_keys, n_b, lo_key); !hit && j < n_b; j++) {
        const u32 k = b_keys[j];
        if ((k >> COARSE_SHIFT) > last) break;
        hit = k >= lo_key && k < qe;
      }
    }
  }
  flag[a_rids[i]] = (hit != (anti != 0)) ? 1u : 0u;
}

// ------------------------------------------------------------------- COUNT
// Regular rows: count(a) = #{b.start < a.end} - #{b.end <= a.start} over the
// regular B rows (two sorted arrays, no candidate is touched); irregular rows on
// either side are settled by the literal predicate.  A rows come ordered by start -- coarsely: the
// order only serves locality -- and the irregular ones (sentinel key, wherever the coarse order
// left them) are settled by k_count_irregular.
__global__ __launch_bounds__(256) void k_count_rows(
    const u32* __restrict__ a_keys, const u32* __restrict__ a_ends, const u32* __restrict__ a_rids,
    u32 n_a_total, SideView a, SideView b, const u32* __restrict__ b_keys_sorted,
    const u32* __restrict__ b_ends_sorted, u32 n_b_total, const u32* __restrict__ irr_b_list,
    const DevMeta* __restrict__ meta, i64* __restrict__ counts_out, i64 uni_len,
    int coarse_b = 0 /* fixed-length B sorted without its lowest digit */) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_a_total) return;
  const u32 n_reg = n_b_total - meta->irr_b;
  const u32 qs = a_keys[i], qe = a_ends[i], r = a_rids[i];
  if (qs >= meta->sentinel) return;  // an irregular A row (real keys lie below the sentinel)
  u32 below, done;  // b.start < a.end; b.end <= a.start
  if (coarse_b) {  // (uni_len > 0, no irregular B row)
    below = coarse_rank(b_keys_sorted, n_reg, qe);
    const i64 t = (i64)qs - uni_len;
    done = t < 0 ? 0u : ((u32)t == U32_MAX ? n_reg : coarse_rank(b_keys_sorted, n_reg, (u32)t + 1u));
    counts_out[r] = (i64)below - (i64)done;
    return;
  }
  below = lower_bound_u32(b_keys_sorted, 0, n_reg, qe);
  if (uni_len > 0) {
    // fixed-length B: its sorted ends are its sorted starts + L, so no second sorted array:
    // #{b.end <= a.start} = #{b.start <= a.start - L}
    const i64 t = (i64)qs - uni_len;
    done = t < 0 ? 0u : upper_bound_u32(b_keys_sorted, 0, n_reg, (u32)t);
  } else {
    done = upper_bound_u32(b_ends_sorted, 0, n_reg, qs);
  }
  i64 cnt = (i64)below - (i64)done;
  const u32 m = meta->irr_b;
  if (m) {  // rare: literal predicate against the irregular B rows
    const int ac = a.chrom[r];
    const i64 as = (i64)a.start[r] + a.start_off, ae = (i64)a.end[r] + a.end_off;
    for (u32 k = 0; k < m; k++) {
      const u32 rb = irr_b_list[k];
      cnt += literal_overlap(ac, as, ae, b.chrom[rb], (i64)b.start[rb] + b.start_off,
                             (i64)b.end[rb] + b.end_off);
    }
  }
  counts_out[r] = cnt;
}

// One block per irregular A row: literal predicate against every B row.
__global__ __launch_bounds__(256) void k_count_irregular(SideView a, SideView b,
                                                          const u32* __restrict__ irr_a_list,
                                                          i64* __restrict__ counts_out) {
  __shared__ u64 lds[256 / WAVE];
  const u32 r = irr_a_list[blockIdx.x];
  const int ac = a.chrom[r];
  const i64 as = (i64)a.start[r] + a.start_off, ae = (i64)a.end[r] + a.end_off;
  u64 c = 0;
  for (u32 j = threadIdx.x; j < b.n; j += 256)
    c += literal_overlap(ac, as, ae, b.chrom[j], (i64)b.start[j] + b.start_off,
                         (i64)b.end[j] + b.end_off);
  c = wave_reduce_sum(c);
  if (lane_id() == 0) lds[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
#pragma unroll
    for (int w = 0; w < 256 / WAVE; w++) t += lds[w];
    counts_out[r] = (i64)t;
  }
}

// ----------------------------------------------------------------- NEAREST
// chrom_lo[c] = first index of chromosome c in the start-sorted B keys.
__global__ void k_chrom_bounds(const u32* __restrict__ chrom_first, int n_chrom,
                               const u32* __restrict__ b_keys, u32 n_b, u32* __restrict__ chrom_lo) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c <= n_chrom) chrom_lo[c] = lower_bound_u32(b_keys, 0, n_b, chrom_first[c]);
}

// NEAREST wants B in (start, end) order.  Equal starts are short runs in real tables,
// so instead of a second full sort (stable by end first, then by start) the run heads
// fix their runs in place after ONE sort by start: an insertion sort of (end, rid) by
// end, one thread per run.  A run longer than NEAREST_TIE_MAX sets meta->aux0 and the
// host repeats the call with the two-sort plan (and keeps using it).
constexpr u32 NEAREST_TIE_MAX = 32;

__global__ __launch_bounds__(256) void k_fix_start_ties(const u32* __restrict__ keys,
                                                        u32* __restrict__ ends, u32* __restrict__ rids,
                                                        u32 n, DevMeta* __restrict__ meta) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 k = keys[i];
  if (i > 0 && keys[i - 1] == k) return;  // not a run head
  u32 len = 1;
  while (i + len < n && len <= NEAREST_TIE_MAX && keys[i + len] == k) len++;
  if (len == 1) return;
  if (len > NEAREST_TIE_MAX) {
    meta->aux0 = 1;
    return;
  }
  for (u32 a = 1; a < len; a++) {
    const u32 e = ends[i + a], r = rids[i + a];
    u32 b = a;
    while (b > 0 && ends[i + b - 1] > e) {
      ends[i + b] = ends[i + b - 1];
      rids[i + b] = rids[i + b - 1];
      b--;
    }
    ends[i + b] = e;
    rids[i + b] = r;
  }
}

struct alignas(16) NearestRec {
  i64 dist;
  int32_t idx;
  int32_t pad;
};

// first index in [lo, hi) with v[idx] >= x (v non-decreasing), searched BACKWARDS from hi:
// doubling steps, then a binary search inside the bracket.  The prefix-max lookups of
// NEAREST land within a few rows of `hi`, so this is 2-4 loads instead of ~20.
__device__ __forceinline__ u32 gallop_back_lower_u32(const u32* __restrict__ v, u32 lo, u32 hi, u32 x) {
  u32 step = 1, right = hi;  // invariant: v[right..hi) >= x
  while (right > lo) {
    const u32 probe = right - lo > step ? right - step : lo;
    if (v[probe] >= x) {
      right = probe;
      step <<= 1;
    } else {
      return lower_bound_u32(v, probe + 1, right, x);
    }
  }
  return lo;
}
// first index in [lo, hi) with v[idx] > x, same search
__device__ __forceinline__ u32 gallop_back_upper_u32(const u32* __restrict__ v, u32 lo, u32 hi, u32 x) {
  u32 step = 1, right = hi;  // invariant: v[right..hi) > x
  while (right > lo) {
    const u32 probe = right - lo > step ? right - step : lo;
    if (v[probe] > x) {
      right = probe;
      step <<= 1;
    } else {
      return upper_bound_u32(v, probe + 1, right, x);
    }
  }
  return lo;
}

// ---- searches through an accessor (an array served partly from LDS) -----------------
template <typename F>
__device__ __forceinline__ u32 lower_bound_f(F f, u32 lo, u32 hi, u32 x) {  // first idx with f(idx) >= x
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (f(mid) < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}
template <typename F>
__device__ __forceinline__ u32 upper_bound_f(F f, u32 lo, u32 hi, u32 x) {  // first idx with f(idx) > x
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (f(mid) <= x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}
// gallop_back_lower_u32 / gallop_back_upper_u32 over an accessor
template <typename F>
__device__ __forceinline__ u32 gallop_back_lower_f(F f, u32 lo, u32 hi, u32 x) {
  u32 step = 1, right = hi;  // invariant: f(right..hi) >= x
  while (right > lo) {
    const u32 probe = right - lo > step ? right - step : lo;
    if (f(probe) >= x) {
      right = probe;
      step <<= 1;
    } else {
      return lower_bound_f(f, probe + 1, right, x);
    }
  }
  return lo;
}
template <typename F>
__device__ __forceinline__ u32 gallop_back_upper_f(F f, u32 lo, u32 hi, u32 x) {
  u32 step = 1, right = hi;  // invariant: f(right..hi) > x
  while (right > lo) {
    const u32 probe = right - lo > step ? right - step : lo;
    if (f(probe) > x) {
      right = probe;
      step <<= 1;
    } else {
      return upper_bound_f(f, probe + 1, right, x);
    }
  }
  return lo;
}

// B sorted by (linearised start, end); b_pmax = inclusive prefix max of ends.
// A rows come sorted by start; results are scattered back by row id.
// Distance CASE of _distance.py:67-87; order ABS(d), start, end (nearest.py:392).
//
// The 1024 A rows of a block are neighbours on the linear axis, so the block first brackets
// all of its lower_bound(b_keys, a.end) results with two cooperative 64-ary searches (min and
// max a.end of the block; the keys are globally sorted, chromosome after chromosome) and stages
// that bracket of b_keys / b_pmax -- plus NR_BACK rows below it, where the upstream lookups
// gallop to -- in LDS.  Every row then searches inside the bracket, in LDS: a row's ~25 dependent
// loads are LDS loads (~0.1 us each) instead of L2 hits (~0.5 us), which is what bounded the kernel
// (10M x 10M: 0.36 ms of searching with one wave-level bracket and global loads).  An index outside
// the staged window (a bracket wider than the stage, a gallop past its lower edge) is read from
// global memory by the same accessor.
constexpr int NR_NT = 256;
#ifndef GIQL_NR_ITEMS
#define GIQL_NR_ITEMS 4
#endif
constexpr int NR_ITEMS = GIQL_NR_ITEMS;
constexpr int NR_CHROMS = 511;           // chromosome tables staged in LDS up to this many chromosomes
constexpr int NR_TQ = NR_NT * NR_ITEMS;  // A rows per block
constexpr int NR_CAP = 4096;             // staged B rows (keys + prefix max: 32 KB)
constexpr u32 NR_BACK = 128;             // rows staged below the bracket

// OUT32 (round 4, giql_hip_nearest32_dev): the result of a row is ONE 8-byte {idx_b : int32, distance : int32}
// record scattered straight to its final place out32[row id] -- SURVEY.md section 8 a9 sizes NEAREST's output as
// (int32, int32) -- so there is no record array and no unpack pass (40 us of a 1.0 ms call at 10M x 10M).  A distance
// past INT32_MAX (coordinates near the ends of the int32 range) is reported through DevMeta::aux1: the caller takes
// the int64 entry point then.
#ifndef GIQL_NR_MIN_WAVES
#define GIQL_NR_MIN_WAVES 4   // (the 36 KB of LDS allow four blocks per CU = 4 waves per SIMD: up to 128 VGPRs cost nothing)
#endif
template <bool OUT32>
__global__ __launch_bounds__(NR_NT, GIQL_NR_MIN_WAVES) void k_nearest(
    const u32* __restrict__ a_keys, const u32* __restrict__ a_ends, const u32* __restrict__ a_rids,
    u32 n_a, int n_chrom, const u32* __restrict__ chrom_first, const u32* __restrict__ chrom_lo,
    const u32* __restrict__ b_keys, const u32* __restrict__ b_pmax, const u32* __restrict__ b_rids,
    u32 n_b, int is_signed, i64 max_distance, NearestRec* __restrict__ rec_out,
    DevMeta* __restrict__ meta, int2* __restrict__ out32 = nullptr) {
  __shared__ u32 s_keys[NR_CAP], s_pmax[NR_CAP];
  __shared__ u32 s_min[NR_NT / WAVE], s_max[NR_NT / WAVE];
  __shared__ u32 s_w[2];
  __shared__ u32 s_cfirst[NR_CHROMS + 1], s_clo[NR_CHROMS + 1];
  const u32 tid = threadIdx.x;
  const u32 base = blockIdx.x * NR_TQ;
  const u32 sentinel = meta->sentinel;
  const bool lds_chroms = n_chrom <= NR_CHROMS;
  if (lds_chroms)
    for (u32 k = tid; k <= (u32)n_chrom; k += NR_NT) {
      s_cfirst[k] = chrom_first[k];
      s_clo[k] = chrom_lo[k];
    }
  u32 qs[NR_ITEMS], qe[NR_ITEMS], rr[NR_ITEMS];
  bool srch[NR_ITEMS];
  u32 emin = U32_MAX, emax = 0u;
#pragma unroll
  for (int u = 0; u < NR_ITEMS; u++) {
    const u32 i = base + u * NR_NT + tid;
    const bool live = i < n_a;
    qs[u] = live ? a_keys[i] : 0u;
    qe[u] = live ? a_ends[i] : 0u;
    rr[u] = live ? a_rids[i] : 0u;
    srch[u] = live && n_b > 0 && qs[u] < sentinel;
    if (srch[u]) {
      emin = qe[u] < emin ? qe[u] : emin;
      emax = qe[u] > emax ? qe[u] : emax;
    }
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    const u32 tmin = (u32)__shfl_xor((int)emin, d, WAVE), tmax = (u32)__shfl_xor((int)emax, d, WAVE);
    emin = tmin < emin ? tmin : emin;
    emax = tmax > emax ? tmax : emax;
  }
  if (lane_id() == 0) {
    s_min[wave_id()] = emin;
    s_max[wave_id()] = emax;
  }
  __syncthreads();
  if (wave_id() == 0) {  // the block's bracket of lower_bound(b_keys, a.end)
    u32 bmin = U32_MAX, bmax = 0u;
#pragma unroll
    for (int k = 0; k < NR_NT / WAVE; k++) {
      bmin = s_min[k] < bmin ? s_min[k] : bmin;
      bmax = s_max[k] > bmax ? s_max[k] : bmax;
    }
    u32 lo = 0, hi = n_b;
    if (bmin <= bmax && n_b > 0) {  // some row of the block searches
      lo = wave_lower_bound_u32(b_keys, 0, n_b, bmin);
      hi = wave_lower_bound_u32(b_keys, lo, n_b, bmax);
    }
    if (lane_id() == 0) {
      s_w[0] = lo;
      s_w[1] = hi;
    }
  }
  __syncthreads();
  const u32 w_lo = s_w[0], w_hi = s_w[1];
  const u32 w0 = w_lo > NR_BACK ? w_lo - NR_BACK : 0u;
  u32 w1 = w_hi < n_b ? w_hi + 1u : n_b;  // one row above: the nearest downstream row of the last lookups
  if (w1 - w0 > (u32)NR_CAP) w1 = w0;     // a bracket wider than the stage: everything from global memory
  for (u32 k = tid; k < w1 - w0; k += NR_NT) {
    s_keys[k] = b_keys[w0 + k];
    s_pmax[k] = b_pmax[w0 + k];
  }
  __syncthreads();
  const u32 wn = w1 - w0;
  auto key_at = [&](u32 j) -> u32 { return (j - w0 < wn) ? s_keys[j - w0] : b_keys[j]; };
  auto pmax_at = [&](u32 j) -> u32 { return (j - w0 < wn) ? s_pmax[j - w0] : b_pmax[j]; };
  u32 jv[NR_ITEMS];   // matched sorted-B index per row (U32_MAX: none)
  i64 dv[NR_ITEMS];
#pragma unroll
  for (int u = 0; u < NR_ITEMS; u++) {
    jv[u] = U32_MAX;
    dv[u] = 0;
    i64 best_d = 0;
#if defined(GIQL_NEAREST_ABLATE) && GIQL_NEAREST_ABLATE == 2  // timing-only build: no search
    if (srch[u] && qs[u] == 0x12345u) {
#else
    if (srch[u]) {
#endif
      const u32 s = qs[u], e = qe[u];
      // inverted row: NEAREST needs start <= end (never masks an earlier error, e.g. a
      // span overflow that made these keys meaningless)
      if (e < s && meta->status == 0) meta->status = -1;
      // chromosome of the row = last c with chrom_first[c] <= key
      const u32 c = upper_bound_u32(lds_chroms ? s_cfirst : chrom_first, 0, (u32)n_chrom + 1, s) - 1;
      const u32 blo = lds_chroms ? s_clo[c] : chrom_lo[c];
      const u32 bhi = lds_chroms ? s_clo[c + 1] : chrom_lo[c + 1];
      if (bhi > blo) {
        // global lower bound inside the block's bracket, clamped to the chromosome
        u32 hi = lower_bound_f(key_at, w_lo, w_hi, e);
        hi = hi < blo ? blo : (hi > bhi ? bhi : hi);
        u32 j = U32_MAX;
        const u32 m = hi > blo ? pmax_at(hi - 1) : 0u;  // largest end among the rows starting below a.end
        if (hi > blo && m > s) {
          // overlap: first row in (start,end) order whose end exceeds a.start
          j = gallop_back_upper_f(pmax_at, blo, hi, s);
          best_d = 0;
        } else {
          i64 up_d = 0, dn_d = 0;
          u32 up = U32_MAX, dn = U32_MAX;
          if (hi > blo) {  // m = the nearest upstream end (<= a.start)
            up = gallop_back_lower_f(pmax_at, blo, hi, m);
            up_d = (i64)s - (i64)m + 1;
          }
          if (hi < bhi) {
            dn = hi;
            dn_d = (i64)key_at(hi) - (i64)e + 1;
          }
          if (up != U32_MAX && (dn == U32_MAX || up_d <= dn_d)) {
            j = up;
            best_d = is_signed ? -up_d : up_d;
          } else if (dn != U32_MAX) {
            j = dn;
            best_d = dn_d;
          }
        }
        if (j != U32_MAX) {
          const i64 ad = best_d < 0 ? -best_d : best_d;
          if (max_distance < 0 || ad <= max_distance) {
            jv[u] = j;
            dv[u] = best_d;
          }
        }
      }
    }
  }
  // the row ids of the matches: NR_ITEMS independent gathers in flight, not one at the end of every search
  int32_t bv[NR_ITEMS];
#pragma unroll
  for (int u = 0; u < NR_ITEMS; u++) bv[u] = jv[u] != U32_MAX ? (int32_t)b_rids[jv[u]] : -1;
#pragma unroll
  for (int u = 0; u < NR_ITEMS; u++) {
    const u32 i = base + u * NR_NT + tid;
    if (i >= n_a) continue;
    const int32_t best = bv[u];
    const i64 best_d = dv[u];
    // ONE 16-byte scattered store per row (a partial-line write costs about the same
    // whatever its width); k_nearest_unpack then streams the records into the two
    // output arrays.  Two scattered stores (4 B + 8 B) took 0.39 of the kernel's 0.77 ms.
    if constexpr (OUT32) {
      const i64 d = best < 0 ? 0 : best_d;
      if (d > 0x7FFFFFFFll || d < -0x7FFFFFFFll) meta->aux1 = 1u;  // does not fit: the int64 entry point's case
      out32[rr[u]] = make_int2(best, (int)d);
    } else {
      NearestRec rec;
      rec.dist = best < 0 ? 0 : best_d;
      rec.idx = best;
      rec.pad = 0;
#if defined(GIQL_NEAREST_ABLATE) && GIQL_NEAREST_ABLATE == 1  // timing-only build: records in sorted order, no scatter
      rec_out[i] = rec;
#else
      rec_out[rr[u]] = rec;
#endif
    }
  }
}

// giql_hip_nearest32_dev with no target row at all: every record is {-1, 0}
__global__ __launch_bounds__(256) void k_nearest32_none(int2* __restrict__ out, u32 n) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = make_int2(-1, 0);
}

__global__ __launch_bounds__(256) void k_nearest_unpack(const NearestRec* __restrict__ rec, u32 n,
                                                        int32_t* __restrict__ idx_out,
                                                        i64* __restrict__ dist_out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const NearestRec v = rec[i];
  idx_out[i] = v.idx;
  dist_out[i] = v.dist;
}


// ------------------------------------------------------------ NEAREST k > 1
// The k nearest B rows of every A row in the reference's order ABS(distance), start, end
// (nearest.py:387-396) from TWO sorted views of B on the linear axis:
//   by (start, end) with the prefix max of the ends: the overlapping rows (distance 0) are the
//     rows i in [first i with pmax > a.start, lower_bound(start, a.end)) whose end exceeds a.start,
//     already in (start, end) order; the downstream rows (start >= a.end, distance start - a.end + 1)
//     follow from that lower bound on, nearest first, ties by (start, end) -- the order itself;
//   by (end, start): the upstream rows (end <= a.start, distance a.start - end + 1) lie below
//     upper_bound(end, a.start), nearest last -- walked backwards RUN by run of equal ends, each
//     run forwards (equal ends = equal distances: ascending start).
// Upstream wins a tie of distances against downstream (its start is smaller).  A row that is both
// (zero-length a and b on one point) is downstream, as the distance CASE's first matching arm says
// (_distance.py:67-87).  One thread per A row (rows sorted by start: neighbouring lanes walk
// neighbouring ranges); results go out as k 16-byte records per row, by row id.
constexpr int NEAREST_K_MAX = 1 << 20;  // (nothing in the walk depends on k: the cap only keeps n_a * k honest)
// Round 3: the block's bracket of lower_bound(b_keys, a.end) is staged in LDS as in k_nearest (keys, prefix max AND
// ends of the (start, end) view: the overlap walk filters on the ends), so the ~20 dependent loads of every row's
// search and its walk over the overlapping rows are LDS loads; the upstream view is only looked at by rows that
// still lack candidates after the overlapping ones (its run heads found by galloping back: equal ends are short
// runs), and from k = 16 on the results go straight to the output arrays (a row's k ids / k distances are 64+ / 128+
// contiguous bytes) instead of through 16-byte records and an unpack pass.
constexpr int NRK_CAP = 2048;  // staged B rows per view (keys + prefix max + ends of the (start, end) view, ends of the (end, start) view: 32 KB)
constexpr u32 NRK_EBACK = 192;   // rows of the (end, start) view staged below the bracket (the upstream walk runs backwards)

template <bool DIRECT>
__global__ __launch_bounds__(NR_NT) void k_nearest_k(
    const u32* __restrict__ a_keys, const u32* __restrict__ a_ends, const u32* __restrict__ a_rids, u32 n_a,
    int n_chrom, const u32* __restrict__ chrom_first, const u32* __restrict__ chrom_lo,
    const u32* __restrict__ chrom_lo_e, const u32* __restrict__ b_keys, const u32* __restrict__ b_ends,
    const u32* __restrict__ b_pmax, const u32* __restrict__ b_rids, const u32* __restrict__ e_ends,
    const u32* __restrict__ e_starts, const u32* __restrict__ e_rids, u32 n_b, int k, int is_signed,
    i64 max_distance, NearestRec* __restrict__ rec_out, int32_t* __restrict__ idx_out, i64* __restrict__ dist_out,
    DevMeta* __restrict__ meta) {
  __shared__ u32 s_keys[NRK_CAP], s_pmax[NRK_CAP], s_ends[NRK_CAP], s_eends[NRK_CAP];
  __shared__ u32 s_min[NR_NT / WAVE], s_max[NR_NT / WAVE], s_smin[NR_NT / WAVE], s_smax[NR_NT / WAVE];
  __shared__ u32 s_w[4];
  __shared__ u32 s_cfirst[NR_CHROMS + 1], s_clo[NR_CHROMS + 1];
  const u32 tid = threadIdx.x;
  const u32 base = blockIdx.x * NR_TQ;
  const u32 sentinel = meta->sentinel;
  const bool lds_chroms = n_chrom <= NR_CHROMS;
  if (lds_chroms)
    for (u32 c = tid; c <= (u32)n_chrom; c += NR_NT) {
      s_cfirst[c] = chrom_first[c];
      s_clo[c] = chrom_lo[c];
    }
  u32 qs[NR_ITEMS], qe[NR_ITEMS], rr[NR_ITEMS];
  bool live[NR_ITEMS], srch[NR_ITEMS];
  u32 emin = U32_MAX, emax = 0u, smin = U32_MAX, smax = 0u;
#pragma unroll
  for (int u = 0; u < NR_ITEMS; u++) {
    const u32 i = base + u * NR_NT + tid;
    live[u] = i < n_a;
    qs[u] = live[u] ? a_keys[i] : 0u;
    qe[u] = live[u] ? a_ends[i] : 0u;
    rr[u] = live[u] ? a_rids[i] : 0u;
    srch[u] = live[u] && n_b > 0 && qs[u] < sentinel;
    if (srch[u]) {
      emin = qe[u] < emin ? qe[u] : emin;
      emax = qe[u] > emax ? qe[u] : emax;
      smin = qs[u] < smin ? qs[u] : smin;
      smax = qs[u] > smax ? qs[u] : smax;
    }
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    const u32 tmin = (u32)__shfl_xor((int)emin, d, WAVE), tmax = (u32)__shfl_xor((int)emax, d, WAVE);
    const u32 umin = (u32)__shfl_xor((int)smin, d, WAVE), umax = (u32)__shfl_xor((int)smax, d, WAVE);
    emin = tmin < emin ? tmin : emin;
    emax = tmax > emax ? tmax : emax;
    smin = umin < smin ? umin : smin;
    smax = umax > smax ? umax : smax;
  }
  if (lane_id() == 0) {
    s_min[wave_id()] = emin;
    s_max[wave_id()] = emax;
    s_smin[wave_id()] = smin;
    s_smax[wave_id()] = smax;
  }
  __syncthreads();
  if (wave_id() == 0) {  // the block's bracket of lower_bound(b_keys, a.end)
    u32 bmin = U32_MAX, bmax = 0u;
#pragma unroll
    for (int w = 0; w < NR_NT / WAVE; w++) {
      bmin = s_min[w] < bmin ? s_min[w] : bmin;
      bmax = s_max[w] > bmax ? s_max[w] : bmax;
    }
    u32 lo = 0, hi = n_b;
    if (bmin <= bmax && n_b > 0) {
      lo = wave_lower_bound_u32(b_keys, 0, n_b, bmin);
      hi = wave_lower_bound_u32(b_keys, lo, n_b, bmax);
    }
    if (lane_id() == 0) {
      s_w[0] = lo;
      s_w[1] = hi;
    }
  } else if (wave_id() == 1) {  // ... and of upper_bound(e_ends, a.start), the entry point of the upstream walk
    u32 bmin = U32_MAX, bmax = 0u;
#pragma unroll
    for (int w = 0; w < NR_NT / WAVE; w++) {
      bmin = s_smin[w] < bmin ? s_smin[w] : bmin;
      bmax = s_smax[w] > bmax ? s_smax[w] : bmax;
    }
    u32 lo = 0, hi = n_b;
    if (bmin <= bmax && n_b > 0) {
      lo = wave_lower_bound_u32(e_ends, 0, n_b, bmin);
      hi = bmax == U32_MAX ? n_b : wave_lower_bound_u32(e_ends, lo, n_b, bmax + 1u);  // = upper_bound(bmax)
    }
    if (lane_id() == 0) {
      s_w[2] = lo;
      s_w[3] = hi;
    }
  }
  __syncthreads();
  const u32 w_lo = s_w[0], w_hi = s_w[1];
  const u32 w0 = w_lo > NR_BACK ? w_lo - NR_BACK : 0u;
  // k rows above: the downstream candidates of the last lookups
  u32 w1 = w_hi + (u32)k < n_b ? w_hi + (u32)k : n_b;
  if (w1 - w0 > (u32)NRK_CAP) w1 = w0;  // a bracket wider than the stage: everything from global memory
  for (u32 t = tid; t < w1 - w0; t += NR_NT) {
    s_keys[t] = b_keys[w0 + t];
    s_pmax[t] = b_pmax[w0 + t];
    s_ends[t] = b_ends[w0 + t];
  }
  // the (end, start) view's ends around the upstream entry points (the walk runs backwards from them, run by run)
  const u32 e_lo = s_w[2], e_hi = s_w[3];
  const u32 e0 = e_lo > NRK_EBACK ? e_lo - NRK_EBACK : 0u;
  u32 e1 = e_hi;
  if (e1 - e0 > (u32)NRK_CAP) e1 = e0;
  for (u32 t = tid; t < e1 - e0; t += NR_NT) s_eends[t] = e_ends[e0 + t];
  __syncthreads();
  const u32 wn = w1 - w0, en = e1 - e0;
  auto key_at = [&](u32 j) -> u32 { return (j - w0 < wn) ? s_keys[j - w0] : b_keys[j]; };
  auto pmax_at = [&](u32 j) -> u32 { return (j - w0 < wn) ? s_pmax[j - w0] : b_pmax[j]; };
  auto end_at = [&](u32 j) -> u32 { return (j - w0 < wn) ? s_ends[j - w0] : b_ends[j]; };
  auto e_end_at = [&](u32 j) -> u32 { return (j - e0 < en) ? s_eends[j - e0] : e_ends[j]; };
#pragma unroll 1
  for (int u = 0; u < NR_ITEMS; u++) {
    if (!live[u]) continue;
    const size_t o = (size_t)rr[u] * (size_t)k;
    int emitted = 0;
    auto emit = [&](i64 d, u32 rid) {
      if (DIRECT) {
        idx_out[o + emitted] = (int32_t)rid;
        dist_out[o + emitted] = d;
      } else {
        NearestRec rec;
        rec.dist = d;
        rec.idx = (int32_t)rid;
        rec.pad = 0;
        rec_out[o + emitted] = rec;
      }
      emitted++;
    };
    if (srch[u]) {
      const u32 s = qs[u], e = qe[u];
      if (e < s && meta->status == 0) meta->status = -1;  // NEAREST needs start <= end
      const u32 c = upper_bound_u32(lds_chroms ? s_cfirst : chrom_first, 0, (u32)n_chrom + 1, s) - 1;
      const u32 blo = lds_chroms ? s_clo[c] : chrom_lo[c];
      const u32 bhi = lds_chroms ? s_clo[c + 1] : chrom_lo[c + 1];
      if (bhi > blo) {
        u32 hi = lower_bound_f(key_at, w_lo, w_hi, e);  // rows [blo, hi) start before a.end
        hi = hi < blo ? blo : (hi > bhi ? bhi : hi);
        if (hi > blo && pmax_at(hi - 1) > s) {
          for (u32 j = gallop_back_upper_f(pmax_at, blo, hi, s); j < hi && emitted < k; j++)
            if (end_at(j) > s) emit(0, b_rids[j]);
        }
        if (emitted < k) {
          u32 dn = hi;
          const u32 elo = chrom_lo_e[c], ehi = chrom_lo_e[c + 1];
          // upstream cursor: the run [run_lo, run_hi) of equal ends being emitted, `cur` inside it
          // (inside the block's bracket; the ends of earlier / later chromosomes lie below / above every key of this one)
          u32 run_lo = upper_bound_f(e_end_at, e_lo, e_hi, s);
          run_lo = run_lo < elo ? elo : (run_lo > ehi ? ehi : run_lo);
          u32 run_hi = run_lo, cur = run_lo;
          while (emitted < k) {
            // next upstream candidate (skipping rows that are downstream by the CASE's first arm)
            bool has_up = false;
            u32 up_e = 0;
            while (true) {
              if (cur == run_hi) {
                if (run_lo == elo) break;
                const u32 last = run_lo - 1;
                const u32 ee = e_end_at(last);
                run_hi = run_lo;
                run_lo = gallop_back_lower_f(e_end_at, elo, last, ee);  // the head of the run that ends at `last`
                cur = run_lo;
              }
              if (e_starts[cur] < e) {
                has_up = true;
                up_e = e_end_at(cur);
                break;
              }
              cur++;
            }
            const bool has_dn = dn < bhi;
            if (!has_up && !has_dn) break;
            const i64 up_d = has_up ? (i64)s - (i64)up_e + 1 : 0;
            const i64 dn_d = has_dn ? (i64)key_at(dn) - (i64)e + 1 : 0;
            const bool take_up = has_up && (!has_dn || up_d <= dn_d);
            const i64 d = take_up ? up_d : dn_d;
            if (max_distance >= 0 && d > max_distance) break;  // every later candidate is at least as far
            emit(take_up ? (is_signed ? -d : d) : d, take_up ? e_rids[cur] : b_rids[dn]);
            if (take_up)
              cur++;
            else
              dn++;
          }
        }
      }
    }
    while (emitted < k) {  // unused slots: idx -1
      if (DIRECT) {
        idx_out[o + emitted] = -1;
        dist_out[o + emitted] = 0;
      } else {
        NearestRec rec;
        rec.dist = 0;
        rec.idx = -1;
        rec.pad = 0;
        rec_out[o + emitted] = rec;
      }
      emitted++;
    }
  }
}

// NEAREST needs start <= end on the B side too.
__global__ __launch_bounds__(256) void k_check_not_inverted(SideView s, DevMeta* __restrict__ meta) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n) return;
  if ((i64)s.end[i] + s.end_off < (i64)s.start[i] + s.start_off && meta->status == 0) meta->status = -1;
}

}  // namespace giql
