// cluster_kernels.hip.h -- CLUSTER / MERGE: the sort + scan
// operator family next to the join.
//
// The reference expands CLUSTER into a window over a window
// (src/giql/expanders/cluster.py:210-300):
//   is_new     = NOT (MAX(end) OVER (PARTITION BY chrom ORDER BY start ROWS BETWEEN
//                UNBOUNDED PRECEDING AND 1 PRECEDING) + distance >= start)
//   cluster_id = SUM(is_new) OVER (PARTITION BY chrom ORDER BY start)
// and MERGE into GROUP BY chrom, cluster_id -> MIN(start), MAX(end)
// (src/giql/expanders/merge.py:186-330).  On the linearised axis (same keys and the
// same onesweep sort as the join) the running MAX is one prefix-max over the sorted
// ends -- ends of earlier chromosomes lie below the chromosome's base, so the
// partition needs no segmentation -- and the SUM is one scan of the flags minus the
// scan value at the partition's first row.
//
// Rows need start <= end (checked on the device): then peers (equal starts) always
// share a cluster and a cluster's MAX(end) is the running max at its last row, which
// is what the kernels below read.
#pragma once
#include "dev_common.hip.h"
#include "select_kernels.hip.h"

namespace giql {

// flags[first sorted row of each non-empty partition] = 1 (flags zeroed by the caller)
__global__ void k_cluster_firsts(const u32* __restrict__ chrom_lo, int n_chrom, u32 n,
                                 u32* __restrict__ flags) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n_chrom) {
    const u32 lo = chrom_lo[c], hi = chrom_lo[c + 1];
    if (lo < hi && lo < n) flags[lo] = 1;
  }
}

// flags[i] |= (running max end of the preceding sorted rows + distance < start)
__global__ __launch_bounds__(256) void k_cluster_flags(const u32* __restrict__ keys,
                                                       const u32* __restrict__ pmax_incl, u32 n,
                                                       u64 distance, u32* __restrict__ flags) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || i == 0) return;
  if ((u64)pmax_incl[i - 1] + distance < (u64)keys[i]) flags[i] = 1;
}

// predicate := ... PREV(col) (src/giql/expanders/cluster.py:281-296, 587-640): a row stays in the running cluster only
// when it is adjacent AND the predicate holds between it (operand side A) and its immediate sorted predecessor
// (side B = the LAG over the same partition and order); a NULL operand makes the predicate not true, i.e. a new
// cluster, as the emitted CASE's ELSE arm does.  A partition's first row is flagged already (its LAG is NULL).
__global__ __launch_bounds__(256) void k_cluster_pred_flags(const u32* __restrict__ rids, u32 n, DevPreds ps,
                                                            u32* __restrict__ flags) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || i == 0 || flags[i]) return;
  if (!sel_eval(ps, (int)rids[i], (int)rids[i - 1])) flags[i] = 1;
}

// cluster id of every row, 1-based within its partition, scattered by row id
__global__ __launch_bounds__(256) void k_cluster_ids(const u32* __restrict__ keys,
                                                     const u32* __restrict__ rids,
                                                     const u32* __restrict__ flags,
                                                     const u32* __restrict__ excl, u32 n,
                                                     const u32* __restrict__ chrom_first,
                                                     int n_chrom, const u32* __restrict__ chrom_lo,
                                                     i64* __restrict__ ids_out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 c = upper_bound_u32(chrom_first, 0, (u32)n_chrom + 1, keys[i]) - 1;
  const u32 base = excl[chrom_lo[c]];
  ids_out[rids[i]] = (i64)(excl[i] + flags[i] - base);
}

// head_pos[g] = sorted index of the first row of merged region g
__global__ __launch_bounds__(256) void k_merge_heads(const u32* __restrict__ flags,
                                                     const u32* __restrict__ excl, u32 n,
                                                     u32* __restrict__ head_pos) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) head_pos[excl[i]] = i;
}

// MERGE with a predicate: a region may end while the running maximum of the partition still lies
// ahead of it, so MAX(end) is the maximum over the region's own rows.  Regions are runs of the
// sorted order: a segmented maximum inside each wave (log steps over lanes of one region), then
// one atomic per (wave, region) -- a region of any length costs n / 64 atomics at most.
// seg_end: zero-initialised, one word per region.
__global__ __launch_bounds__(256) void k_merge_segmax(const u32* __restrict__ ends,
                                                      const u32* __restrict__ excl,
                                                      const u32* __restrict__ flags, u32 n,
                                                      u32* __restrict__ seg_end) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  const u32 g = live ? excl[i] + flags[i] - 1u : 0xFFFFFFFFu;  // (every partition's first row is a head)
  u32 e = live ? ends[i] : 0u;
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const u32 oe = (u32)__shfl_up((int)e, d);
    const u32 og = (u32)__shfl_up((int)g, d);
    if (lane >= d && og == g && oe > e) e = oe;
  }
  const u32 ng = (u32)__shfl_down((int)g, 1);
  if (live && (lane == WAVE - 1 || ng != g)) atomicMax(&seg_end[g], e);
}

// one merged region per thread: chrom, MIN(start) = the head's start (sorted by start),
// MAX(end) = running max at the region's last row (seg_end[g] under a predicate), COUNT(*) = its rows
__global__ __launch_bounds__(256) void k_merge_rows(const u32* __restrict__ head_pos, u32 m, u32 n,
                                                    const u32* __restrict__ keys,
                                                    const u32* __restrict__ pmax_incl,
                                                    const u32* __restrict__ seg_end,
                                                    const u32* __restrict__ chrom_first,
                                                    const i64* __restrict__ chrom_base, int n_chrom,
                                                    int* __restrict__ out_chrom,
                                                    int* __restrict__ out_start,
                                                    int* __restrict__ out_end,
                                                    i64* __restrict__ out_count) {
  const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= m) return;
  const u32 i0 = head_pos[g];
  const u32 i1 = g + 1 < m ? head_pos[g + 1] : n;
  const u32 key = keys[i0];
  const u32 c = upper_bound_u32(chrom_first, 0, (u32)n_chrom + 1, key) - 1;
  const i64 b = chrom_base[c];
  out_chrom[g] = (int)c;
  out_start[g] = (int)((i64)key - b);
  out_end[g] = (int)((i64)(seg_end ? seg_end[g] : pmax_incl[i1 - 1]) - b);
  if (out_count) out_count[g] = (i64)(i1 - i0);
}

// ---- distinct intervals (GROUP BY chrom, start, end) -------------------------------------
// rows sorted by (start, end): a row heads a group when its (key, end) differs from the
// previous row's
__global__ __launch_bounds__(256) void k_group_flags(const u32* __restrict__ keys,
                                                     const u32* __restrict__ ends, u32 n,
                                                     u32* __restrict__ flags) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flags[i] = (i == 0 || keys[i] != keys[i - 1] || ends[i] != ends[i - 1]) ? 1u : 0u;
}

// group index of every row (scattered by row id) and one representative row per group
__global__ __launch_bounds__(256) void k_group_ids(const u32* __restrict__ rids,
                                                   const u32* __restrict__ flags,
                                                   const u32* __restrict__ excl, u32 n,
                                                   int* __restrict__ group_of_row,
                                                   int* __restrict__ rep_row) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 g = excl[i] + flags[i] - 1u;
  const u32 r = rids[i];
  group_of_row[r] = (int)g;
  if (flags[i]) rep_row[g] = (int)r;
}

// sums[group_of_row[i]] += values[i]
__global__ __launch_bounds__(256) void k_segment_sum(const i64* __restrict__ values,
                                                     const int* __restrict__ group_of_row, u64 n,
                                                     u32 n_groups, unsigned long long* __restrict__ sums,
                                                     DevMeta* __restrict__ meta) {
  const u64 stride = (u64)gridDim.x * 256;
  bool bad = false;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int g = group_of_row[i];
    if (g < 0 || (u32)g >= n_groups) bad = true;
    else atomicAdd(&sums[g], (unsigned long long)values[i]);
  }
  if (bad) atomicMin(&meta->status, -1);
}

}  // namespace giql
