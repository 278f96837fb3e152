// giql_amd/csrc/join_kernels.hip.h -- the interval-overlap join proper.
//
// Coordinates are linearised: key = chrom_base[chrom] + canonical coordinate, so
// one flat u32 axis holds every chromosome back to back and the per-chromosome
// partition of the reference (intersects_duckdb.py:1317-1330) is implicit: an
// interval never reaches the next chromosome's range.
//
// With both sides sorted by linearised start, the overlap set
//   a.start < b.end AND a.end > b.start          (intersects.py:149-154)
// of well-formed rows (start < end) is the DISJOINT union of two range queries:
//   class 1:  a.start in [b.start, b.end)   -- query b over sorted A starts
//   class 2:  b.start in (a.start, a.end)   -- query a over sorted B starts
// Every enumerated candidate is a match: no predicate test, no wasted reads.
// Rows with end <= start ("irregular": zero-length / inverted) do not satisfy
// that identity; they get a sentinel key (sorted past every real row) and go
// through the literal-predicate kernels at the bottom of this file.
#pragma once

#include <limits.h>

#include "dev_common.hip.h"

namespace giql {

// ------------------------------------------------------------------ spans
__global__ void k_init_minmax(int* __restrict__ gmin, int* __restrict__ gmax, int n_chrom,
                              DevMeta* __restrict__ meta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_chrom) {
    gmin[i] = INT_MAX;
    gmax[i] = INT_MIN;
  }
  if (i == 0) {
    meta->total_span = 0;
    meta->n_out = 0;
    meta->n_out_irr = 0;
    meta->n_out_c1 = 0;
    meta->sentinel = U32_MAX;
    meta->irr_a = 0;
    meta->irr_b = 0;
    meta->status = 0;
    meta->aux0 = 0;
    meta->aux1 = 0;
    meta->len_min_a = INT_MAX;
    meta->len_max_a = 0;
    meta->len_min_b = INT_MAX;
    meta->len_max_b = 0;
    meta->aligned_ok = 0;
    meta->unsorted_a = 0;
    meta->unsorted_b = 0;
    meta->inverted_a = 0;
    meta->inverted_b = 0;
  }
}

#ifndef GIQL_MM_NT
#define GIQL_MM_NT 256
#endif
// grid caps = ONE resident wave of blocks on 256 CUs (4 / 8 blocks per CU): measured, the
// min/max pass takes 0.28 ms with 1024 blocks against 0.35 ms with 2048, linearize 0.34 ms
// with 2048 against 0.38 ms with 8192 (tools/phase_ab.sh)
#ifndef GIQL_MM_BLOCKS
#define GIQL_MM_BLOCKS 1024
#endif
// waves per SIMD the span kernels' register allocation must allow (0: whatever the compiler takes)
#ifndef GIQL_MM_MIN_WAVES
#define GIQL_MM_MIN_WAVES 0
#endif
#if GIQL_MM_MIN_WAVES > 0
#define GIQL_MM_LAUNCH_BOUNDS(NT) __launch_bounds__(NT, GIQL_MM_MIN_WAVES)
#else
#define GIQL_MM_LAUNCH_BOUNDS(NT) __launch_bounds__(NT)
#endif
constexpr int MM_NT = GIQL_MM_NT;            // threads of the plain min/max pass
constexpr int MM_NT_HIST = 2 * GIQL_MM_NT;   // ... and of the one that also counts digits (0.343 -> 0.331 ms)
constexpr int MM_ITEMS = 8;
constexpr int MM_LDS_CHROMS = 4096;
constexpr int MM_MAX_BLOCKS = GIQL_MM_BLOCKS;  // grid cap; len_part holds 2 sides x blocks x {min,max}

// Digit histogram inside the span pass (HIST).  With every chromosome base a multiple of
// 2^24 (k_chrom_offsets, aligned form) the three low digits of key = base[c] + pos are the
// three low bytes of pos whatever the base turns out to be, and the top digit is
// base[c] >> 24 plus pos >> 24: so the pass that finds the bases can already count the
// digits -- the low three directly, the top one per (chromosome, pos >> 24), folded once
// the bases are known (k_fold_top).  The fixed-length side of the uniform form then needs
// no linearize pass at all: its first sort pass computes the keys from (chrom, start).
// All arithmetic wraps in u32, so histogram and keys agree on ANY input (memory safety of
// the sort does not depend on the layout being meaningful; aligned_ok says whether it is).
constexpr int LIN_HIST_REPLICAS = 64;  // ghist[replica][4][256], block b adds to b % 64
constexpr int MM_HIST_CHROMS = 32;   // chromosomes the per-chromosome top-digit histogram holds
constexpr int MM_TOP_WORDS = MM_HIST_CHROMS * 256;

// Per-chromosome min/max of the raw coordinates (both columns).  LDS-privatised
// atomics; a per-thread run cache keeps chromosome-sorted input (the common BED
// case) from serialising on one LDS address.
// HIST: 0 = min/max and lengths only; 1 = also the four digit histograms; 2 = only the two high
// digits (the three-stage sort scatters on bits 16-31 only: its low bits are sorted in LDS, and the
// two per-row LDS atomics of the low digits are what bounds this pass).
// The body: block `bid` of `nblk` over one side.  HIST is a compile-time constant in k_chrom_minmax (the body is
// inlined: the branches fold) and a per-side run-time value in k_chrom_minmax2 (both sides in ONE launch).
template <int NT>
__device__ __forceinline__ void chrom_minmax_body(const int HIST, const int* __restrict__ chrom,
                                                  const int* __restrict__ start, const int* __restrict__ end, i64 n,
                                                  int n_chrom, int* __restrict__ gmin, int* __restrict__ gmax,
                                                  DevMeta* __restrict__ meta, int len_bias, int which,
                                                  int* __restrict__ len_part, int start_off,
                                                  u32* __restrict__ hist_partial, u32* __restrict__ top_partial,
                                                  const u32 bid, const u32 nblk, u32* s_hist, u32* s_top) {
  extern __shared__ int mm_lds[];
  if (HIST) {
    for (int k = threadIdx.x; k < 3 * 256; k += NT) s_hist[k] = 0;
    for (int k = threadIdx.x; k < MM_TOP_WORDS; k += NT) s_top[k] = 0;
    __syncthreads();
  }
  const bool use_lds = n_chrom <= MM_LDS_CHROMS;
  int* lmin = use_lds ? mm_lds : gmin;
  int* lmax = use_lds ? mm_lds + n_chrom : gmax;
  if (use_lds) {
    for (int c = threadIdx.x; c < n_chrom; c += NT) {
      lmin[c] = INT_MAX;
      lmax[c] = INT_MIN;
    }
    __syncthreads();
  }
  int cur = -1, mn = INT_MAX, mx = INT_MIN;
  int lmn = INT_MAX, lmx = 0;  // canonical length range (0 as soon as a row is irregular)
  bool bad = false;
  // Is the side already in (chrom id, start) order -- the order of the linear axis?  One compare per row with
  // its predecessor, which is the previous LANE's row (consecutive lanes hold consecutive rows): a DPP shift,
  // no LDS; lane 0 of a wave loads its predecessor itself.  An irregular row (sentinel key) counts as out of order.
  bool inv = false;
  bool neg = false;  // a row with canonical end < canonical start
  // (pc0, ps0: the predecessor of lane 0's row, loaded by the caller in the same batch as the rows themselves --
  // loaded here, after the rows had arrived, it was one more exposed round trip per item: 240 -> 267 us at 100M rows)
  auto order = [&](const int c, const int s, const int pc0, const int ps0, const bool ok) {
    int pc = __builtin_amdgcn_update_dpp(0, c, 0x138, 0xF, 0xF, false);   // wave_shr:1
    int ps = __builtin_amdgcn_update_dpp(0, s, 0x138, 0xF, 0xF, false);
    if (lane_id() == 0) {
      pc = pc0;
      ps = ps0;
    }
    if (ok && (pc > c || (pc == c && ps > s))) inv = true;
  };
  // This pass is instruction-heavy (round 4: ~150 instructions per row in the ISA, more than half of them scalar), and
  // timing-only builds without any LDS atomic run no faster.  The per-row work is kept branch-free where that paid,
  // and what only pays on SORTED input -- the per-thread run cache of the chromosome range, the wave-uniform shortcut of
  // the histogram -- is tried only while the wave has seen no row out of order (`sorted_so_far`, wave-uniform:
  // refreshed once per tile from the order check's own flag): 0.30-0.32 -> 0.285-0.29 ms.  (A fully straight-line loop
  // for shuffled input -- a third fewer vector, six times fewer scalar instructions -- was SLOWER, 0.32 ms: the pass is
  // not bound by instruction issue; DESIGN.md section 7, profiles/r04y_span_branch_free_rows_ab.log.)
  bool sorted_so_far = true;
  // One row: length range, per-chromosome min/max through the run cache, digit counts.
  auto row = [&](const int c, const int s, const int e, const bool ok) {
    if (ok) {
      // canonical length (end - start + the encodings' offset), saturated to int range; <= 0: an irregular row --
      // this side is not uniform (lmn = 0) and not "sorted"
      const i64 len64 = (i64)e - (i64)s + (i64)len_bias;
      const bool regular = len64 > 0;
      const int len = (int)(len64 > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : len64);
      lmn = regular ? (len < lmn ? len : lmn) : 0;
      lmx = (regular && len > lmx) ? len : lmx;
      inv = inv || !regular;
      neg = neg || len64 < 0;
      if (c < 0 || c >= n_chrom) {
        bad = true;
      } else if (!sorted_so_far && use_lds) {
        // unsorted input: the chromosome changes from row to row and the run cache would flush every time
        const int lo = s < e ? s : e, hi = s < e ? e : s;
#if !defined(GIQL_MM_ABLATE)
        atomicMin(&mm_lds[c], lo);
        atomicMax(&mm_lds[n_chrom + c], hi);
#endif
      } else {
        if (c != cur) {
#if defined(GIQL_MM_ABLATE)  // timing-only build: (almost) no LDS atomics
          if (cur >= 0 && (s & 0xFF) == 0) {
#else
          if (cur >= 0) {
#endif
            // (two explicit branches: through the generic `lmin` pointer these compile to FLAT atomics)
            if (use_lds) {
              atomicMin(&mm_lds[cur], mn);
              atomicMax(&mm_lds[n_chrom + cur], mx);
            } else {
              atomicMin(&gmin[cur], mn);
              atomicMax(&gmax[cur], mx);
            }
          }
          cur = c;
          mn = INT_MAX;
          mx = INT_MIN;
        }
        const int lo = s < e ? s : e, hi = s < e ? e : s;
        mn = lo < mn ? lo : mn;
        mx = hi > mx ? hi : mx;
      }
    }
#if defined(GIQL_MM_ABLATE_HIST)  // timing-only build: (almost) no histogram atomics -- what the loads alone cost
    if (HIST && (s & 0x3FF) == 0) {
#else
    if (HIST) {
#endif
      // n_chrom <= MM_HIST_CHROMS here (host); a bad id (flagged above) is masked into range
      const u32 pos = (u32)(s + start_off);
      const u32 tb = ((u32)c & (MM_HIST_CHROMS - 1)) * 256u + (pos >> 24);
      if (HIST == 1 && ok) {
        atomicAdd(&s_hist[pos & 0xFFu], 1u);
        atomicAdd(&s_hist[256 + ((pos >> 8) & 0xFFu)], 1u);
      }
      // chromosome- and position-sorted input makes the two high digits wave-uniform:
      // one add per wave instead of 64 serialised ones on the same LDS word
      if (!sorted_so_far) {   // (wave-uniform) shuffled input: the test would fail for every row
        if (ok) {
          atomicAdd(&s_hist[512 + ((pos >> 16) & 0xFFu)], 1u);
          atomicAdd(&s_top[tb], 1u);
        }
        return;
      }
      const u64 act = __ballot(ok);
      if (act != 0) {
        const u32 d2 = (pos >> 16) & 0xFFu;
        // the rows of a wave are consecutive and the bound cuts a SUFFIX of lanes, so the
        // first executing lane is valid whenever any lane is: readfirstlane (no LDS shuffle)
        const u32 d2f = (u32)__builtin_amdgcn_readfirstlane((int)d2);
        const u32 tbf = (u32)__builtin_amdgcn_readfirstlane((int)tb);
        const bool leader = lane_id() == (u32)__ffsll((long long)act) - 1u;
        if (__ballot(ok && d2 == d2f) == act) {
          if (leader) atomicAdd(&s_hist[512 + d2f], (u32)__popcll(act));
        } else if (ok) {
          atomicAdd(&s_hist[512 + d2], 1u);
        }
        if (__ballot(ok && tb == tbf) == act) {
          if (leader) atomicAdd(&s_top[tbf], (u32)__popcll(act));
        } else if (ok) {
          atomicAdd(&s_top[tb], 1u);
        }
      }
    }
  };
  // Tiles of 4 * NT consecutive rows, grid-stride.  Round 4: 16-byte loads -- thread tid holds rows 4 tid .. 4 tid + 3
  // of the tile (one dwordx4 per column) and, where the columns are 16-byte aligned, loads them non-temporally: the
  // box reads 7.0-7.1 TB/s that way against 6.2-6.4 with default-policy and ~4.6 with 4-byte loads (stream probe,
  // profiles/r04a_stream_probe.log; nt on 4-byte loads is SLOWER than plain ones: 0.33 -> 0.39 ms here).  Two tiles
  // are in flight per iteration (six 16-byte loads per thread).  A row's predecessor is the thread's previous
  // element, for element 0 the previous lane's element 3 (a DPP shift), for lane 0 a load of its own.
  constexpr u32 TILE = 4u * NT;
  const u64 n_full = (u64)n / TILE;
  const bool vec_ok = ((((uintptr_t)chrom) | ((uintptr_t)start) | ((uintptr_t)end)) & 15u) == 0;
  typedef int mm_i4 __attribute__((ext_vector_type(4)));
  auto tile4 = [&](const u64 t, mm_i4& c4, mm_i4& s4, mm_i4& e4, int& pc, int& ps) {
    const u64 r0 = t * TILE + 4u * threadIdx.x;
    if (vec_ok) {
      c4 = ld_stream(reinterpret_cast<const mm_i4*>(chrom + r0));
      s4 = ld_stream(reinterpret_cast<const mm_i4*>(start + r0));
      e4 = ld_stream(reinterpret_cast<const mm_i4*>(end + r0));
    } else {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        c4[u] = chrom[r0 + u];
        s4[u] = start[r0 + u];
        e4[u] = end[r0 + u];
      }
    }
    pc = ps = INT_MIN;
    if (lane_id() == 0 && r0 != 0) {  // row 0 of the table has no predecessor
      pc = chrom[r0 - 1];
      ps = start[r0 - 1];
    }
  };
  auto rows4 = [&](const mm_i4& c4, const mm_i4& s4, const mm_i4& e4, const int pc0, const int ps0) {
    // element 0 against the previous lane's element 3 (lane 0: the loaded predecessor), 1..3 inside the thread
    int pc = __builtin_amdgcn_update_dpp(0, c4[3], 0x138, 0xF, 0xF, false);   // wave_shr:1
    int ps = __builtin_amdgcn_update_dpp(0, s4[3], 0x138, 0xF, 0xF, false);
    if (lane_id() == 0) {
      pc = pc0;
      ps = ps0;
    }
    if (pc > c4[0] || (pc == c4[0] && ps > s4[0])) inv = true;
#pragma unroll
    for (int u = 1; u < 4; u++)
      if (c4[u - 1] > c4[u] || (c4[u - 1] == c4[u] && s4[u - 1] > s4[u])) inv = true;
#pragma unroll
    for (int u = 0; u < 4; u++) row(c4[u], s4[u], e4[u], true);
    if (sorted_so_far) sorted_so_far = __ballot(inv) == 0ull;
  };
  {
    u64 t = bid;
    for (; t + nblk < n_full; t += 2ull * nblk) {   // two tiles in flight
      mm_i4 c4a, s4a, e4a, c4b, s4b, e4b;
      int pca, psa, pcb, psb;
      tile4(t, c4a, s4a, e4a, pca, psa);
      tile4(t + nblk, c4b, s4b, e4b, pcb, psb);
      rows4(c4a, s4a, e4a, pca, psa);
      rows4(c4b, s4b, e4b, pcb, psb);
    }
    if (t < n_full) {
      mm_i4 c4a, s4a, e4a;
      int pca, psa;
      tile4(t, c4a, s4a, e4a, pca, psa);
      rows4(c4a, s4a, e4a, pca, psa);
    }
  }
  if (bid == (u32)(n_full % nblk)) {  // the ragged tail: one block
    const u64 base = n_full * TILE;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const u64 i = base + (u64)u * NT + threadIdx.x;
      const bool ok = i < (u64)n;
      const int c_ = ok ? chrom[i] : 0, s_ = ok ? start[i] : 0;
      const bool pred = ok && i > 0 && lane_id() == 0;
      order(c_, s_, pred ? chrom[i - 1] : INT_MIN, pred ? start[i - 1] : INT_MIN, ok);
      row(c_, s_, ok ? end[i] : 0, ok);
    }
  }
  if (HIST) {
    __syncthreads();
    const size_t rep = bid % LIN_HIST_REPLICAS;
    u32* g = hist_partial + rep * 1024;
    for (int k = threadIdx.x; k < 3 * 256; k += NT) {
      const u32 v = s_hist[k];
      if (v) atomicAdd(&g[k], v);
    }
    u32* gt = top_partial + rep * MM_TOP_WORDS;
    for (int k = threadIdx.x; k < MM_TOP_WORDS; k += NT) {
      const u32 v = s_top[k];
      if (v) atomicAdd(&gt[k], v);
    }
  }
  if (cur >= 0) {
    if (use_lds) {
      atomicMin(&mm_lds[cur], mn);
      atomicMax(&mm_lds[n_chrom + cur], mx);
    } else {
      atomicMin(&gmin[cur], mn);
      atomicMax(&gmax[cur], mx);
    }
  }
  if (bad) meta->status = -4;  // GIQL_ERR_CHROM
  if (__ballot(inv) != 0ull && lane_id() == 0) *(which ? &meta->unsorted_b : &meta->unsorted_a) = 1u;
  if (__ballot(neg) != 0ull && lane_id() == 0) *(which ? &meta->inverted_b : &meta->inverted_a) = 1u;
  {
    // block-reduce the length range into this block's slot (no global atomics:
    // thousands of waves hitting two addresses serialise for ~0.2 ms)
    __shared__ int s_len[2][NT / WAVE];
    int a = lmn, b = lmx;
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
      const int ta = __shfl_xor(a, d, WAVE), tb = __shfl_xor(b, d, WAVE);
      a = ta < a ? ta : a;
      b = tb > b ? tb : b;
    }
    if (lane_id() == 0) {
      s_len[0][wave_id()] = a;
      s_len[1][wave_id()] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 1; k < NT / WAVE; k++) {
        a = s_len[0][k] < a ? s_len[0][k] : a;
        b = s_len[1][k] > b ? s_len[1][k] : b;
      }
      len_part[(which * MM_MAX_BLOCKS + bid) * 2 + 0] = a;
      len_part[(which * MM_MAX_BLOCKS + bid) * 2 + 1] = b;
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int c = threadIdx.x; c < n_chrom; c += NT) {
      if (lmin[c] <= lmax[c]) {
        atomicMin(&gmin[c], lmin[c]);
        atomicMax(&gmax[c], lmax[c]);
      }
    }
  }
}

template <int HIST, int NT>
__global__ GIQL_MM_LAUNCH_BOUNDS(NT) void k_chrom_minmax(const int* __restrict__ chrom,
                                                         const int* __restrict__ start,
                                                         const int* __restrict__ end, i64 n,
                                                         int n_chrom, int* __restrict__ gmin,
                                                         int* __restrict__ gmax,
                                                         DevMeta* __restrict__ meta, int len_bias,
                                                         int which, int* __restrict__ len_part,
                                                         int start_off, u32* __restrict__ hist_partial,
                                                         u32* __restrict__ top_partial) {
  __shared__ u32 s_hist[HIST ? 3 * 256 : 1];
  __shared__ u32 s_top[HIST ? MM_TOP_WORDS : 1];
  chrom_minmax_body<NT>(HIST, chrom, start, end, n, n_chrom, gmin, gmax, meta, len_bias, which, len_part, start_off,
                        hist_partial, top_partial, blockIdx.x, gridDim.x, s_hist, s_top);
}

// Both sides in ONE launch (round 4): the smaller side's span pass is a latency-bound 20-50 us kernel of its own
// in front of the larger side's; its blocks -- a share of the grid in proportion to its rows -- run beside the
// larger side's instead.  Blocks [0, a.nblk) take side A, the rest side B.
struct MmSide {
  const int* chrom;
  const int* start;
  const int* end;
  i64 n;
  int len_bias, start_off, hist;
  u32 nblk;
  u32* hist_partial;
  u32* top_partial;
};
template <int NT>
__global__ GIQL_MM_LAUNCH_BOUNDS(NT) void k_chrom_minmax2(MmSide a, MmSide b, int n_chrom, int* __restrict__ gmin,
                                                      int* __restrict__ gmax, DevMeta* __restrict__ meta,
                                                      int* __restrict__ len_part) {
  __shared__ u32 s_hist[3 * 256];
  __shared__ u32 s_top[MM_TOP_WORDS];
  if (blockIdx.x < a.nblk)
    chrom_minmax_body<NT>(a.hist, a.chrom, a.start, a.end, a.n, n_chrom, gmin, gmax, meta, a.len_bias, 0, len_part,
                          a.start_off, a.hist_partial, a.top_partial, blockIdx.x, a.nblk, s_hist, s_top);
  else
    chrom_minmax_body<NT>(b.hist, b.chrom, b.start, b.end, b.n, n_chrom, gmin, gmax, meta, b.len_bias, 1, len_part,
                          b.start_off, b.hist_partial, b.top_partial, blockIdx.x - a.nblk, b.nblk, s_hist, s_top);
}

// Single block: chrom_base[c] = (exclusive prefix of spans) - (lowest canonical
// coordinate of c), so key = chrom_base[c] + canonical coordinate.
__global__ __launch_bounds__(256) void k_chrom_offsets(const int* __restrict__ gmin,
                                                        const int* __restrict__ gmax, int n_chrom,
                                                        int off_min, int off_max,
                                                        i64* __restrict__ chrom_base,
                                                        u32* __restrict__ chrom_first,
                                                        DevMeta* __restrict__ meta,
                                                        const int* __restrict__ len_part, int nblk_a,
                                                        int nblk_b, int want_aligned,
                                                        u32* __restrict__ abase) {
  __shared__ u64 lds[256 / WAVE + 1];
  __shared__ u64 carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  // canonical length range per side from the min/max kernels' block partials
  if (threadIdx.x < 2 * WAVE) {
    const int which = threadIdx.x / WAVE;  // wave 0 -> side A, wave 1 -> side B
    const int nblk = which ? nblk_b : nblk_a;
    int a = INT_MAX, b = 0;
    for (int k = lane_id(); k < nblk; k += WAVE) {
      const int ta = len_part[(which * MM_MAX_BLOCKS + k) * 2 + 0];
      const int tb = len_part[(which * MM_MAX_BLOCKS + k) * 2 + 1];
      a = ta < a ? ta : a;
      b = tb > b ? tb : b;
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
      const int ta = __shfl_xor(a, d, WAVE), tb = __shfl_xor(b, d, WAVE);
      a = ta < a ? ta : a;
      b = tb > b ? tb : b;
    }
    if (lane_id() == 0) {
      if (which) {
        meta->len_min_b = a;
        meta->len_max_b = b;
      } else {
        meta->len_min_a = a;
        meta->len_max_a = b;
      }
    }
  }
  __syncthreads();
  for (int base = 0; base < n_chrom; base += 256) {
    const int c = base + threadIdx.x;
    i64 lo = 0;
    u64 span = 0;
    if (c < n_chrom && gmin[c] <= gmax[c]) {
      lo = (i64)gmin[c] + off_min;
      span = (u64)((i64)gmax[c] + off_max - lo + 1);
    }
    u64 total;
    const u64 ex = block_excl_scan<u64, 256>(span, lds, total);
    const u64 carry = carry_s;
    if (c < n_chrom) {
      chrom_base[c] = (i64)(carry + ex) - lo;
      const u64 f = carry + ex;
      chrom_first[c] = f > 0xFFFFFFFFull ? U32_MAX : (u32)f;
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const u64 total = carry_s;
    meta->total_span = total;
    chrom_first[n_chrom] = total > 0xFFFFFFFFull ? U32_MAX : (u32)total;
    if (total > 0xFFFFFFFFull) {
      meta->status = -5;  // GIQL_ERR_SPAN
      meta->sentinel = U32_MAX;
    } else {
      meta->sentinel = (u32)total;  // one past the largest real key
    }
    if (want_aligned) {
      // Aligned form (n_chrom <= MM_HIST_CHROMS): base[c] = (2^24-buckets of the chromosomes
      // before c) << 24, key = base[c] + canonical coordinate.  abase is written whatever the
      // outcome -- wrapped mod 2^32 it stays consistent with the span pass's histogram -- and
      // replaces the tight bases only when every coordinate is >= 0 and the buckets fit below
      // the top one (kept for the sentinel).
      u32 running = 0;
      bool ok = true;
      // (ids past n_chrom are an error, reported after the call; their rows are counted and
      // keyed under id & 31 with base 0, so the sort stays consistent until then)
      for (int c = n_chrom; c < MM_HIST_CHROMS; c++) abase[c] = 0;
      for (int c = 0; c < n_chrom; c++) {
        abase[c] = (running & 0xFFu) << 24;
        if (gmin[c] <= gmax[c]) {
          const i64 lo = (i64)gmin[c] + off_min, hi = (i64)gmax[c] + off_max;
          if (lo < 0 || hi > 0x7FFFFFFFll) ok = false;
          const u32 nb = (u32)((hi < 0 ? 0 : hi) >> 24) + 1u;
          running += nb;
          if (running > 255u) ok = false;
        }
      }
      if (ok) {
        running = 0;
        for (int c = 0; c < n_chrom; c++) {
          chrom_base[c] = (i64)running << 24;
          chrom_first[c] = running << 24;
          if (gmin[c] <= gmax[c]) running += (u32)(((i64)gmax[c] + off_max) >> 24) + 1u;
        }
        const u64 atotal = (u64)running << 24;
        meta->total_span = atotal;
        chrom_first[n_chrom] = (u32)atotal;
        meta->sentinel = (u32)atotal;
        meta->aligned_ok = 1;
      }
    }
  }
}

// Top-digit histogram of the aligned keys from the per-(chromosome, pos >> 24) counts of the
// span pass: hist[3][(j + base[c] >> 24) mod 256] += sum_r top[r][c][j], added into replica 0's
// digit-3 row (zero in every replica until now) for k_digit_offsets.
__global__ __launch_bounds__(256) void k_fold_top(const u32* __restrict__ top_partial,
                                                   const u32* __restrict__ abase,
                                                   u32* __restrict__ hist_partial) {
  // one block per chromosome slot, thread = pos >> 24 bin: 64 replica loads per thread
  const u32 c = blockIdx.x, j = threadIdx.x;
  u32 sum = 0;
#pragma unroll 8
  for (int r = 0; r < LIN_HIST_REPLICAS; r++) sum += top_partial[(size_t)r * MM_TOP_WORDS + c * 256 + j];
  if (sum) atomicAdd(&hist_partial[3 * 256 + ((j + (abase[c] >> 24)) & 0xFFu)], sum);
}

// Both sides in one launch each (the query side of the fixed-length form is sorted from its raw columns
// too, round 3): blockIdx.y = side.
__global__ __launch_bounds__(256) void k_fold_top2(const u32* __restrict__ top_a, const u32* __restrict__ top_b,
                                                    const u32* __restrict__ abase, u32* __restrict__ hist_a,
                                                    u32* __restrict__ hist_b) {
  const u32* top_partial = blockIdx.y ? top_b : top_a;
  u32* hist_partial = blockIdx.y ? hist_b : hist_a;
  const u32 c = blockIdx.x, j = threadIdx.x;
  u32 sum = 0;
#pragma unroll 8
  for (int r = 0; r < LIN_HIST_REPLICAS; r++) sum += top_partial[(size_t)r * MM_TOP_WORDS + c * 256 + j];
  if (sum) atomicAdd(&hist_partial[3 * 256 + ((j + (abase[c] >> 24)) & 0xFFu)], sum);
}

// -------------------------------------------------------------- linearise
#ifndef GIQL_LIN_NT
#define GIQL_LIN_NT 256
#endif
#ifndef GIQL_LIN_BLOCKS
#define GIQL_LIN_BLOCKS 2048
#endif
constexpr int LIN_NT = GIQL_LIN_NT;
constexpr int LIN_MAX_BLOCKS = GIQL_LIN_BLOCKS;
constexpr int LIN_BASE_CAP = 1024;     // chromosome bases staged in LDS (8 KB)
#ifndef GIQL_LIN_UNROLL
#define GIQL_LIN_UNROLL 4
#endif
#ifndef GIQL_LIN_UNIFORM_FROM
#define GIQL_LIN_UNIFORM_FROM 2  // digits >= this try the wave-uniform shortcut
#endif

// keys[i] = linearised canonical start, ends[i] = linearised canonical end.
// Irregular rows (canonical end <= start) get the sentinel key and are appended
// to irr_list -- unless keep_irregular, where every row keeps its real key (the
// prefix-max operators are exact for any row).  which = 0 for side A, 1 for B.
// keys / ends may be NULL (only the irregular list is wanted).
// hist_partial (optional): 64 replicas of the 4 x 256 digit histogram of the keys
// (for the onesweep sort); each block adds its LDS histogram to one replica, so an
// address sees at most grid/64 adds; k_digit_offsets sums the replicas.
template <bool END_HIST>
__global__ __launch_bounds__(LIN_NT) void k_linearize(
    const int* __restrict__ chrom, const int* __restrict__ start, const int* __restrict__ end, u32 n,
    int start_off, int end_off, int n_chrom, const i64* __restrict__ chrom_base,
    u32* __restrict__ keys, u32* __restrict__ ends, u32* __restrict__ irr_list,
    DevMeta* __restrict__ meta, int which, int keep_irregular, u32* __restrict__ hist_partial,
    u32* __restrict__ hist_partial_end) {
  __shared__ u32 s_hist[4 * 256];
  __shared__ u32 s_hist_e[END_HIST ? 4 * 256 : 1];  // same histogram over the END keys
  // the per-chromosome bases sit in LDS: the base lookup depends on the row's chrom
  // load, and an LDS hit is a far shorter second hop than a global one
  __shared__ i64 s_base[LIN_BASE_CAP];
  const bool lds_base = n_chrom <= LIN_BASE_CAP;
  if (lds_base)
    for (int k = threadIdx.x; k < n_chrom; k += LIN_NT) s_base[k] = chrom_base[k];
  if (hist_partial) {
#pragma unroll
    for (int k = threadIdx.x; k < 4 * 256; k += LIN_NT) {
      s_hist[k] = 0;
      if (END_HIST) s_hist_e[k] = 0;
    }
  }
  __syncthreads();
  const u32 sentinel = meta->sentinel;
  u32* irr_count = which ? &meta->irr_b : &meta->irr_a;
  const u32 stride = gridDim.x * LIN_NT;
  // every lane runs the same number of iterations so the ballots are full-wave
  const u32 n_iter = (n + stride - 1) / stride;
  constexpr int UNROLL = GIQL_LIN_UNROLL;  // rows in flight per thread (a 1-row loop is latency-bound)
  const u32 i0 = blockIdx.x * LIN_NT + threadIdx.x;
  for (u32 it0 = 0; it0 < n_iter; it0 += UNROLL) {
    u32 idx[UNROLL];
    bool okv[UNROLL];
    int cv[UNROLL], sv[UNROLL], ev[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const u64 i64_ = (u64)i0 + (u64)(it0 + u) * stride;
      okv[u] = (it0 + u) < n_iter && i64_ < n;
      idx[u] = (u32)i64_;
      cv[u] = okv[u] ? chrom[idx[u]] : 0;
      sv[u] = okv[u] ? start[idx[u]] : 0;
      ev[u] = (okv[u] && end) ? end[idx[u]] : 1;  // end == NULL: a side known to be regular
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      if (it0 + u >= n_iter) break;  // wave-uniform
      const bool ok = okv[u];
      const u32 i = idx[u];
      bool irr = false;
      u32 k = sentinel, ke = sentinel;
      if (ok) {
        const i64 cs = (i64)sv[u] + start_off;
        const i64 ce = (i64)ev[u] + end_off;
        const int c = cv[u];
        const bool c_ok = c >= 0 && c < n_chrom;  // bad ids were flagged by k_chrom_minmax
        irr = c_ok && !keep_irregular && end != nullptr && ce <= cs;
        if (c_ok && !irr) {
          const i64 b = lds_base ? s_base[c] : chrom_base[c];
          k = (u32)(b + cs);
          ke = (u32)(b + ce);
        }
#if defined(GIQL_LIN_ABLATE) && (GIQL_LIN_ABLATE & 2)  // timing-only build: no key stores
        if (k == 0x12345u && ke == 0x54321u) keys[i] = k;
#else
        if (keys) keys[i] = k;
        if (ends) ends[i] = ke;
#endif
      }
#if defined(GIQL_LIN_ABLATE) && (GIQL_LIN_ABLATE & 1)  // timing-only build: no histogram
      if (false) {
#else
      if (hist_partial) {
#endif
        const u64 act = __ballot(ok);
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const u32 d = (k >> (8 * p)) & 0xFFu;
          if (p >= GIQL_LIN_UNIFORM_FROM) {
            // chromosome-sorted input makes the HIGH digits wave-uniform: one add.
            // Valid lanes are a prefix of the wave (row index grows with the lane),
            // so lane 0 is valid whenever any lane is: readfirstlane, no LDS shuffle.
            // (The low digits never are: they skip the test and its ballots.)
            const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
            const u64 same = __ballot(ok && d == d0);
            if (act != 0 && same == act) {
              if (lane_id() == 0)
                atomicAdd(&s_hist[p * 256 + d0], (u32)__popcll(act));
            } else if (ok) {
              atomicAdd(&s_hist[p * 256 + d], 1u);
            }
          } else if (ok) {
            atomicAdd(&s_hist[p * 256 + d], 1u);
          }
          if (END_HIST && ok) atomicAdd(&s_hist_e[p * 256 + ((ke >> (8 * p)) & 0xFFu)], 1u);
        }
      }
      const u64 m = __ballot(irr);
      if (m) {
        u32 base = 0;
        if (lane_id() == 0) base = atomicAdd(irr_count, (u32)__popcll(m));
        base = __shfl(base, 0, WAVE);
        if (irr) irr_list[base + (u32)__popcll(m & lanemask_lt())] = i;
      }
    }
  }
  if (hist_partial) {
    __syncthreads();
    u32* g = hist_partial + (size_t)(blockIdx.x % LIN_HIST_REPLICAS) * 1024;
#pragma unroll
    for (int k = threadIdx.x; k < 4 * 256; k += LIN_NT) {
      const u32 v = s_hist[k];
      if (v) atomicAdd(&g[k], v);
      if (END_HIST) {
        const u32 ve = s_hist_e[k];
        if (ve) atomicAdd(&hist_partial_end[(size_t)(blockIdx.x % LIN_HIST_REPLICAS) * 1024 + k], ve);
      }
    }
  }
}

// ---- query keys against a table INDEX (giql_hip_index_create_dev, round 4) ----
// The index fixed the linear axis when it was built: chromosome c of the indexed table owns keys
// [first[c], first[c + 1]) (2^24-aligned bases; every indexed row starts AND ends inside).  A query row is placed
// on that axis: key = first[c] + start, end key = first[c] + end CLAMPED to the chromosome's range (no indexed row
// starts beyond it, and an unclamped end would reach into the next chromosome's keys).  A row that cannot match --
// its chromosome is not in the index, its end lies at or below 0, its start at or beyond the range -- gets the
// sentinel key: it sorts behind every live row and is counted in *dead (the windows of the bucket stage skip that
// many rows at the end of the sorted order, as they skip irregular rows in the ordinary plan).  An IRREGULAR row
// (canonical end <= start) that is otherwise live cannot be answered by range queries: flagged in *irregular, the
// host declines the call.  The 4 x 256 digit histogram of the keys for the sort, as in k_linearize.
__global__ __launch_bounds__(LIN_NT) void k_index_query_keys(
    const int* __restrict__ chrom, const int* __restrict__ start, const int* __restrict__ end, u32 n, int start_off,
    int end_off, int n_chrom_idx, const u32* __restrict__ first, u32 sentinel, u32* __restrict__ keys,
    u32* __restrict__ ends, u32* __restrict__ dead, u32* __restrict__ irregular, int* __restrict__ len_max,
    u32* __restrict__ hist_partial) {
  __shared__ u32 s_hist[4 * 256];
  __shared__ u32 s_first[MM_HIST_CHROMS + 1];
  for (int k = threadIdx.x; k < 4 * 256; k += LIN_NT) s_hist[k] = 0;
  for (int k = threadIdx.x; k <= n_chrom_idx && k <= MM_HIST_CHROMS; k += LIN_NT) s_first[k] = first[k];
  __syncthreads();
  const u32 stride = gridDim.x * LIN_NT;
  u32 n_dead = 0;
  int lmax = 0;
  bool irr = false;
  for (u32 i = blockIdx.x * LIN_NT + threadIdx.x; i < n; i += stride) {
    const int c = chrom[i];
    const i64 cs = (i64)start[i] + start_off, ce = (i64)end[i] + end_off;
    u32 k = sentinel, ke = sentinel;
    if (c >= 0 && c < n_chrom_idx) {
      const i64 lo = (i64)s_first[c], hi = (i64)s_first[c + 1];
      if (ce <= cs) {
        irr = true;  // (an irregular row on an indexed chromosome: the literal predicate may still hold for it)
      } else if (ce > 0 && lo + cs < hi) {
        const i64 s_ = cs < 0 ? 0 : cs;     // (no indexed row starts below 0: the range query loses nothing)
        k = (u32)(lo + s_);
        const i64 e_ = lo + ce;
        ke = (u32)(e_ > hi ? hi : e_);
        const i64 len = (i64)ke - (i64)k;
        lmax = len > lmax ? (int)len : lmax;
      }
    }
    keys[i] = k;
    ends[i] = ke;
    if (k == sentinel) n_dead++;
#pragma unroll
    for (int p = 0; p < 4; p++) atomicAdd(&s_hist[p * 256 + ((k >> (8 * p)) & 0xFFu)], 1u);
  }
  n_dead = wave_reduce_sum(n_dead);
  lmax = (int)wave_reduce_max_u32((u32)lmax);
  if (lane_id() == 0) {
    if (n_dead) atomicAdd(dead, n_dead);
    if (lmax) atomicMax(len_max, lmax);
  }
  if (__ballot(irr) != 0ull && lane_id() == 0) *irregular = 1u;
  __syncthreads();
  u32* g = hist_partial + (size_t)(blockIdx.x % LIN_HIST_REPLICAS) * 1024;
  for (int k = threadIdx.x; k < 4 * 256; k += LIN_NT) {
    const u32 v = s_hist[k];
    if (v) atomicAdd(&g[k], v);
  }
}

// gbase[p][d] = exclusive scan over d of sum_r replica[r][p][d]; one block per digit p.
__global__ __launch_bounds__(256) void k_digit_offsets(const u32* __restrict__ partial, u32 n_blocks,
                                                        u32* __restrict__ gbase) {
  __shared__ u32 lds[256 / WAVE + 1];
  const int p = blockIdx.x;
  u32 c = 0;
#pragma unroll 8
  for (u32 b = 0; b < n_blocks; b++) c += partial[(size_t)b * 1024 + p * 256 + threadIdx.x];
  u32 total;
  const u32 ex = block_excl_scan<u32, 256>(c, lds, total);
  gbase[p * 256 + threadIdx.x] = ex;
}

__global__ __launch_bounds__(256) void k_digit_offsets2(const u32* __restrict__ partial_a,
                                                         const u32* __restrict__ partial_b, u32 n_blocks,
                                                         u32* __restrict__ gbase_a, u32* __restrict__ gbase_b) {
  __shared__ u32 lds[256 / WAVE + 1];
  const u32* partial = blockIdx.y ? partial_b : partial_a;
  u32* gbase = blockIdx.y ? gbase_b : gbase_a;
  const int p = blockIdx.x;
  u32 c = 0;
#pragma unroll 8
  for (u32 b = 0; b < n_blocks; b++) c += partial[(size_t)b * 1024 + p * 256 + threadIdx.x];
  u32 total;
  const u32 ex = block_excl_scan<u32, 256>(c, lds, total);
  gbase[p * 256 + threadIdx.x] = ex;
}

// ------------------------------------------------------------- range count
// 512 threads x 1 query each, 8192 staged keys (32 KB): 0.149 ms against 0.225 ms for the
// former 256 x 2 / 10240 (tools/phase_ab.sh count); two queries per thread serialise four
// dependent LDS searches
#ifndef GIQL_RC_NT
#define GIQL_RC_NT 512
#endif
#ifndef GIQL_RC_CAP
#define GIQL_RC_CAP 8192
#endif
constexpr int RC_NT = GIQL_RC_NT;
constexpr int RC_LDS_CAP = GIQL_RC_CAP;  // staged S keys
constexpr int C1_NT = 256;               // class-1 kernels keep 256-thread blocks
constexpr int RC_MARGIN = 1024;    // S entries staged past the next tile's window

// query start + signed offset, clamped to the u32 key axis
__device__ __forceinline__ u32 shift_key(u32 k, i64 off) {
  const i64 v = (i64)k + off;
  return v < 0 ? 0u : (v > (i64)U32_MAX ? U32_MAX : (u32)v);
}

// w_lo[t] = first S index a query of tile t can match = lower_bound(S, first
// query start + lo_off); w_lo[n_tiles] = |S|.  One thread per tile: the count
// kernel then starts with its S window known (no serial per-block search).
// key_mask: the queries may be ordered by their high key bits only (a query side sorted without
// its lowest digit, run_sort_onesweep skip_digits): every query of tile t then has a key >= the
// tile's first key with the unordered bits cleared, and that is where the window starts.
__global__ void k_count_partition(const u32* __restrict__ qs, u32 nq_total,
                                  const u32* __restrict__ irr_q, const u32* __restrict__ ss,
                                  u32 ns_total, const u32* __restrict__ irr_s, i64 lo_off, u32 tq,
                                  u32 n_tiles, u32* __restrict__ w_lo, u32 key_mask = 0xFFFFFFFFu) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n_tiles) return;
  const u32 nq = nq_total - *irr_q;
  const u32 ns = ns_total - *irr_s;
  const u64 q = (u64)t * tq;
  w_lo[t] = (t < n_tiles && q < nq) ? lower_bound_u32(ss, 0, ns, shift_key(qs[q] & key_mask, lo_off)) : ns;
}

// Shared by the count / emit kernels: the block's S window [w0, w0 + len) staged
// in LDS; lower_bound over [w0, ns) searches the staged part first and finishes
// in HBM only when the bound falls past it.
struct SWindow {
  const u32* __restrict__ ss;
  const u32* s_tile;
  u32 w0, len, ns, last, w_end;
  __device__ __forceinline__ void bounds(u32 xs, u32 xe, u32& lo, u32& hi) const {
    if (len && last >= xs)
      lo = w0 + lower_bound_u32(s_tile, 0, len, xs);
    else
      lo = lower_bound_u32(ss, w_end, ns, xs);
    if (lo < w_end && last >= xe)
      hi = w0 + lower_bound_u32(s_tile, lo - w0, len, xe);
    else
      hi = lower_bound_u32(ss, lo > w_end ? lo : w_end, ns, xe);
  }
};

template <int CAP, int NT>
__device__ __forceinline__ SWindow stage_window(const u32* __restrict__ ss, u32 ns, u32 w0, u32 w1,
                                                u32* s_tile) {
  u32 len = ns - w0;
  const u64 want = (u64)(w1 - w0) + RC_MARGIN;
  if (want < len) len = (u32)want;
  if (len > (u32)CAP) len = CAP;
  // 4 independent loads in flight per thread (a plain loop waits for each load)
  for (u32 k = threadIdx.x; k < len; k += 4 * NT) {
    const u32 k1 = k + NT, k2 = k + 2 * NT, k3 = k + 3 * NT;
    const u32 v0 = ss[w0 + k];
    const u32 v1 = k1 < len ? ss[w0 + k1] : 0u;
    const u32 v2 = k2 < len ? ss[w0 + k2] : 0u;
    const u32 v3 = k3 < len ? ss[w0 + k3] : 0u;
    s_tile[k] = v0;
    if (k1 < len) s_tile[k1] = v1;
    if (k2 < len) s_tile[k2] = v2;
    if (k3 < len) s_tile[k3] = v3;
  }
  __syncthreads();
  SWindow w;
  w.ss = ss;
  w.s_tile = s_tile;
  w.w0 = w0;
  w.len = len;
  w.ns = ns;
  w.last = len ? s_tile[len - 1] : 0u;
  w.w_end = w0 + len;
  return w;
}

// For the sorted queries of one block, count the points of the sorted set S in
// [qs + lo_off, qe).  lo_off = 0 is class 1 (closed low end), 1 is class 2, and
// 1 - L is the uniform-length form: when every S row has length L,
//   overlap  <=>  s.start < q.end AND s.start + L > q.start
//            <=>  s.start in [q.start - L + 1, q.end)
// -- ONE range per query row, no class split and no `end` on the S side.
// Writes lo (first matching index in S) and cnt.  Rows past the regular prefix
// (sentinel keys) get cnt = 0.  The block's S window [w_lo[t], w_lo[t+1] +
// margin) is staged in LDS and searched per lane there (LDS tile + per-lane
// binary search); a bound that falls past the staged part is finished in HBM.
template <int ITEMS, int CAP>
__global__ __launch_bounds__(RC_NT) void k_range_count(
    const u32* __restrict__ qs, const u32* __restrict__ qe, u32 nq_total,
    const u32* __restrict__ irr_q, const u32* __restrict__ ss, u32 ns_total,
    const u32* __restrict__ irr_s, i64 lo_off, const u32* __restrict__ w_lo_arr,
    u32* __restrict__ lo_out, u32* __restrict__ cnt_out) {
  constexpr u32 TQ = RC_NT * ITEMS;
  __shared__ u32 s_tile[CAP];
  const u32 nq = nq_total - *irr_q;
  const u32 ns = ns_total - *irr_s;
  const u32 bid = blockIdx.x;
  const u32 q0 = bid * TQ;
  const u32 tid = threadIdx.x;
  u32 xs[ITEMS], xe[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const u32 q = q0 + i * RC_NT + tid;
    const bool ok = q < nq;
    xs[i] = ok ? shift_key(qs[q], lo_off) : U32_MAX;
    xe[i] = ok ? qe[q] : U32_MAX;
  }
  const SWindow w = stage_window<CAP, RC_NT>(ss, ns, w_lo_arr[bid], w_lo_arr[bid + 1], s_tile);
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const u32 q = q0 + i * RC_NT + tid;
    if (q >= nq_total) continue;
    u32 lo = 0, hi = 0;
    if (q < nq) w.bounds(xs[i], xe[i], lo, hi);
    lo_out[q] = lo;
    cnt_out[q] = hi - lo;
  }
}

// ---- class 1 without per-row arrays -------------------------------------------
// Class-1 queries are the B rows (the big side) and most of them match nothing,
// so materialising (lo, cnt, offset) per row costs more HBM traffic than the pairs
// they describe.  Instead: k_c1_count writes one total per block, a tiny scan
// turns those into block bases, and k_c1_emit recomputes the bounds and writes the
// pairs straight away.  Each thread owns C1_ITEMS CONSECUTIVE sorted rows (16-byte
// loads), so after one binary search the next bound is found by galloping forward
// from the previous one: ~2 LDS probes per bound instead of ~11.
// C1_ITEMS = 8 for sides of millions of rows (2048 B rows per block); 2 for small sides, where 8 rows
// per thread leave two waves per SIMD and the chain search -> gather -> store of one thread is
// what the kernel waits for.
constexpr int C1_ITEMS_MAX = 8;
constexpr int C1_CAP = 3072;             // staged A starts (12 KB)
constexpr u32 C1_COOP = 16;              // matches per row above which a wave co-writes

// first idx in [from, len) with v[idx] >= x, given v[i] < x for all i < from
__device__ __forceinline__ u32 gallop_lb(const u32* v, u32 len, u32 from, u32 x) {
  if (from >= len || v[from] >= x) return from;
  u32 lo = from, step = 1;  // v[lo] < x
  while (lo + step < len && v[lo + step] < x) {
    lo += step;
    step <<= 1;
  }
  u32 hi = lo + step < len ? lo + step : len;  // v[hi] >= x, or hi == len
  lo += 1;
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (v[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// Bounds of C1_ITEMS consecutive sorted queries against the staged window.
// prev = staged-relative lower bound of the previous query (0 for the first).
template <int C1_ITEMS>
struct C1Bounds {
  u32 lo[C1_ITEMS], cnt[C1_ITEMS];
};

template <int C1_ITEMS>
__device__ __forceinline__ void c1_bounds(const SWindow& w, const u32 (&xs)[C1_ITEMS],
                                          const u32 (&xe)[C1_ITEMS], u32 q_first, u32 nq,
                                          C1Bounds<C1_ITEMS>& out) {
  u32 prev = 0;
#pragma unroll
  for (int i = 0; i < C1_ITEMS; i++) {
    out.lo[i] = 0;
    out.cnt[i] = 0;
    if (q_first + i >= nq) continue;
    u32 rel = i == 0 ? lower_bound_u32(w.s_tile, 0, w.len, xs[0]) : gallop_lb(w.s_tile, w.len, prev, xs[i]);
    prev = rel;
    u32 lo, hi;
    if (rel < w.len || w.w_end >= w.ns) {
      lo = w.w0 + rel;
      const u32 r2 = gallop_lb(w.s_tile, w.len, rel, xe[i]);
      hi = (r2 < w.len || w.w_end >= w.ns) ? w.w0 + r2 : lower_bound_u32(w.ss, w.w_end, w.ns, xe[i]);
    } else {  // both bounds lie past the staged window: finish in HBM
      lo = lower_bound_u32(w.ss, w.w_end, w.ns, xs[i]);
      hi = lower_bound_u32(w.ss, lo, w.ns, xe[i]);
    }
    out.lo[i] = lo;
    out.cnt[i] = hi - lo;
  }
}

template <int C1_ITEMS>
__device__ __forceinline__ void c1_load8(const u32* __restrict__ p, u32 q, u32 n_ok, u32 fill,
                                         u32 (&x)[C1_ITEMS]) {
  static_assert(C1_ITEMS == 2 || C1_ITEMS % 4 == 0, "vector loads of 2 or 4 words");
  if (q + C1_ITEMS <= n_ok) {
    if constexpr (C1_ITEMS == 2) {
      const uint2 t = *reinterpret_cast<const uint2*>(p + q);
      x[0] = t.x; x[1] = t.y;
    } else {
#pragma unroll
      for (int k = 0; k < C1_ITEMS / 4; k++) {
        const uint4 t = *reinterpret_cast<const uint4*>(p + q + 4 * k);
        x[4 * k] = t.x; x[4 * k + 1] = t.y; x[4 * k + 2] = t.z; x[4 * k + 3] = t.w;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < C1_ITEMS; i++) x[i] = (q + i < n_ok) ? p[q + i] : fill;
  }
}

template <int C1_ITEMS>
__global__ __launch_bounds__(C1_NT) void k_c1_count(
    const u32* __restrict__ qs, const u32* __restrict__ qe, u32 nq_total,
    const u32* __restrict__ irr_q, const u32* __restrict__ ss, u32 ns_total,
    const u32* __restrict__ irr_s, const u32* __restrict__ w_lo_arr, u64* __restrict__ block_sum) {
  __shared__ u32 s_tile[C1_CAP];
  __shared__ u64 s_red[C1_NT / WAVE];
  const u32 nq = nq_total - *irr_q;
  const u32 ns = ns_total - *irr_s;
  const u32 tid = threadIdx.x;
  const u32 q = blockIdx.x * (C1_NT * C1_ITEMS) + tid * C1_ITEMS;
  u32 xs[C1_ITEMS], xe[C1_ITEMS];
  c1_load8(qs, q, nq, U32_MAX, xs);
  c1_load8(qe, q, nq, U32_MAX, xe);
  const SWindow w = stage_window<C1_CAP, C1_NT>(ss, ns, w_lo_arr[blockIdx.x], w_lo_arr[blockIdx.x + 1], s_tile);
  C1Bounds<C1_ITEMS> b;
  c1_bounds(w, xs, xe, q, nq, b);
  u64 total = 0;
#pragma unroll
  for (int i = 0; i < C1_ITEMS; i++) total += b.cnt[i];
  total = wave_reduce_sum(total);
  if (lane_id() == 0) s_red[wave_id()] = total;
  __syncthreads();
  if (tid == 0) {
    u64 t = 0;
#pragma unroll
    for (int k = 0; k < C1_NT / WAVE; k++) t += s_red[k];
    block_sum[blockIdx.x] = t;
  }
}

// block_base[] = exclusive scan of block_sum.  Output slots of a block are given
// out thread by thread (row order), so every slot is written exactly once; rows
// with many matches are written cooperatively by their wave (coalesced).
template <int C1_ITEMS>
__global__ __launch_bounds__(C1_NT) void k_c1_emit(
    const u32* __restrict__ qs, const u32* __restrict__ qe, const u32* __restrict__ q_rid,
    u32 nq_total, const u32* __restrict__ irr_q, const u32* __restrict__ ss,
    const u32* __restrict__ s_rid, u32 ns_total, const u32* __restrict__ irr_s,
    const u32* __restrict__ w_lo_arr, const u64* __restrict__ block_base, u64 out_base,
    int32_t* __restrict__ row_q, int32_t* __restrict__ row_s) {
  __shared__ u32 s_tile[C1_CAP];
  __shared__ u64 s_scan[C1_NT / WAVE + 1];
  const u32 nq = nq_total - *irr_q;
  const u32 ns = ns_total - *irr_s;
  const u32 tid = threadIdx.x;
  const u32 q = blockIdx.x * (C1_NT * C1_ITEMS) + tid * C1_ITEMS;
  u32 xs[C1_ITEMS], xe[C1_ITEMS], rid[C1_ITEMS];
  c1_load8(qs, q, nq, U32_MAX, xs);
  c1_load8(qe, q, nq, U32_MAX, xe);
  c1_load8(q_rid, q, nq, 0u, rid);
  const SWindow w = stage_window<C1_CAP, C1_NT>(ss, ns, w_lo_arr[blockIdx.x], w_lo_arr[blockIdx.x + 1], s_tile);
  C1Bounds<C1_ITEMS> b;
  c1_bounds(w, xs, xe, q, nq, b);
  u64 mine = 0;
#pragma unroll
  for (int i = 0; i < C1_ITEMS; i++) mine += b.cnt[i];
  u64 total;
  u64 o = out_base + block_base[blockIdx.x] + block_excl_scan<u64, C1_NT>(mine, s_scan, total);
  if (total == 0) return;  // block-uniform
  // most rows match 0-2 points: fetch those ids for all rows first (16 gathers
  // in flight) instead of one dependent gather per pair
  u32 g0[C1_ITEMS], g1[C1_ITEMS];
#pragma unroll
  for (int i = 0; i < C1_ITEMS; i++) {
    g0[i] = b.cnt[i] > 0 ? s_rid[b.lo[i]] : 0u;
    g1[i] = b.cnt[i] > 1 ? s_rid[b.lo[i] + 1] : 0u;
  }
#pragma unroll
  for (int i = 0; i < C1_ITEMS; i++) {
    const u32 c = b.cnt[i];
    const bool big = c > C1_COOP;
    if (!big) {
      if (c > 0) {
        row_q[o] = (int32_t)rid[i];
        row_s[o] = (int32_t)g0[i];
      }
      if (c > 1) {
        row_q[o + 1] = (int32_t)rid[i];
        row_s[o + 1] = (int32_t)g1[i];
      }
      for (u32 k = 2; k < c; k++) {
        row_q[o + k] = (int32_t)rid[i];
        row_s[o + k] = (int32_t)s_rid[b.lo[i] + k];
      }
    }
    // rows with many matches: the whole wave writes them, 64 pairs per step
    u64 m = __ballot(big);
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const u32 c2 = __shfl(c, src, WAVE);
      const u32 lo2 = __shfl(b.lo[i], src, WAVE);
      const u32 rid2 = __shfl(rid[i], src, WAVE);
      const u64 o2 = __shfl(o, src, WAVE);
      for (u32 k = lane_id(); k < c2; k += WAVE) {
        row_q[o2 + k] = (int32_t)rid2;
        row_s[o2 + k] = (int32_t)s_rid[lo2 + k];
      }
    }
    o += c;
  }
}

// --------------------------------------------------------------- partition
#ifndef GIQL_FILL_NT
#define GIQL_FILL_NT 1024  // 16384-pair tiles: 0.88 -> 0.80 ms against 256 (4096-pair tiles), tools/fill_ab.sh
#endif
constexpr int FILL_NT = GIQL_FILL_NT;
constexpr int FILL_QCAP = 4 * FILL_NT;  // query records staged per output tile (a quarter of its pairs)

// Merge-path split of one class's output among blocks: every block materialises
// exactly `tile` pairs.  part[t] = last query q (relative to q_base) whose
// offset is <= out_base + t*tile;  part[n_tiles] = nq - 1.
// n_out_dev (optional): the number of outputs is read from the device and n_tiles is
// only an upper bound (a fill launched before the host has learned the count); a count above
// `cap` (the caller's buffers) means that early fill is void: nothing is partitioned.
__global__ void k_partition(const u64* __restrict__ off, u32 nq, u64 out_base, u32 tile,
                            u32 n_tiles, u32* __restrict__ part,
                            const u64* __restrict__ n_out_dev = nullptr, u64 cap = ~0ull) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (n_out_dev) {
    if (*n_out_dev > cap) return;
    const u64 nt = (*n_out_dev + tile - 1) / tile;
    n_tiles = nt < (u64)n_tiles ? (u32)nt : n_tiles;
  }
  if (t > n_tiles) return;
  if (t == n_tiles) {
    part[t] = nq > 0 ? nq - 1 : 0u;
    return;
  }
  const u64 p = out_base + (u64)t * tile;
  part[t] = (u32)(upper_bound_u64(off, 0, (u64)nq + 1, p) - 1);
}

// -------------------------------------------------------------------- fill
// One 16384-pair tile per block (1024 threads x 16).  off / lo / q_rid are the query arrays, s_rid the
// other side's sorted row ids.  Outputs [out_base, out_base + n_out) go to row_q
// (the query side's ids) and row_s.
//
// The tile's query records {relative offset, lo, rid} are staged in LDS.  Each
// wave owns TILE/4 consecutive outputs and walks them 64 at a time: the rows that
// START inside the current 64-output window set the bit of their start position
// in a 64-bit mask (DPP OR-reduction); the row of output l is then the popcount of
// the mask bits 0..l -- a merge of the two sorted sequences by ballot/scan, well
// under one instruction per pair, instead of a binary search per pair.
// Per pair: one gather of s_rid (coalesced inside a row) + two coalesced stores.

// Slow path of one window (more than 63 rows start in it, or rows without matches
// share a start position): markers through LDS + max-scan.  Out of line: rare.
__device__ __noinline__ u32 fill_window_slow(const u32* s_rel, u32* mark, u32 nqt, u32 k_cur,
                                             u32 pc) {
  const u32 lane = lane_id();
  mark[lane] = 0;
  __builtin_amdgcn_wave_barrier();
  u32 base = k_cur + 1;
  while (true) {
    const u32 kk = base + lane;
    const bool in = kk < nqt && s_rel[kk] < pc + WAVE;
    if (in) atomicMax(&mark[s_rel[kk] - pc], kk - k_cur);
    if (__ballot(in) != ~0ull) break;  // fewer than 64 candidates qualified: done
    base += WAVE;
  }
  __builtin_amdgcn_wave_barrier();
  return wave_incl_scan_max_u32(mark[lane]);
}

template <int ITEMS, bool FULL>
__device__ __forceinline__ void fill_wave(const u32* s_rel, const u32* s_jbase,
                                          const u32* s_qrid, u32* mark, u32 nqt, u32 tile_len,
                                          u64 tile_start, const u32* __restrict__ s_rid,
                                          int32_t* __restrict__ row_q, int32_t* __restrict__ row_s) {
  constexpr u32 TILE = FILL_NT * ITEMS;
  constexpr u32 PER_WAVE = TILE / (FILL_NT / WAVE);
  constexpr int NWIN = PER_WAVE / WAVE;
  const u32 lane = lane_id();
  const u32 p_w0 = wave_id() * PER_WAVE;
  if (!FULL && p_w0 >= tile_len) return;  // wave-uniform
  // last row whose outputs start at or before this wave's first output (row 0 of
  // the tile starts at or before the tile, so the search is over rows 1..)
  u32 k_cur = upper_bound_u32(s_rel, 1, nqt, p_w0) - 1;
  u32 qr[NWIN], sr[NWIN];
  int32_t* rq = row_q + tile_start;
  int32_t* rs = row_s + tile_start;
#pragma unroll
  for (int it = 0; it < NWIN; it++) {
    const u32 pc = p_w0 + it * WAVE;
    qr[it] = 0;
    sr[it] = 0;
    if (!FULL && pc >= tile_len) continue;  // wave-uniform
    // candidates: rows k_cur+1 .. k_cur+64; in-window ones have rel in [pc, pc+64)
    const u32 kk0 = k_cur + 1 + lane;
    const u32 r0 = kk0 < nqt ? s_rel[kk0] : U32_MAX;
    const bool in0 = r0 < pc + WAVE;
    const u64 m0 = __ballot(in0);
    u32 kd;
    // The in-window rows are a prefix of the lanes (starts are sorted).  With few of
    // them -- the common case once rows average more than ~16 matches -- each is
    // broadcast with one v_readlane and compared: ~3 VALU per row instead of the
    // ~30 of the mask build below.  Rows without matches (shared starts) need no
    // special care here: every row starting at or before the position counts.
    const u32 n_in = (u32)__popcll(m0);
    if (n_in <= 4u) {  // wave-uniform
      const int ri = (int)(r0 - pc);  // in-window rows: 0..63
      kd = 0;
      if (n_in > 0u) kd += (u32)((int)lane >= __builtin_amdgcn_readlane(ri, 0));
      if (n_in > 1u) kd += (u32)((int)lane >= __builtin_amdgcn_readlane(ri, 1));
      if (n_in > 2u) kd += (u32)((int)lane >= __builtin_amdgcn_readlane(ri, 2));
      if (n_in > 3u) kd += (u32)((int)lane >= __builtin_amdgcn_readlane(ri, 3));
    } else {
      // rows without matches share their successor's start: sorted, so compare with
      // the previous lane (wave_shr:1 crosses the 16-lane rows on gfx9)
      const u32 rprev = (u32)__builtin_amdgcn_update_dpp((int)(pc - 1u), (int)r0, 0x138, 0xF, 0xF, false);
      const u64 mdup = __ballot(in0 && rprev == r0);
      if (m0 != ~0ull && mdup == 0) {  // wave-uniform fast path
        const u64 bit = in0 ? (1ull << (r0 - pc)) : 0ull;
        u32 blo = (u32)bit, bhi = (u32)(bit >> 32);
#define GIQL_OR_STEP(ctrl, rm)                                                  \
  blo |= (u32)__builtin_amdgcn_update_dpp(0, (int)blo, ctrl, rm, 0xF, false);  \
  bhi |= (u32)__builtin_amdgcn_update_dpp(0, (int)bhi, ctrl, rm, 0xF, false);
        GIQL_OR_STEP(GIQL_DPP_ROW_SHR(1), 0xF)
        GIQL_OR_STEP(GIQL_DPP_ROW_SHR(2), 0xF)
        GIQL_OR_STEP(GIQL_DPP_ROW_SHR(4), 0xF)
        GIQL_OR_STEP(GIQL_DPP_ROW_SHR(8), 0xF)
        GIQL_OR_STEP(GIQL_DPP_ROW_BCAST15, 0xA)
        GIQL_OR_STEP(GIQL_DPP_ROW_BCAST31, 0xC)
#undef GIQL_OR_STEP
        const u32 mlo = (u32)__builtin_amdgcn_readlane((int)blo, WAVE - 1);
        const u32 mhi = (u32)__builtin_amdgcn_readlane((int)bhi, WAVE - 1);
        // rows starting at positions <= my lane: mbcnt counts mask bits BELOW the
        // lane, so add my own position's bit
        kd = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u)) +
             (u32)((lane < 32 ? (mlo >> lane) : (mhi >> (lane - 32))) & 1u);
      } else {
        kd = fill_window_slow(s_rel, mark, nqt, k_cur, pc);
      }
    }
    const u32 k = k_cur + kd;
    const u32 p_rel = pc + lane;
    // rel of row 0 is stored signed (-(tile_start - off[first row])): one formula
    const u32 j = s_jbase[k] + p_rel;  // = lo[k] + (p_rel - rel0[k])
    qr[it] = s_qrid[k];
    // issued now, waited for at the stores: flies under the following windows
    if (FULL || p_rel < tile_len) sr[it] = s_rid[j];
    k_cur = (u32)__builtin_amdgcn_readlane((int)k, WAVE - 1);
  }
#pragma unroll
  for (int it = 0; it < NWIN; it++) {
    const u32 p_rel = p_w0 + it * WAVE + lane;
    if (FULL || p_rel < tile_len) {
#if defined(GIQL_FILL_STORE_NT)  // probe: non-temporal output stores
      __builtin_nontemporal_store((int32_t)qr[it], rq + p_rel);
      __builtin_nontemporal_store((int32_t)sr[it], rs + p_rel);
#else
      rq[p_rel] = (int32_t)qr[it];
      rs[p_rel] = (int32_t)sr[it];
#endif
    }
  }
}

template <int ITEMS>
__global__ __launch_bounds__(FILL_NT) void k_fill(
    const u64* __restrict__ off, const u32* __restrict__ lo, const u32* __restrict__ q_rid, u32 nq,
    const u32* __restrict__ s_rid, const u32* __restrict__ part, u64 out_base, u64 n_out,
    int32_t* __restrict__ row_q, int32_t* __restrict__ row_s,
    const u64* __restrict__ n_out_dev = nullptr, u64 cap = ~0ull) {
  constexpr u32 TILE = FILL_NT * ITEMS;
  if (n_out_dev) {  // launched with an upper-bound grid before the host knew the count
    n_out = *n_out_dev;
    // more pairs than the caller's buffers hold: the early fill is discarded by the host
    // (GIQL_ERR_CAPACITY), and its last tile would store past the end -- write nothing
    if (n_out > cap || (u64)blockIdx.x * TILE >= n_out) return;
  }
  __shared__ u32 s_rel[FILL_QCAP];  // unsigned relative starts; [0] unused by searches
  // lo[k] - rel0[k] (mod 2^32), rel0 = the relative start with its true (<= 0) value for row 0:
  // the S index of output p of row k is s_jbase[k] + p
  __shared__ u32 s_jbase[FILL_QCAP];
  __shared__ u32 s_qrid[FILL_QCAP];
  __shared__ u32 s_mark[FILL_NT / WAVE][WAVE];
  const u32 tid = threadIdx.x;
  // (an XCD-aware block -> tile map was measured here and in k_range_count: no gain)
  const u32 bid = blockIdx.x;
  const u64 tile_rel = (u64)bid * TILE;  // relative to out_base
  const u64 tile_start = out_base + tile_rel;
  const u64 rem = n_out - tile_rel;
  const u32 tile_len = rem < (u64)TILE ? (u32)rem : TILE;
  const u32 qf = part[bid];
  u32 ql = part[bid + 1];
  if (ql >= nq) ql = nq - 1;
  const u32 nqt = ql - qf + 1;
  // The records of the tile's query rows are loaded in the SAME round trip as off[qf] (which
  // decides whether the staged form applies): a tile starts with a chain of dependent global
  // loads -- part, off[qf], the records -- and with two 1024-thread blocks per CU every
  // microsecond of that chain is a slot standing idle (~15 us per tile).
  constexpr int STAGE_IT = FILL_QCAP / FILL_NT;
  const bool fits = nqt <= (u32)FILL_QCAP;
  const u64 off_first = off[qf];
  u64 ov[STAGE_IT];
  u32 lv[STAGE_IT], rv[STAGE_IT];
  if (fits) {
#pragma unroll
    for (int it = 0; it < STAGE_IT; it++) {
      const u32 k = tid + it * FILL_NT;
      const bool ok = k < nqt;
      ov[it] = ok ? off[qf + k] : 0;
      lv[it] = ok ? lo[qf + k] : 0;
      rv[it] = ok ? q_rid[qf + k] : 0;
    }
  }
  const u64 first_delta = tile_start - off_first;
  const bool staged = fits && first_delta < 0x7FFFFFFFull;
  if (staged) {
#pragma unroll
    for (int it = 0; it < STAGE_IT; it++) {
      const u32 k = tid + it * FILL_NT;
      if (k < nqt) {
        const u64 o = ov[it];
        u32 r = 0;
        if (o > tile_start) {
          const u64 d = o - tile_start;
          r = d > (u64)tile_len ? tile_len : (u32)d;
        }
        s_rel[k] = r;
        s_jbase[k] = lv[it] - (k == 0 ? (u32)(-(int)(u32)first_delta) : r);
        s_qrid[k] = rv[it];
      }
    }
    __syncthreads();
    if (tile_len == TILE)
      fill_wave<ITEMS, true>(s_rel, s_jbase, s_qrid, s_mark[wave_id()], nqt, tile_len,
                             tile_start, s_rid, row_q, row_s);
    else
      fill_wave<ITEMS, false>(s_rel, s_jbase, s_qrid, s_mark[wave_id()], nqt, tile_len,
                              tile_start, s_rid, row_q, row_s);
    return;
  }
  // fallback (more rows than the LDS stage holds, e.g. long runs of empty rows):
  // one binary search per pair straight on the offsets
#pragma unroll 4
  for (int i = 0; i < ITEMS; i++) {
    const u32 p_rel = i * FILL_NT + tid;
    if (p_rel < tile_len) {
      const u64 p = tile_start + p_rel;
      const u32 q = (u32)(upper_bound_u64(off, qf, (u64)ql + 1, p) - 1);
      const u32 j = lo[q] + (u32)(p - off[q]);
      row_q[p] = (int32_t)q_rid[q];
      row_s[p] = (int32_t)s_rid[j];
    }
  }
}

// ------------------------------------------------ irregular rows (literal)
struct SideView {
  const int* chrom;
  const int* start;
  const int* end;
  u32 n;
  int start_off, end_off;
};

__device__ __forceinline__ bool literal_overlap(int ac, i64 as, i64 ae, int bc, i64 bs, i64 be) {
  return ac == bc && as < be && ae > bs;
}

// Pairs involving an irregular row, each counted once:
//   part X: (irregular a) x (every b)          -- thread per B row
//   part Y: (regular a)   x (irregular b)      -- thread per A row
// cnt has n_b + n_a entries [X | Y].
__global__ void k_irr_count(SideView a, SideView b, const u32* __restrict__ irr_a_list,
                            const u32* __restrict__ irr_b_list, const DevMeta* __restrict__ meta,
                            u32* __restrict__ cnt) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.n + b.n) return;
  u32 c = 0;
  if (t < b.n) {
    const int bc = b.chrom[t];
    const i64 bs = (i64)b.start[t] + b.start_off, be = (i64)b.end[t] + b.end_off;
    const u32 m = meta->irr_a;
    for (u32 k = 0; k < m; k++) {
      const u32 r = irr_a_list[k];
      c += literal_overlap(a.chrom[r], (i64)a.start[r] + a.start_off, (i64)a.end[r] + a.end_off, bc,
                           bs, be);
    }
  } else {
    const u32 i = t - b.n;
    const int ac = a.chrom[i];
    const i64 as = (i64)a.start[i] + a.start_off, ae = (i64)a.end[i] + a.end_off;
    if (ae > as) {
      const u32 m = meta->irr_b;
      for (u32 k = 0; k < m; k++) {
        const u32 r = irr_b_list[k];
        c += literal_overlap(ac, as, ae, b.chrom[r], (i64)b.start[r] + b.start_off,
                             (i64)b.end[r] + b.end_off);
      }
    }
  }
  cnt[t] = c;
}

__global__ void k_irr_fill(SideView a, SideView b, const u32* __restrict__ irr_a_list,
                           const u32* __restrict__ irr_b_list, const DevMeta* __restrict__ meta,
                           const u64* __restrict__ off, int32_t* __restrict__ row_a,
                           int32_t* __restrict__ row_b) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.n + b.n) return;
  u64 o = off[t];
  if (t < b.n) {
    const int bc = b.chrom[t];
    const i64 bs = (i64)b.start[t] + b.start_off, be = (i64)b.end[t] + b.end_off;
    const u32 m = meta->irr_a;
    for (u32 k = 0; k < m; k++) {
      const u32 r = irr_a_list[k];
      if (literal_overlap(a.chrom[r], (i64)a.start[r] + a.start_off, (i64)a.end[r] + a.end_off, bc,
                          bs, be)) {
        row_a[o] = (int32_t)r;
        row_b[o] = (int32_t)t;
        o++;
      }
    }
  } else {
    const u32 i = t - b.n;
    const int ac = a.chrom[i];
    const i64 as = (i64)a.start[i] + a.start_off, ae = (i64)a.end[i] + a.end_off;
    if (ae > as) {
      const u32 m = meta->irr_b;
      for (u32 k = 0; k < m; k++) {
        const u32 r = irr_b_list[k];
        if (literal_overlap(ac, as, ae, b.chrom[r], (i64)b.start[r] + b.start_off,
                            (i64)b.end[r] + b.end_off)) {
          row_a[o] = (int32_t)i;
          row_b[o] = (int32_t)r;
          o++;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- checksum
__global__ __launch_bounds__(256) void k_pairs_checksum(const int32_t* __restrict__ row_a,
                                                         const int32_t* __restrict__ row_b, u64 n,
                                                         u64* __restrict__ out) {
  __shared__ u64 lds[256 / WAVE];
  u64 acc = 0;
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    u64 x = ((u64)(u32)row_a[i] << 32) | (u64)(u32)row_b[i];
    x *= 0x9E3779B97F4A7C15ull;
    x ^= (x >> 32);
    x *= 0xD6E8FEB86659FD93ull;
    acc += x;
  }
  acc = wave_reduce_sum(acc);
  if (lane_id() == 0) lds[wave_id()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
#pragma unroll
    for (int w = 0; w < 256 / WAVE; w++) t += lds[w];
    atomicAdd((unsigned long long*)out, (unsigned long long)t);
  }
}

}  // namespace giql
