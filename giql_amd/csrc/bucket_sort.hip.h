// giql_amd/csrc/bucket_sort.hip.h -- last stage of the three-stage sort.
//
// The four-pass LSD sort (onesweep.hip.h) moves every row through HBM four times.  Here only
// the two HIGH digits go through global scatter passes (bits 16-23, then 24-31): after them the
// rows sharing key >> 16 -- a "bucket", a few thousand rows of a genome-scale table -- sit
// contiguously, in input order.  The low 16 bits are then sorted bucket by bucket INSIDE LDS
// (k_bucket_sort), read once and written once, in place: three trips through HBM instead of
// four.  This is what the 160 KB of LDS per CU is for: a 4096-row bucket with its bin table
// takes 32 KB, so four 512-thread blocks share a CU.
//
// Stable (ties keep their input order), like the onesweep passes, so a sort built from the two
// stages is a stable sort by the full 32-bit key and can serve as one leg of a two-key sort.
//
// A bucket larger than BS_CAP rows (pile-ups: real read tables have windows far denser than the
// average) does not fit the LDS form; its block sorts it with a small two-pass LSD radix sort
// through the sort's OTHER ping-pong buffer instead (bucket_sort_big: tile by tile, stable, ~10x
// the cost per row -- paid by the hot buckets only).  Past BS_BIG_MAX rows in one bucket (a table
// squeezed into a few 65536-wide windows) that serial form would dominate: the block sets
// meta->status = GIQL_STATUS_RESORT and the host repeats the call with the four-pass sort (and
// keeps using it on that context) -- the kernels downstream of a failed sort are memory-safe on
// unsorted keys, exactly as they are for the context's other wrong guesses.
#pragma once

#include "dev_common.hip.h"
#include "onesweep.hip.h"

namespace giql {

constexpr u32 BS_BUCKETS = 1u << 16;
// Bucket width by density (round 4): a bucket is the rows sharing key >> W.  W = 16 for tables of up to ~2,800 rows
// per 65,536 keys; denser tables (0.13-1 G rows on a human-genome axis) take W = 15 / 14 / 13, so that the average
// bucket stays within what the LDS stage holds -- after THREE global passes (bits 8-15, 16-23, 24-31: the rows come
// grouped by key >> 8, which groups them by key >> W for every W >= 8) instead of two.  Inside the kernels the low W
// key bits are scaled up to 16 (key << (16 - W)), so the bin table, the ranks and the packed words of the body are
// the same for every width; W travels in BsFuse::wbits.
constexpr int BS_MIN_WBITS = 13;
constexpr u32 BS_MAX_BUCKETS = 1u << (32 - BS_MIN_WBITS);
__host__ __device__ __forceinline__ constexpr u32 bs_n_buckets(u32 wbits) { return 1u << (32u - wbits); }
constexpr int BS_NT = 512;
constexpr int BS_ITEMS = 8;
constexpr int BS_NW = BS_NT / WAVE;
constexpr u32 BS_CAP = BS_NT * BS_ITEMS;  // rows per bucket the LDS sort holds (12-bit local index)
constexpr int GIQL_STATUS_RESORT = -100;  // internal: never crosses the C ABI
#ifndef GIQL_BS_BIG_MAX
#define GIQL_BS_BIG_MAX (1u << 18)
#endif
constexpr u32 BS_BIG_MAX = GIQL_BS_BIG_MAX;  // rows of one bucket the in-block global-memory sort takes

// bnd[v] = first row of the top-16-sorted keys with key >= v << 16, v in [0, 65536]; the search
// runs inside the top digit's row range, which the sort's own digit offsets give (gb3 = the 256
// exclusive offsets of the bits 24-31 digit).
__global__ __launch_bounds__(256) void k_bucket_bounds(const u32* __restrict__ keys, u32 n,
                                                        const u32* __restrict__ gb3,
                                                        u32* __restrict__ bnd, u32 wbits) {
  const u32 v = blockIdx.x * 256 + threadIdx.x;
  const u32 nbk = bs_n_buckets(wbits);
  if (v > nbk) return;
  if (v == nbk) {
    bnd[v] = n;
    return;
  }
  const u32 d3 = v >> (24u - wbits);
  const u32 lo = gb3[d3];
  const u32 hi = d3 == 255u ? n : gb3[d3 + 1];
  bnd[v] = lower_bound_u32(keys, lo < n ? lo : n, hi < n ? hi : n, v << wbits);
}

// ---- the range count of the fixed-length INNER form, fused into the bucket sort (round 3) ----
// In that form (giql_hip.hip, inner_plan_core) every query row q matches the rows of the sorted side U
// whose key lies in [q.key + lo_off, q.end): two lower bounds per query over U's sorted keys.  The block
// that sorts bucket v HAS the bucket's keys in LDS -- as the scanned bin table of bucket_sort_body, from
// which the rank of any 16-bit value is ONE cell read -- so it answers every bound that falls into its
// bucket and U's sorted keys never go back to HBM (only the row ids do): the count kernel, its read of
// the 100M keys and their write disappear.  The queries whose lower bound (q.key + lo_off) or upper bound
// (q.end <= q.key + len_max_q) can fall into bucket v = [K0, K1) have q.key in [K0 - len_max_q, K1 - lo_off):
// a contiguous window of the start-sorted queries, found per bucket by k_bucket_bounds_fused.  Every
// regular query gets its `lo` from exactly one block and its `hi` from exactly one block.
constexpr u32 BS_FUSE_WCAP = 1u << 15;  // longest query the windows allow for (the host takes the fused form only below it)
struct BsFuse {
  const u32* qkey;   // the query side's sorted keys (ordered at least by key & key_mask) ...
  const u32* qend;   // ... and end keys
  u32* qwin;         // [2 * BS_BUCKETS] {first, past-last} sorted query row whose bounds may fall into bucket v
  u32* lo_out;       // per sorted query row: first matching sorted U row ...
  u32* hi_out;       // ... and one past the last
  i64 lo_off;
  // FUSE == 2 (the join itself in the bucket stage, see bucket_join_emit): the pairs leave from here
  const u32* qrid;   // the sorted query rows' ids
  int32_t* row_q;    // output: query-side row id of every pair ...
  int32_t* row_s;    // ... and the sorted side's
  u64 cap;           // pairs the outputs hold
  unsigned long long* cursor;  // pairs handed out so far (DevMeta::n_out): a block takes its output range with ONE atomic add
  u32 wbits = 16;    // key bits of a bucket (BS_MIN_WBITS .. 16)
};

__device__ __forceinline__ u32 bs_shift_key(u32 k, i64 off) {  // = shift_key of join_kernels.hip.h
  const i64 x = (i64)k + off;
  return x < 0 ? 0u : (x > (i64)U32_MAX ? U32_MAX : (u32)x);
}

// k_bucket_bounds + the query windows of the fused count, one thread per value: threads [0, 65536] the
// bucket boundaries, the next 65536 the windows' first rows, the last 65536 their ends.  gbq3 = the query
// sort's own top-digit offsets (they narrow each search to 1/256 of the rows); key_mask as in
// k_count_partition (a query side sorted without its lowest digit is ordered by key & 0xFFFFFF00 only).
__global__ __launch_bounds__(256) void k_bucket_bounds_fused(const u32* __restrict__ keys, u32 n,
                                                              const u32* __restrict__ gb3,
                                                              u32* __restrict__ bnd, u32* __restrict__ big_list,
                                                              BsFuse fq, u32 nq_total,
                                                              const u32* __restrict__ irr_q,
                                                              const u32* __restrict__ gbq3, u32 key_mask,
                                                              const int* __restrict__ len_max_q,
                                                              u32* __restrict__ zero_ptr, u32 zero_words,
                                                              const int* __restrict__ len_max_u = nullptr) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t == 0) big_list[0] = 0;  // the queue of buckets too large for LDS starts empty
  // the status words + ticket of the chained scan that follows the bucket sort (scan.hip.h) start at zero
  for (u32 i = t; i < zero_words; i += gridDim.x * 256) zero_ptr[i] = 0u;
  const u32 wbits = fq.wbits, nbk = bs_n_buckets(wbits);
  if (t <= nbk) {
    if (t == nbk) {
      bnd[t] = n;
      return;
    }
    const u32 d3 = t >> (24u - wbits);
    const u32 lo = gb3[d3];
    const u32 hi = d3 == 255u ? n : gb3[d3 + 1];
    bnd[t] = lower_bound_u32(keys, lo < n ? lo : n, hi < n ? hi : n, t << wbits);
    return;
  }
  const u32 u = t - (nbk + 1);
  if (u >= 2 * nbk) return;
  const u32 v = u & (nbk - 1);
  const bool upper = u >= nbk;
  const u32 nq = nq_total - *irr_q;  // the regular rows: a sorted prefix
  const u32 k0 = v << wbits;
  u32 target;  // lower: first row with masked key >= target; upper: first row with masked key > target
  if (!upper) {
    int lm = *len_max_q;
    const u32 w = lm < 0 ? 0u : ((u32)lm > BS_FUSE_WCAP ? BS_FUSE_WCAP : (u32)lm);
    target = k0 > w ? k0 - w : 0u;
  } else {
    // fixed-length form (lo_off <= 0): query keys up to K1 - 1 - lo_off put their lower bound into the bucket; general
    // form (len_max_u given): the queries double as the POINTS of the bucket rows' own ranges [u.start, u.end), which
    // reach up to the longest row above the bucket
    u64 reach = (u64)(-fq.lo_off);
    if (len_max_u) {
      const int lu = *len_max_u;
      reach = lu < 0 ? 0ull : ((u32)lu > BS_FUSE_WCAP ? (u64)BS_FUSE_WCAP : (u64)lu);
    }
    const u64 tmax = (u64)k0 + (u64)((1u << wbits) - 1u) + reach;
    if (tmax >= (u64)U32_MAX) {
      fq.qwin[2 * v + 1] = nq;
      return;
    }
    target = (u32)tmax;
  }
  const u32 tm = target & key_mask;
  const u32 d3 = tm >> 24;
  u32 lo = gbq3[d3];
  u32 hi = d3 == 255u ? nq : gbq3[d3 + 1];
  lo = lo < nq ? lo : nq;
  hi = hi < nq ? hi : nq;
  while (lo < hi) {  // (an 8-ary search with 7 probes in flight per round was SLOWER: 40 vs 25 us -- the kernel is bound by its number of random loads, not by their latency)
    const u32 mid = lo + ((hi - lo) >> 1);
    const u32 km = fq.qkey[mid] & key_mask;
    if (upper ? km <= tm : km < tm)
      lo = mid + 1;
    else
      hi = mid;
  }
  fq.qwin[2 * v + (upper ? 1 : 0)] = lo;
}

// One bucket per block, in place.  PAYLOAD bit 0: a rid array travels with the keys, bit 1: an
// end array.
//
// What bounds this kernel before HBM does is the LDS: an access with a per-lane random address
// costs ~10 cycles per wave instruction (bank conflicts), and a row has ~110 CU cycles per 64
// rows to stay under the 0.29 ms that reading and writing 1.6 GB takes.  A ballot-ranked radix
// round costs ~5 such accesses and ~60 VALU instructions per row and digit; binning + compare
// loops ~11.  This form needs FOUR random accesses per row and no ranking at all:
//
//   Every row is ONE word {low 16 key bits : 16, row index inside the bucket : 12} -- distinct
//   inside a bucket, so ascending word order IS the stable order.  The 16 key bits split into a
//   bin (top 11 bits) and a sub-value (low 5 bits).  Each row makes ONE 64-bit LDS atomic add of
//   {1 << sub : 32 | 1 : 32} on its bin's cell: the low half counts the bin's rows, the high
//   half is the sum of one power of two per row -- a bitmap of the sub-values present as long as
//   they are distinct, which popcount(high) == count tells exactly (equal powers carry, and a
//   carry only ever lowers the popcount; carries leave through bit 63, never into the count).
//   A block scan turns counts into bin starts, and a row of a bin with distinct sub-values has
//   its final place from ONE read of its cell: start + popcount(bitmap below its sub-value).
//   Rows of the other bins (equal 16-bit keys: ~3 % of random rows; pile-ups) gather in their
//   bin's own output range by arrival order and count the smaller words there (the square of the
//   bin's size, bounded by the bucket).  Low key halves (16-bit) and the payload then go through
//   LDS by final place in one round and leave in row order.
constexpr int BS_LOG_BINS = 11;
constexpr int BS_SUB_BITS = 16 - BS_LOG_BINS;  // 5: the sub-values of a bin fit a 32-bit map
#ifndef GIQL_BS_MIN_WAVES
#define GIQL_BS_MIN_WAVES 8  // waves per SIMD the register allocation must allow (8 = 64 VGPRs: four blocks per CU)
#endif

// timing-only builds (-DGIQL_BS_ABLATE=k): the kernel stops after its k-th stage (results invalid)
#if defined(GIQL_BS_ABLATE)
#define GIQL_BS_STOP(k)                                        \
  do {                                                         \
    if (GIQL_BS_ABLATE == (k)) {                               \
      u32 acc_ = 0;                                            \
      for (int i_ = 0; i_ < R; i_++) acc_ += pk[i_] ^ pay[i_] ^ slot[i_]; \
      if (acc_ == 0x12345u) kp[0] = acc_;                      \
      return;                                                  \
    }                                                          \
  } while (0)
#else
#define GIQL_BS_STOP(k) do { } while (0)
#endif

constexpr u32 BS_CELL_START_MASK = 0x1FFFu;  // cell low half after the scan: start : 13 | count : 13 | ... | dup : 1
constexpr u32 BS_CELL_DUP = 1u << 31;
constexpr u32 BS_NB = 1u << BS_LOG_BINS;

// The body for a bucket of (R - 1) * BS_NT < cnt <= R * BS_NT rows: row i * BS_NT + tid is item i
// of thread tid, so every item but the last is a full round (no bounds predicates).
// rank of a 16-bit value inside the bucket = rows of the bucket with a smaller low key half.  Valid
// between the barrier that ends the placing of the distinct-key rows (the bins with equal keys have
// gathered their words in s_buf by then) and the barrier before s_buf / the cells are overwritten.
__device__ __forceinline__ u32 bs_rank16(u32 x16, const u64* s_cell, const u32* s_buf) {
  const u64 cell = s_cell[x16 >> BS_SUB_BITS];
  const u32 lo = (u32)cell, start = lo & BS_CELL_START_MASK;
  if (!(lo & BS_CELL_DUP)) return start + (u32)__popc((u32)(cell >> 32) & ((1u << (x16 & 31u)) - 1u));
  const u32 m = (lo >> 13) & BS_CELL_START_MASK;
  u32 c = 0;
  for (u32 j = 0; j < m; j++) c += (u32)((s_buf[start + j] >> 12) < x16);
  return start + c;
}

// ... of a full key x that lies in the block's bucket (sh = 16 - W: the bucket's low key bits scaled up to 16)
__device__ __forceinline__ u32 bs_rank_key(u32 x, u32 sh, const u64* s_cell, const u32* s_buf) {
  return bs_rank16((x << sh) & 0xFFFFu, s_cell, s_buf);
}

// Barrier of the bucket sort's body.  __syncthreads() also drains the wave's outstanding GLOBAL loads and stores
// (s_waitcnt vmcnt(0)); the fused form has the query bounds' stores in flight in the middle of the kernel, and
// every block would stand still for their acknowledgement: its barriers wait for the wave's LDS operations only.
template <int LDS_ONLY>
__device__ __forceinline__ void bs_sync() {
  if (LDS_ONLY)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else
    __syncthreads();
}
#ifndef GIQL_BS_QPRE
#define GIQL_BS_QPRE 2
#endif
constexpr int BS_QPRE = GIQL_BS_QPRE;  // probe rounds whose values are loaded with the rows

// ---- the join itself in the bucket stage (FUSE == 2, round 3) ----
// With the bounds answered from LDS (FUSE == 1) the sorted row ids still made a round trip through HBM: written by
// the bucket sort (4 B per row), gathered again by the fill, with the per-query bounds, their scan and the fill's
// partition in between -- 1.2 GB of the 11.3 GB a headline join moved.  But the block that sorts bucket v holds
// everything the pairs of that bucket are made of: the bucket's row ids in sorted order (staged in LDS by final
// place) and the window of queries whose range reaches into the bucket.  So in the one-call form (the caller's
// output buffers are known, giql_hip_inner_join_dev) it writes the pairs itself: every query of the window gets
// its range CLAMPED to the bucket (a range that spans two buckets is emitted half by each), the block adds up its
// pairs, takes its place in the output with one atomic add on a global cursor, and its waves write the runs --
// one query per wave iteration, the query's numbers broadcast from the lane that ranked it (v_readlane).  Neither
// the sorted ids nor any per-query array reaches HBM, and the count / scan / partition / fill launches are gone.
// The order of the pairs depends on the order the blocks arrive at the cursor: unspecified, as the order of an
// INNER join's rows always was (bag semantics, intersects_duckdb.py:1283-1330 emits no ORDER BY).
// A window holds at most BJ_QR queries per thread (their ranks live in registers while the bin table is reused);
// denser windows, like buckets too large for LDS, go to the queue of k_bucket_sort_big.
#ifndef GIQL_BJ_MIN_WAVES
#define GIQL_BJ_MIN_WAVES 8  // the join form keeps its queries' ranks in registers next to the rows: 64 VGPRs at two rounds, four blocks per CU
#endif
#ifndef GIQL_BJG_MIN_WAVES
#define GIQL_BJG_MIN_WAVES 6  // ... the general form its rows' class-1 ranges as well (82 VGPRs)
#endif
// Queries a thread ranks (window cap = BJ_QR x 512).  Every round costs its instructions whether the window fills it or
// not, and its ranks live in registers: at the headline sizes (636 queries per window on average) 4 rounds / 78 VGPRs
// ran the kernel in 1.02 ms, 3 rounds / 69 VGPRs in 0.93-0.98, 2 rounds / 64 VGPRs (four blocks per CU) in 0.91-0.93.
// Two it is: a window over the 1024-query cap is not lost to the slow path but runs the same body with BJ_QR_CROWD
// rounds in the queue kernel (below).
#ifndef GIQL_BJ_QR
#define GIQL_BJ_QR 2
#endif
constexpr int BJ_QR = GIQL_BJ_QR;
constexpr u32 BJ_WCAP = BJ_QR * BS_NT;
// A window over that cap but within BJ_QR_CROWD rounds (real query tables are clustered: a promoter-dense stretch
// holds many times the average) still runs in LDS -- the same body with more rounds per thread, one block per such
// bucket, in the queue kernel (k_bucket_sort_big), where occupancy does not matter; only beyond that does a bucket
// take the global-memory path.
constexpr int BJ_QR_CROWD = 8;
constexpr u32 BJ_WCAP_CROWD = BJ_QR_CROWD * BS_NT;

// The tail of a FUSE == 2 block (see above).  On entry every row knows its final place (slot), the bin table and
// the gathered equal-key bins are still valid, and no barrier has passed since the last of them was read.
//
// GENERAL (FUSE == 3, the two-class join of rows of ANY length; the bucket rows carry their end keys): the queries'
// own ranges over the bucket's rows are class 2 exactly as above (range (q.start, q.end): lo_off = +1), and class 1
// -- the queries that START inside a bucket row, q.start in [u.start, u.end) -- is answered from the other side: the
// window's keys are staged in LDS (in the bin table's place, once the ranks are done), every bucket row searches
// them for its own range (a lower bound and a short walk: 0.3 matches per row at 10M x 100M) and writes its few
// pairs itself, behind the class-2 runs.  Every pair (q, u) leaves from the block of u's bucket.
template <int R, bool GENERAL, int QR, bool OUTMAJOR = false, bool W16 = false>
__device__ __forceinline__ void bucket_join_tail(const u32 (&pk)[R], const u32 (&pay)[R], const u32 (&slot)[R], u32 cnt,
                                                 u32 v, const u32* __restrict__ ep, u32* s_buf, u64* s_cell,
                                                 u32* s_jtot, const BsFuse& fq, u32 qw0, u32 nw,
                                                 const u32 (&jq_key)[QR], const u32 (&jq_end)[QR],
                                                 u32* s_bend = nullptr, u32* s_run = nullptr) {
  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const u32 wb = W16 ? 16u : fq.wbits, sh = 16u - wb;  // (W16: the 65,536-key bucket's shifts and masks fold to constants)
  const u32 k0 = v << wb;
#if defined(GIQL_BJ_ABLATE)  // timing-only builds (results invalid, tools/bj_ablate.sh): the tail stops after its k-th stage
#define GIQL_BJ_STOP(k) do { if (GIQL_BJ_ABLATE == (k)) { if (pay[0] == 0x12345u && slot[0] == 77u) s_buf[0] = 1; return; } } while (0)
#else
#define GIQL_BJ_STOP(k) do { } while (0)
#endif
  GIQL_BJ_STOP(1);  // the sort alone
  if (GENERAL) {
    // class 1 is answered from the queries' side too (below): the rows' end keys by final place, and the bucket's
    // longest row (how far below a query's start a row that still covers it can begin)
    u32 lmax = 0, pe[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
      const u32 r = i * BS_NT + tid;
      pe[i] = (i < R - 1 || r < cnt) ? s_bend[r] : 0u;  // parked by input place (bucket_sort_body)
    }
    bs_sync<2>();
#pragma unroll
    for (int i = 0; i < R; i++) {
      const u32 r = i * BS_NT + tid;
      if (i < R - 1 || r < cnt) {
        const u32 key = k0 | (pk[i] >> (12u + sh));
        s_bend[slot[i]] = pe[i];
        const u32 len = pe[i] > key ? pe[i] - key : 0u;
        lmax = len > lmax ? len : lmax;
      }
    }
    lmax = wave_reduce_max_u32(lmax);
    if (lane == 0) s_jtot[QR * BS_NW + 4 + w] = lmax;  // (read behind the barrier that ends the ranking)
  }
  // ranks of my queries' bounds inside this bucket, clamped to it
  u32 q_lo[QR], q_cnt[QR], q_rid[QR], incl[QR];
#pragma unroll
  for (int i = 0; i < QR; i++) {
    const u32 j = i * BS_NT + tid;
    q_lo[i] = q_cnt[i] = q_rid[i] = 0;
    if (j < nw) {
      const u32 xlo = bs_shift_key(jq_key[i], fq.lo_off), xhi = jq_end[i];
      const u32 lo_l = xlo < k0 ? 0u : ((xlo >> wb) != v ? cnt : bs_rank_key(xlo, sh, s_cell, s_buf));
      const u32 hi_l = xhi <= k0 ? 0u : ((xhi >> wb) != v ? cnt : bs_rank_key(xhi, sh, s_cell, s_buf));
      q_lo[i] = lo_l;
      q_cnt[i] = hi_l > lo_l ? hi_l - lo_l : 0u;
      if (q_cnt[i]) q_rid[i] = fq.qrid[qw0 + j];  // (flies under the scan and the staging below)
    }
    incl[i] = wave_incl_scan_add_u32(q_cnt[i]);
    // a wave's pairs (< 2^18: 64 queries x at most 4096 rows each) and, above them, its RUNS (queries with a pair)
    const u32 runs_w = OUTMAJOR ? (u32)__popcll(__ballot(q_cnt[i] != 0u)) : 0u;
    if (lane == WAVE - 1) s_jtot[i * BS_NW + w] = incl[i] | (runs_w << 24);
  }
  bs_sync<2>();  // every rank has been read: s_buf and the cells are free; the wave totals are in
  GIQL_BJ_STOP(2);  // + ranks
  // pairs of the block, and where each (slot, wave) group of queries starts among them; the same for the runs
  u32 total = 0, q_off[QR], n_runs = 0, run0[QR];
#pragma unroll
  for (int i = 0; i < QR; i++) {
    u32 mine = 0, mine_r = 0;
#pragma unroll
    for (int k = 0; k < BS_NW; k++) {
      if (k == (int)w) {
        mine = total;
        mine_r = n_runs;
      }
      const u32 t = s_jtot[i * BS_NW + k];
      total += t & 0xFFFFFFu;
      n_runs += t >> 24;
    }
    q_off[i] = mine + incl[i] - q_cnt[i];
    run0[i] = mine_r;  // the first run of my wave in this round (my rank among them is added where the table is written)
  }
  const u32 total2 = total;  // the class-2 pairs: the outputs [0, total2) of the block
  // class 1 (GENERAL): the bucket rows that COVER a query's start, q.start in [u.start, u.end).  Asked from the
  // query's side: such a row starts in (q.start - longest row of the bucket, q.start] -- a contiguous stretch of the
  // sorted bucket, both ends one bin-table read -- and is kept when its end key (staged above by final place) lies
  // beyond q.start.  A handful of candidates per query at read lengths; asked from the rows' side (every row
  // binary-searching the window's keys) it was a third of the kernel: 0.50 of 1.39 ms.
  u32 c_lo[GENERAL ? QR : 1], c_hi[GENERAL ? QR : 1], c_cnt[GENERAL ? QR : 1], c_off[GENERAL ? QR : 1];
  if (GENERAL) {
    u32 lmax = 0;
#pragma unroll
    for (int k = 0; k < BS_NW; k++) lmax = s_jtot[QR * BS_NW + 4 + k] > lmax ? s_jtot[QR * BS_NW + 4 + k] : lmax;
    u32 incl1[QR];
#pragma unroll
    for (int i = 0; i < QR; i++) {
      const u32 j = i * BS_NT + tid;
      c_lo[i] = c_hi[i] = c_cnt[i] = 0;
      if (j < nw && jq_key[i] >= k0) {
        const u32 x = jq_key[i];
        // rows with key <= x, and rows with key <= x - lmax (those end at or before x)
        const u32 hi_p = ((x >> wb) != v || ((x + 1u) >> wb) != v) ? cnt : bs_rank_key(x + 1u, sh, s_cell, s_buf);
        u32 lo_p = 0;
        if (x - k0 >= lmax) {  // y = x - lmax >= K0
          const u32 y = x - lmax;
          lo_p = ((y >> wb) != v || ((y + 1u) >> wb) != v) ? cnt : bs_rank_key(y + 1u, sh, s_cell, s_buf);
        }
        u32 c = 0;
        for (u32 p = lo_p; p < hi_p; p++) c += (u32)(s_bend[p] > x);
        c_lo[i] = lo_p;
        c_hi[i] = hi_p;
        c_cnt[i] = c;
        if (c && !q_cnt[i]) q_rid[i] = fq.qrid[qw0 + j];
      }
      incl1[i] = wave_incl_scan_add_u32(c_cnt[i]);
    }
    bs_sync<2>();  // every lmax has been read: the slots are the class-1 wave totals' now
#pragma unroll
    for (int i = 0; i < QR; i++)
      if (lane == WAVE - 1) s_jtot[i * BS_NW + w] = incl1[i];  // (the class-2 totals were consumed above)
    bs_sync<2>();
#pragma unroll
    for (int i = 0; i < QR; i++) {
      u32 mine = 0;
#pragma unroll
      for (int k = 0; k < BS_NW; k++) {
        if (k == (int)w) mine = total;
        total += s_jtot[i * BS_NW + k];
      }
      c_off[i] = mine + incl1[i] - c_cnt[i];
    }
  }
  if (total == 0) return;  // block-uniform: no query of the window meets a row of this bucket
  // the block's place in the output: one atomic, in flight while the row ids are staged by final place
  unsigned long long base = 0;
  if (tid == 0) base = atomicAdd(fq.cursor, (unsigned long long)total);
#pragma unroll
  for (int i = 0; i < R; i++) {
    const u32 r = i * BS_NT + tid;
    if (i < R - 1 || r < cnt) s_buf[slot[i]] = pay[i];
  }
  // the run table of the output-major emission (below): {first output, first sorted row, query id} per run, in the
  // order of the outputs, in the bin table's memory (every rank has been read)
  constexpr u32 RC = QR * BS_NT;
  if constexpr (OUTMAJOR) {
#pragma unroll
    for (int i = 0; i < QR; i++) {
      const u64 mr = __ballot(q_cnt[i] != 0u);
      if (q_cnt[i] != 0u) {
        const u32 r = run0[i] + __builtin_amdgcn_mbcnt_hi((u32)(mr >> 32), __builtin_amdgcn_mbcnt_lo((u32)mr, 0u));
        s_run[r] = q_off[i];
        s_run[RC + 1 + r] = q_lo[i];
        s_run[2 * RC + 1 + r] = q_rid[i];
      }
    }
    if (tid == 0) s_run[n_runs] = total2;
  }
  unsigned long long* s_jbase = reinterpret_cast<unsigned long long*>(s_jtot + QR * BS_NW);
  if (tid == 0) *s_jbase = base;
  bs_sync<2>();
  base = *s_jbase;
  GIQL_BJ_STOP(3);  // + place in the output, ids staged
  if (base + total > fq.cap) return;  // the caller's buffers are too small: the count still adds up, nothing is written
  int32_t* const rq = fq.row_q + base;
  int32_t* const rs = fq.row_s + base;
  if constexpr (OUTMAJOR) {
   if (total2 != 0u) {
    // OUTPUT-MAJOR emission (round 4).  Every wave takes a contiguous stretch of the block's class-2 output in windows
    // of 256 pairs ALIGNED to 1 KB of the output arrays; a lane owns FOUR consecutive pairs of a window and writes them
    // with one 16-byte store per array (the run-major loop below writes one run per iteration -- ~35 of 64 lanes at the
    // headline's run lengths, two partial lines per store, four v_readlane per run; the first output-major version
    // wrote one pair per lane and window: the kernel is bound by its instruction count, and the per-window work --
    // which run does my position belong to -- is shared by four pairs now).  The run of a lane's first pair comes from
    // a binary search of the (sorted) run table between the run the window opens in and the last run that starts
    // inside it (one coalesced LDS read tells how many do: 2-8 at the headline's sizes); its other three pairs walk on.
    const bool vec_ok = ((((uintptr_t)fq.row_q) | ((uintptr_t)fq.row_s)) & 15u) == 0;   // block-uniform
    const u32 shift = (u32)(base & 255ull);
    const u32 n_win = (shift + total2 + 255u) >> 8;
    const u32 per = (n_win + BS_NW - 1) / BS_NW;
    u32 win = w * per;
    const u32 win_end = win + per < n_win ? win + per : n_win;
    if (win < win_end) {
      int o_base = (int)(win * 256u) - (int)shift;   // block-local output of lane 0's first pair (negative only in the first window)
      const u32 o_first = o_base < 0 ? 0u : (u32)o_base;
      u32 j0 = 0, hi_ = n_runs;   // s_run[j0] <= o_first < s_run[hi_]  (s_run[0] = 0, s_run[n_runs] = total2)
      while (hi_ - j0 > 1u) {
        const u32 mid = (j0 + hi_) >> 1;
        if (s_run[mid] <= o_first)
          j0 = mid;
        else
          hi_ = mid;
      }
      typedef int bj_i4 __attribute__((ext_vector_type(4)));
      for (; win < win_end; win++, o_base += 256) {
        // how many runs start inside the window (64 or more: the search below takes the whole table)
        const u32 cand = j0 + 1u + lane;
        const u32 s_c = cand < n_runs ? s_run[cand] : 0x7FFFFFFFu;
        const u32 nin = (u32)__popcll(__ballot((int)s_c < o_base + 256));   // a prefix of the lanes: the table is sorted
        const int o0 = o_base + 4 * (int)lane;
        const u32 oc = o0 < 0 ? 0u : (u32)o0;
        u32 j = j0, hj = nin == 64u ? n_runs : j0 + nin + 1u;   // s_run[j] <= oc < s_run[hj] (or oc past the block's output)
        if (hj > n_runs) hj = n_runs;
        while (hj - j > 1u) {
          const u32 mid = (j + hj) >> 1;
          if (s_run[mid] <= oc)
            j = mid;
          else
            hj = mid;
        }
        u32 off = s_run[j], nxt = s_run[j + 1u], lo = s_run[RC + 1 + j];
        int rid = (int)s_run[2 * RC + 1 + j];
        bj_i4 q4, s4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int o = o0 + k;
          int sv = 0;
          if (o >= 0 && (u32)o < total2) {
            while ((u32)o >= nxt) {   // the next run (short runs: more than one step)
              j++;
              off = nxt;
              nxt = s_run[j + 1u];
              lo = s_run[RC + 1 + j];
              rid = (int)s_run[2 * RC + 1 + j];
            }
            sv = (int)s_buf[lo + ((u32)o - off)];
          }
          q4[k] = rid;
          s4[k] = sv;
        }
        if (vec_ok && o0 >= 0 && (u32)o0 + 3u < total2) {
          *reinterpret_cast<bj_i4*>(rq + o0) = q4;
          *reinterpret_cast<bj_i4*>(rs + o0) = s4;
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int o = o0 + k;
            if (o >= 0 && (u32)o < total2) {
              rq[o] = q4[k];
              rs[o] = s4[k];
            }
          }
        }
        // the run the next window opens in: the last valid lane's last run
        j0 = (u32)__builtin_amdgcn_readlane((int)j, 63);
      }
    }
   }
  }
  // one query per wave iteration: its numbers come from the lane that ranked it (the general form, whose registers
  // are taken by its class-1 ranges, and the crowded windows of the queue kernel, whose run table would not fit
  // the bin table's memory)
  if constexpr (!OUTMAJOR)
#pragma unroll
  for (int i = 0; i < QR; i++) {
    u64 m = __ballot(q_cnt[i] != 0u);
    while (m) {
      const int l = __ffsll((long long)m) - 1;
      m &= m - 1;
      const u32 c = (u32)__builtin_amdgcn_readlane((int)q_cnt[i], l);
      const u32 lo = (u32)__builtin_amdgcn_readlane((int)q_lo[i], l);
      const u32 off = (u32)__builtin_amdgcn_readlane((int)q_off[i], l);
      const int qr = __builtin_amdgcn_readlane((int)q_rid[i], l);
      for (u32 k = lane; k < c; k += WAVE) {
        rq[off + k] = qr;
        rs[off + k] = (int32_t)s_buf[lo + k];
      }
    }
  }
  if (GENERAL) {  // class 1: a query's few pairs from its own thread, behind the class-2 runs
#pragma unroll
    for (int i = 0; i < QR; i++) {
      if (c_cnt[i]) {
        const u32 x = jq_key[i];
        u32 o = c_off[i];
        for (u32 p = c_lo[i]; p < c_hi[i]; p++) {
          if (s_bend[p] > x) {
            rq[o] = (int32_t)q_rid[i];
            rs[o] = (int32_t)s_buf[p];
            o++;
          }
        }
      }
    }
  }
}


template <int PAYLOAD, int R, int FUSE = 0, int QR = BJ_QR, bool W16 = false>
__device__ __forceinline__ void bucket_sort_body(u32* __restrict__ kp, u32* __restrict__ pp,
                                                 u32* __restrict__ ep, u32 cnt, u32 v, u32* s_buf,
                                                 u64* s_cell, u32* s_scan, const BsFuse& fq,
                                                 u32 b0, u32 qw0, u32 qw1, u32* s_jtot = nullptr,
                                                 u32* s_bend = nullptr) {
  constexpr int BIN_SHIFT = 12 + BS_SUB_BITS;
  constexpr int PER = BS_NB / BS_NT;  // cells scanned per thread
  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const u32 wb = W16 ? 16u : fq.wbits, sh = 16u - wb;  // block-uniform (SGPRs); constants in the W16 instances
#define GIQL_BS_OK(i, r) ((i) < R - 1 || (r) < cnt)
  u32 pk[R], pay[R], slot[R];
#pragma unroll
  for (int i = 0; i < R; i++) {
    const u32 r = i * BS_NT + tid;
    const bool ok = GIQL_BS_OK(i, r);
    pk[i] = ok ? kp[r] : 0u;
    pay[i] = (PAYLOAD && ok) ? pp[r] : 0u;  // used at the very end: the load flies under everything
    slot[i] = 0;
  }
  // general join form: the end keys ride in with the rows and wait in LDS by INPUT place (bucket_join_tail moves
  // them to their final places); loaded in the tail they were a second exposed round trip per block
  u32 pe0[FUSE == 3 ? R : 1];
  if constexpr (FUSE == 3) {
#pragma unroll
    for (int i = 0; i < R; i++) {
      const u32 r = i * BS_NT + tid;
      pe0[i] = GIQL_BS_OK(i, r) ? ep[r] : 0u;
    }
  }
  // fused count: this thread's first query of the window, loaded in the same round trip as the rows (every
  // __syncthreads drains the wave's outstanding loads: issued any later, the round trip stands exposed)
  // Two threads per query row: the even one answers its lower bound (from the key), the odd one its upper
  // bound (from the end key) -- one load, one cell read and one store each, spread over all eight waves.
  const u32* q_src = (tid & 1u) ? fq.qend : fq.qkey;
  u32* q_dst = (tid & 1u) ? fq.hi_out : fq.lo_out;
  const u32 n_probe = FUSE == 1 ? 2u * (qw1 - qw0) : 0u;
  // (the first BS_QPRE rounds' values are loaded up front, with the rows: a window of one bucket's queries is
  // one round, the three buckets a query side grouped by bucket only brings along are three)
  u32 q_val[BS_QPRE];
#pragma unroll
  for (int k = 0; k < BS_QPRE; k++) {
    q_val[k] = 0;
    if (FUSE == 1 && tid + k * BS_NT < n_probe) q_val[k] = q_src[qw0 + ((tid + k * BS_NT) >> 1)];
  }
  // the join form: one thread per query of the window, key and end key, up to QR rounds
  u32 jq_key[FUSE >= 2 ? QR : 1], jq_end[FUSE >= 2 ? QR : 1];
  if constexpr (FUSE >= 2) {
#pragma unroll
    for (int k = 0; k < QR; k++) {
      const u32 j = k * BS_NT + tid;
      const bool in = j < qw1 - qw0;
      jq_key[k] = in ? fq.qkey[qw0 + j] : 0u;
      jq_end[k] = in ? fq.qend[qw0 + j] : 0u;
    }
  }
  // cells zeroed while the loads fly
#pragma unroll
  for (int k = 0; k < PER; k++) s_cell[tid + k * BS_NT] = 0;
  bs_sync<FUSE>();
  GIQL_BS_STOP(1);  // loads + table zeroing
#pragma unroll
  for (int i = 0; i < R; i++) {
    const u32 r = i * BS_NT + tid;
    if (GIQL_BS_OK(i, r)) {
      pk[i] = ((pk[i] << (12u + sh)) & 0x0FFFF000u) | r;
      const u32 sub = (pk[i] >> 12) & 31u;
      const u64 add = ((u64)(1u << sub) << 32) | 1ull;
      slot[i] = (u32)atomicAdd((unsigned long long*)&s_cell[pk[i] >> BIN_SHIFT], (unsigned long long)add);
    }
  }
  if constexpr (FUSE == 3) {
#pragma unroll
    for (int i = 0; i < R; i++) {
      const u32 r = i * BS_NT + tid;
      if (GIQL_BS_OK(i, r)) s_bend[r] = pe0[i];
    }
  }
  bs_sync<FUSE>();
  GIQL_BS_STOP(2);  // + binning atomics
  {
    // exclusive scan of the bin counts: PER consecutive cells per thread, DPP scan per wave, the
    // wave totals through LDS (every thread adds up the waves below its own)
    u32 c[PER], bm[PER], t = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const u64 cell = s_cell[tid * PER + k];
      c[k] = (u32)cell;
      bm[k] = (u32)(cell >> 32);
      t += c[k];
    }
    const u32 incl = wave_incl_scan_add_u32(t);
    if (lane == WAVE - 1) s_scan[w] = incl;
    bs_sync<FUSE>();
    u32 ex = incl - t;
#pragma unroll
    for (int k = 0; k < BS_NW; k++)
      if (k < (int)w) ex += s_scan[k];
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const u32 dup = (u32)__popc(bm[k]) != c[k] ? BS_CELL_DUP : 0u;
      s_cell[tid * PER + k] = ((u64)bm[k] << 32) | (u64)(ex | (c[k] << 13) | dup);
      ex += c[k];
    }
  }
  bs_sync<FUSE>();
  GIQL_BS_STOP(3);  // + scan
  bool any_dup = false;
#pragma unroll
  for (int i = 0; i < R; i++) {
    const u32 r = i * BS_NT + tid;
    if (GIQL_BS_OK(i, r)) {
      const u64 cell = s_cell[pk[i] >> BIN_SHIFT];
      const u32 lo = (u32)cell, map = (u32)(cell >> 32);
      const u32 start = lo & BS_CELL_START_MASK;
      if (lo & BS_CELL_DUP) {
        s_buf[start + slot[i]] = pk[i];  // gathered inside the bin's own output range
        slot[i] = U32_MAX;               // place still to be found
        any_dup = true;
      } else {
        const u32 sub = (pk[i] >> 12) & 31u;
        slot[i] = start + (u32)__popc(map & ((1u << sub) - 1u));
      }
    }
  }
  bs_sync<FUSE>();
  GIQL_BS_STOP(4);  // + places of the rows with distinct keys
  if (FUSE == 1) {
    // the bounds of the query window that fall into this bucket: one cell read each
#pragma unroll
    for (int k = 0; k < BS_QPRE; k++) {
      const u32 i = tid + k * BS_NT;
      if (i < n_probe) {
        const u32 x = (tid & 1u) ? q_val[k] : bs_shift_key(q_val[k], fq.lo_off);
        if ((x >> wb) == v) q_dst[qw0 + (i >> 1)] = b0 + bs_rank_key(x, sh, s_cell, s_buf);
      }
    }
    for (u32 i = tid + BS_QPRE * BS_NT; i < n_probe; i += BS_NT) {  // wider windows (dense query tables)
      const u32 q = qw0 + (i >> 1);
      const u32 qv = q_src[q];
      const u32 x = (tid & 1u) ? qv : bs_shift_key(qv, fq.lo_off);
      if ((x >> wb) == v) q_dst[q] = b0 + bs_rank_key(x, sh, s_cell, s_buf);
    }
  }
  if (any_dup) {
    // four words from the bin's start at once (independent loads: one LDS round trip per row
    // instead of one per bin-mate), the rare longer bin in a loop
#pragma unroll
    for (int i = 0; i < R; i++) {
      if (slot[i] == U32_MAX) {  // only rows can hold it
        const u32 lo = (u32)s_cell[pk[i] >> BIN_SHIFT];
        const u32 start = lo & BS_CELL_START_MASK, m = (lo >> 13) & BS_CELL_START_MASK;
        const u32 x = pk[i];
        const u32 w0 = s_buf[start], w1 = s_buf[start + 1], w2 = s_buf[start + 2], w3 = s_buf[start + 3];
        u32 c = (u32)(w0 < x) + (u32)(m > 1u && w1 < x) + (u32)(m > 2u && w2 < x) + (u32)(m > 3u && w3 < x);
        for (u32 j = 4; j < m; j++) c += (u32)(s_buf[start + j] < x);
        slot[i] = start + c;
      }
    }
  }
  if constexpr (FUSE >= 2) {  // the pairs leave from here: no sorted array is stored
    // (the run table of the output-major emission takes the bin table's place once the ranks are done -- when it fits)
    // ... and in the fixed-length form: the general form's class-1 ranges take the registers (80 VGPRs; with the
    // run table on top the compiler spilled 259 of them: 3.03 -> 3.67 ms)
#if defined(GIQL_BJ_RUN_MAJOR)  // A/B aid: round 3's run-major emission
    constexpr bool OUTMAJOR = false;
#else
    constexpr bool OUTMAJOR = FUSE == 2 && 3u * QR * BS_NT + 1u <= 2u * BS_NB;
#endif
    bucket_join_tail<R, FUSE == 3, QR, OUTMAJOR, W16>(pk, pay, slot, cnt, v, ep, s_buf, s_cell, s_jtot, fq, qw0, qw1 - qw0, jq_key,
                                                 jq_end, s_bend, OUTMAJOR ? reinterpret_cast<u32*>(s_cell) : nullptr);
    return;
  }
  bs_sync<FUSE>();  // every gathered bin has been read: s_buf and the cells are free
  GIQL_BS_STOP(5);  // + places of the rows with equal keys
  uint16_t* s_key16 = reinterpret_cast<uint16_t*>(s_cell);
#pragma unroll
  for (int i = 0; i < R; i++) {
    const u32 r = i * BS_NT + tid;
    if (GIQL_BS_OK(i, r)) {
      if (!FUSE) s_key16[slot[i]] = (uint16_t)(pk[i] >> 12);  // the fused form never stores the sorted keys
      if (PAYLOAD) s_buf[slot[i]] = pay[i];
    }
  }
  bs_sync<FUSE>();
  GIQL_BS_STOP(6);  // + staging by final place (everything but the stores)
  // keys: the high half is the bucket's number.  Every load of the block's rows completed before
  // the barriers above, so the in-place stores cannot overtake a load.
#pragma unroll
  for (int i = 0; i < R; i++) {
    const u32 r = i * BS_NT + tid;
    if (GIQL_BS_OK(i, r)) {
      if (!FUSE) kp[r] = (v << wb) | ((u32)s_key16[r] >> sh);
      if (PAYLOAD) pp[r] = s_buf[r];
    }
  }
  if (PAYLOAD == 3) {
#pragma unroll
    for (int i = 0; i < R; i++) {
      const u32 r = i * BS_NT + tid;
      pay[i] = GIQL_BS_OK(i, r) ? ep[r] : 0u;
    }
    bs_sync<FUSE>();  // the rid round has left s_buf
#pragma unroll
    for (int i = 0; i < R; i++) {
      const u32 r = i * BS_NT + tid;
      if (GIQL_BS_OK(i, r)) s_buf[slot[i]] = pay[i];
    }
    bs_sync<FUSE>();  // every `end` of the bucket is in LDS: the in-place stores may start
#pragma unroll
    for (int i = 0; i < R; i++) {
      const u32 r = i * BS_NT + tid;
      if (GIQL_BS_OK(i, r)) ep[r] = s_buf[r];
    }
  }
#undef GIQL_BS_OK
}


// A bucket of BS_CAP < cnt <= BS_BIG_MAX rows: stable LSD radix sort on the low 16 key bits by ONE
// block, two 8-bit passes, buffer 0 -> buffer 1 -> buffer 0 (the bucket's own row range in both:
// no other block touches it).  Per pass: a digit histogram of the whole bucket (LDS atomics), its
// scan, then the tiles of BS_CAP rows in order -- each ranked stably inside the tile (wave ballots
// + per-wave digit counters, as the global passes do) on top of the running per-digit base.
template <int PAYLOAD>
__device__ __forceinline__ void bucket_sort_big(u32* __restrict__ k0, u32* __restrict__ e0, u32* __restrict__ r0,
                                             u32* __restrict__ k1, u32* __restrict__ e1, u32* __restrict__ r1,
                                             u32 cnt, u32* s_wcnt /* [BS_NW][256] */, u32* s_base /* [256] */,
                                             u32* s_scan /* [BS_NW] */) {
  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
  for (int p = 0; p < 2; p++) {
    const int shift = 8 * p;
    u32* ks = p ? k1 : k0;
    u32* kd = p ? k0 : k1;
    u32* es = p ? e1 : e0;
    u32* ed = p ? e0 : e1;
    u32* rs = p ? r1 : r0;
    u32* rd = p ? r0 : r1;
    __syncthreads();
    if (tid < OS_BINS) s_base[tid] = 0;
    __syncthreads();
    for (u32 i = tid; i < cnt; i += BS_NT) atomicAdd(&s_base[(ks[i] >> shift) & 0xFFu], 1u);
    __syncthreads();
    if (tid < OS_BINS) {  // exclusive scan of the 256 counts (4 waves)
      const u32 c = s_base[tid];
      const u32 incl = wave_incl_scan_add_u32(c);
      if (lane == WAVE - 1) s_scan[w] = incl;
      s_base[tid] = incl - c;
    }
    __syncthreads();
    if (tid < OS_BINS) {
      u32 wb = 0;
      for (int k = 0; k < OS_BINS / WAVE; k++)
        if (k < (int)w) wb += s_scan[k];
      s_base[tid] += wb;
    }
    for (u32 t0 = 0; t0 < cnt; t0 += BS_CAP) {
      __syncthreads();  // the previous tile's bases are final; its counters are free
      for (int k = tid; k < BS_NW * OS_BINS; k += BS_NT) s_wcnt[k] = 0;
      __syncthreads();
      // wave-striped: item i of lane l of wave w is row t0 + w * (BS_ITEMS * 64) + i * 64 + l
      u32 key[BS_ITEMS], rank[BS_ITEMS];
      u32* wc = s_wcnt + w * OS_BINS;
#pragma unroll
      for (int i = 0; i < BS_ITEMS; i++) {
        const u32 r = t0 + w * (BS_ITEMS * WAVE) + i * WAVE + lane;
        const bool ok = r < cnt;
        key[i] = ok ? ks[r] : 0u;
        const u32 d = (key[i] >> shift) & 0xFFu;
        const u64 active = __ballot(ok);
        u32 below, total;
        wave_match8(d, active, below, total);
        rank[i] = 0;
        if (ok) {
          const u32 pre = wc[d];
          rank[i] = pre + below;
          if (below == 0) wc[d] = pre + total;  // the peers have read `pre` (one wave, in-order LDS)
        }
      }
      __syncthreads();
      u32 tile_count = 0;
      if (tid < OS_BINS) {  // per-wave counts -> exclusive bases across the waves of this tile
        u32 run = 0;
#pragma unroll
        for (int k = 0; k < BS_NW; k++) {
          const u32 c = s_wcnt[k * OS_BINS + tid];
          s_wcnt[k * OS_BINS + tid] = run;
          run += c;
        }
        tile_count = run;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < BS_ITEMS; i++) {
        const u32 r = t0 + w * (BS_ITEMS * WAVE) + i * WAVE + lane;
        if (r < cnt) {
          const u32 d = (key[i] >> shift) & 0xFFu;
          const u32 pos = s_base[d] + wc[d] + rank[i];
          kd[pos] = key[i];
          if (PAYLOAD & 1) rd[pos] = rs[r];
          if (PAYLOAD & 2) ed[pos] = es[r];
        }
      }
      __syncthreads();
      if (tid < OS_BINS) s_base[tid] += tile_count;
    }
  }
}

// W16: the instance for buckets of 65,536 keys (the headline's): W is a compile-time constant there -- carried at run
// time it cost the join form 1.3 % (0.786 -> 0.797 ms); the narrow widths share ONE generic instance.
template <int PAYLOAD, int FUSE = 0, bool W16 = false>
__global__ __launch_bounds__(BS_NT, FUSE == 3 ? GIQL_BJG_MIN_WAVES : (FUSE == 2 ? GIQL_BJ_MIN_WAVES : GIQL_BS_MIN_WAVES)) void k_bucket_sort(u32* __restrict__ keys, u32* __restrict__ ends,
                                                           u32* __restrict__ rids,
                                                           const u32* __restrict__ bnd,
                                                           DevMeta* __restrict__ meta,
                                                           u32* __restrict__ big_list, BsFuse fq = BsFuse()) {
  static_assert(BS_NB % BS_NT == 0, "bins must be a multiple of the block size");
  static_assert(BS_NB * sizeof(u64) >= BS_CAP * sizeof(uint16_t), "the cell table doubles as the 16-bit key stage");
  static_assert(BS_SUB_BITS == 5, "one 32-bit map of sub-values per bin");
  __shared__ u32 s_buf[BS_CAP + 4];  // + 4: the four-wide read of a gathered bin may run past the last row
  __shared__ u64 s_cell[BS_NB];  // {sub-value map : 32 | count : 32}, after the scan {map | dup, count, start}
  __shared__ u32 s_scan[BS_NW];
  // join forms: wave totals of the class-2 pair counts, the block's output base (64 bits), wave totals of class 1
  __shared__ __attribute__((aligned(8))) u32 s_jtot[FUSE >= 2 ? BJ_QR * BS_NW + 4 + BS_NW : 2];
  __shared__ u32 s_bend[FUSE == 3 ? BS_CAP : 1];  // general join form: the rows' end keys by final place
  const u32 v = blockIdx.x;
  const u32 b0 = bnd[v];
  const u32 cnt = bnd[v + 1] - b0;
  u32 qw0 = 0, qw1 = 0;
  if (FUSE) {
    qw0 = fq.qwin[2 * v];
    qw1 = fq.qwin[2 * v + 1];
  }
  if (FUSE >= 2) {
    if (cnt == 0u || qw0 >= qw1) return;  // block-uniform: no row or no query, no pair (nothing else leaves this kernel)
  } else if (cnt < 2u && qw0 >= qw1) {
    return;  // block-uniform: nothing to sort, no bound to answer
  }
  if (cnt > BS_CAP || (FUSE >= 2 && qw1 - qw0 > BJ_WCAP)) {
    // too large for LDS: queued for k_bucket_sort_big (a launch of its own keeps this kernel free of
    // the big path's registers and scratch frame); past BS_BIG_MAX the whole call is repeated.
    // Bit 31 of the entry: only the WINDOW is too large for this kernel's rounds -- the bucket still runs in LDS there
    if (threadIdx.x == 0) {
      if (cnt > BS_BIG_MAX || !big_list)
        meta->status = GIQL_STATUS_RESORT;
      else
        big_list[1 + atomicAdd(&big_list[0], 1u)] =
            v | ((FUSE >= 2 && cnt <= BS_CAP && qw1 - qw0 <= BJ_WCAP_CROWD) ? 0x80000000u : 0u);
    }
    return;
  }
  u32* kp = keys + b0;
  // the payload that rides along in registers: rid when there is one, else end
  u32* pp = (PAYLOAD & 1) ? rids + b0 : ((PAYLOAD & 2) ? ends + b0 : nullptr);
  u32* ep = (PAYLOAD == 3) ? ends + b0 : nullptr;  // a second payload array takes a round of its own
#define GIQL_BS_BODY(RR) bucket_sort_body<PAYLOAD, RR, FUSE, BJ_QR, W16>(kp, pp, ep, cnt, v, s_buf, s_cell, s_scan, fq, b0, qw0, qw1, s_jtot, s_bend)
  switch ((cnt + BS_NT - 1) / BS_NT) {  // rows per thread: 1..BS_ITEMS, block-uniform
    case 0: case 1: GIQL_BS_BODY(1); break;  // (0: an empty bucket with bounds to answer)
    case 2: GIQL_BS_BODY(2); break;
    case 3: GIQL_BS_BODY(3); break;
    case 4: GIQL_BS_BODY(4); break;
    case 5: GIQL_BS_BODY(5); break;
    case 6: GIQL_BS_BODY(6); break;
    case 7: GIQL_BS_BODY(7); break;
    default: GIQL_BS_BODY(8); break;
  }
#undef GIQL_BS_BODY
}

// The buckets k_bucket_sort queued (big_list[0] = how many): one block each, grid-stride.
template <int PAYLOAD, int FUSE = 0>
__global__ __launch_bounds__(BS_NT) void k_bucket_sort_big(u32* __restrict__ keys, u32* __restrict__ ends,
                                                            u32* __restrict__ rids, u32* __restrict__ keys1,
                                                            u32* __restrict__ ends1, u32* __restrict__ rids1,
                                                            const u32* __restrict__ bnd,
                                                            const u32* __restrict__ big_list, BsFuse fq = BsFuse()) {
  __shared__ u32 s_wcnt[BS_NW * OS_BINS];
  __shared__ u32 s_base[OS_BINS];
  __shared__ u32 s_scan[BS_NW];
  // join forms: the LDS body with BJ_QR_CROWD rounds for the buckets whose window alone was too large
  __shared__ u32 s_buf[FUSE >= 2 ? BS_CAP + 4 : 1];
  __shared__ u64 s_cell[FUSE >= 2 ? BS_NB : 1];
  __shared__ __attribute__((aligned(8))) u32 s_jtot[FUSE >= 2 ? BJ_QR_CROWD * BS_NW + 4 + BS_NW : 2];
  __shared__ u32 s_bend[FUSE == 3 ? BS_CAP : 1];
  const u32 n_big = big_list[0];
  for (u32 i = blockIdx.x; i < n_big; i += gridDim.x) {
    const u32 entry = big_list[1 + i];
    const u32 v = entry & 0x7FFFFFFFu;
    const u32 b0 = bnd[v];
    const u32 cnt = bnd[v + 1] - b0;
    if constexpr (FUSE >= 2) {
      if (entry >> 31) {  // block-uniform
        u32* kp = keys + b0;
        u32* pp = rids + b0;
        u32* ep = (PAYLOAD == 3) ? ends + b0 : nullptr;
        const u32 qw0 = fq.qwin[2 * v], qw1 = fq.qwin[2 * v + 1];
#define GIQL_BS_BODY(RR) \
  bucket_sort_body<PAYLOAD, RR, FUSE, BJ_QR_CROWD>(kp, pp, ep, cnt, v, s_buf, s_cell, s_scan, fq, b0, qw0, qw1, s_jtot, s_bend)
        switch ((cnt + BS_NT - 1) / BS_NT) {
          case 0: case 1: GIQL_BS_BODY(1); break;
          case 2: GIQL_BS_BODY(2); break;
          case 3: GIQL_BS_BODY(3); break;
          case 4: GIQL_BS_BODY(4); break;
          case 5: GIQL_BS_BODY(5); break;
          case 6: GIQL_BS_BODY(6); break;
          case 7: GIQL_BS_BODY(7); break;
          default: GIQL_BS_BODY(8); break;
        }
#undef GIQL_BS_BODY
        __syncthreads();  // the LDS arrays are reused by the block's next bucket
        continue;
      }
    }
    bucket_sort_big<PAYLOAD>(keys + b0, (PAYLOAD & 2) ? ends + b0 : nullptr, (PAYLOAD & 1) ? rids + b0 : nullptr,
                             keys1 + b0, (PAYLOAD & 2) ? ends1 + b0 : nullptr, (PAYLOAD & 1) ? rids1 + b0 : nullptr, cnt,
                             s_wcnt, s_base, s_scan);
    __syncthreads();
    if (FUSE >= 2) {
      // the join form of a queued bucket (too many rows for LDS, or too many queries in its window): the bucket is
      // sorted in global memory now; the window in chunks of one query per thread, each chunk's pairs placed by
      // one atomic and written as in bucket_join_tail
      __shared__ unsigned long long s_jbase;
      const u32* kb = keys + b0;
      const u32* rb = rids + b0;
      const u32 wb = fq.wbits;
      const u32 k0 = v << wb, qw0 = fq.qwin[2 * v], qw1 = fq.qwin[2 * v + 1];
      const u32 lane = lane_id(), w = wave_id();
      for (u32 c0 = qw0; c0 < qw1; c0 += BS_NT) {  // block-uniform
        const u32 q = c0 + threadIdx.x;
        u32 q_lo = 0, q_cnt = 0, q_rid = 0;
        if (q < qw1) {
          const u32 xlo = bs_shift_key(fq.qkey[q], fq.lo_off), xhi = fq.qend[q];
          const u32 lo_l = xlo < k0 ? 0u : ((xlo >> wb) != v ? cnt : lower_bound_u32(kb, 0, cnt, xlo));
          const u32 hi_l = xhi <= k0 ? 0u : ((xhi >> wb) != v ? cnt : lower_bound_u32(kb, 0, cnt, xhi));
          q_lo = lo_l;
          q_cnt = hi_l > lo_l ? hi_l - lo_l : 0u;
          if (q_cnt) q_rid = fq.qrid[q];
        }
        const u32 incl = wave_incl_scan_add_u32(q_cnt);
        if (lane == WAVE - 1) s_scan[w] = incl;
        __syncthreads();
        u32 total = 0, mine = 0;
#pragma unroll
        for (int k = 0; k < BS_NW; k++) {
          if (k == (int)w) mine = total;
          total += s_scan[k];
        }
        if (threadIdx.x == 0) s_jbase = total ? atomicAdd(fq.cursor, (unsigned long long)total) : 0ull;
        __syncthreads();
        const unsigned long long base = s_jbase;
        if (total != 0u && base + total <= fq.cap) {
          const u32 q_off = mine + incl - q_cnt;
          int32_t* const rq = fq.row_q + base;
          int32_t* const rs = fq.row_s + base;
          u64 m = __ballot(q_cnt != 0u);
          while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u32 c = (u32)__builtin_amdgcn_readlane((int)q_cnt, l);
            const u32 lo = (u32)__builtin_amdgcn_readlane((int)q_lo, l);
            const u32 off = (u32)__builtin_amdgcn_readlane((int)q_off, l);
            const int qr = __builtin_amdgcn_readlane((int)q_rid, l);
            for (u32 k = lane; k < c; k += WAVE) {
              rq[off + k] = qr;
              rs[off + k] = (int32_t)rb[lo + k];
            }
          }
        }
        __syncthreads();  // s_scan and s_jbase are reused by the next chunk
      }
      if (FUSE == 3) {
        // general form, class 1: the bucket's rows in chunks of one per thread, each against the window's keys
        // (global memory here), a chunk's pairs placed by one atomic, every row writing its own
        const u32* eb = ends + b0;
        const u32* qk = fq.qkey + qw0;
        const u32 nw = qw1 - qw0;
        for (u32 r0 = 0; r0 < cnt; r0 += BS_NT) {  // block-uniform
          const u32 r = r0 + threadIdx.x;
          u32 lo1 = 0, c = 0;
          if (r < cnt) {
            lo1 = lower_bound_u32(qk, 0, nw, kb[r]);
            const u32 end = eb[r];
            while (lo1 + c < nw && qk[lo1 + c] < end) c++;
          }
          const u32 incl = wave_incl_scan_add_u32(c);
          if (lane == WAVE - 1) s_scan[w] = incl;
          __syncthreads();
          u32 total = 0, mine = 0;
#pragma unroll
          for (int k = 0; k < BS_NW; k++) {
            if (k == (int)w) mine = total;
            total += s_scan[k];
          }
          if (threadIdx.x == 0) s_jbase = total ? atomicAdd(fq.cursor, (unsigned long long)total) : 0ull;
          __syncthreads();
          const unsigned long long base = s_jbase;
          if (total != 0u && base + total <= fq.cap) {
            unsigned long long o = base + mine + incl - c;
            const int32_t mr = r < cnt ? (int32_t)rb[r] : 0;
            for (u32 k = 0; k < c; k++, o++) {
              fq.row_q[o] = (int32_t)fq.qrid[qw0 + lo1 + k];
              fq.row_s[o] = mr;
            }
          }
          __syncthreads();
        }
      }
    } else if (FUSE) {
      // this bucket's keys are sorted in global memory now: the bounds of its query window by binary search
      // (the block's own stores are visible to it after the barrier)
      const u32* kb = keys + b0;
      for (u32 q = fq.qwin[2 * v] + threadIdx.x; q < fq.qwin[2 * v + 1]; q += BS_NT) {
        const u32 xs = bs_shift_key(fq.qkey[q], fq.lo_off), xe = fq.qend[q];
        if ((xs >> fq.wbits) == v) fq.lo_out[q] = b0 + lower_bound_u32(kb, 0, cnt, xs);
        if ((xe >> fq.wbits) == v) fq.hi_out[q] = b0 + lower_bound_u32(kb, 0, cnt, xe);
      }
    }
  }
}

}  // namespace giql
