// giql_amd/csrc/onesweep.hip.h -- single-pass-per-digit stable LSD radix sort
// ("onesweep": global digit histograms up front, per-tile offsets by decoupled
// look-back) of (key, end, rid) triples.
//
// Compared with the three-launch pass of radix_sort.hip.h this reads the keys
// once per pass instead of twice and removes the tile-histogram scan:
//   per pass:  read 12 B/row, write 12 B/row   (algorithmic: exactly that)
// Tiles are 8192 rows (1024 threads x 8) so a 256-bin scatter writes ~32-row
// (128 B) runs per array; the arrays are staged through ONE 32 KB LDS buffer in
// turn; two blocks (32 waves) are resident per CU.
//
// Order of a tile (what bounds it is the chain of look-backs, tools/os_timeline.py):
//   key loads -> stable ranking (wave ballots) -> per-digit count PUBLISHED ->
//   positions, keys (and a single payload) through LDS into registers in sorted
//   order -> look-back walk (its wait overlaps all of the above in the predecessors)
//   -> scatter stores.
//
// Inter-workgroup protocol (cdna_hip_programming.md G16, "granule" form): the
// only cross-block data is one 32-bit status word per (tile, digit) holding
// {flag:2, count:30}.  It is written by ONE relaxed agent-scope atomic store and
// polled with relaxed agent-scope atomic loads (sc1: L1 is bypassed), so no
// fence is needed and nothing depends on workgroup placement.  The look-back
// cannot deadlock whatever the dispatch order: a block that finds a predecessor
// silent for too long computes that tile itself (see k_onesweep).
#pragma once

#include "dev_common.hip.h"

namespace giql {

constexpr int OS_BINS = 256;
constexpr int OS_MIN_TILE = 4096;  // smallest tile of any instantiated variant
constexpr u32 OS_FLAG_AGG = 1u << 30;
constexpr u32 OS_FLAG_PREFIX = 2u << 30;
constexpr u32 OS_VALUE_MASK = (1u << 30) - 1u;
constexpr u32 OS_MAX_ROWS = (1u << 30) - 1u;
constexpr u32 OS_NO_TILE = 0xFFFFFFFFu;
#ifndef GIQL_OS_LB_WIDTH
#define GIQL_OS_LB_WIDTH 8
#endif
constexpr int OS_LB_WIDTH = GIQL_OS_LB_WIDTH;  // status words polled per look-back round
#ifndef GIQL_OS_PRESTAGE
#define GIQL_OS_PRESTAGE 1  // (key, one payload) sorts: both LDS rounds before the look-back walk
#endif

constexpr u32 OS_HELP_AFTER = 1u << 11;  // look-back polls (a few ms, several whole passes) before a block computes a silent predecessor itself

// Timeline build (-DGIQL_OS_TIMELINE, tools/os_timeline.py): thread 0 of every block stamps the
// 100 MHz wall clock at the phase boundaries of its tile into g_os_tl[tile][k]; a diagnostic
// aid only -- the product build compiles none of it.
#if defined(GIQL_OS_TIMELINE)
constexpr u32 OS_TL_TILES = 1u << 15;
__device__ unsigned long long g_os_tl[OS_TL_TILES * 16];
#define GIQL_TL(tile, k)                                                             \
  do {                                                                               \
    if (threadIdx.x == 0 && (tile) < OS_TL_TILES) g_os_tl[(size_t)(tile) * 16 + (k)] = wall_clock64(); \
  } while (0)
#else
#define GIQL_TL(tile, k) do { } while (0)
#endif

// Stable in-wave rank of one item by its 8-bit digit.  For every digit bit b the
// wave ballots the bit (m) and each lane ORs into `mis` the lanes whose bit differs
// from its own: mis |= m ^ (mybit ? ~0 : 0).  peers = active & ~mis are the lanes
// holding the same digit.  Shaped so hipcc emits, per bit, v_bfe_i32 + v_cmp +
// 2 x v_xor + (OR folded three-way): ~5 VALU instead of the ~11 it produces for the
// naive `peers &= bit ? m : ~m`.
__device__ __forceinline__ void wave_match8(u32 d, u64 active, u32& below, u32& total) {
  u32 mis_lo = 0, mis_hi = 0;
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const int s = __builtin_amdgcn_sbfe((int)d, b, 1);  // 0 or -1
    const u64 m = __ballot(s != 0);
    mis_lo |= (u32)m ^ (u32)s;
    mis_hi |= (u32)(m >> 32) ^ (u32)s;
  }
  const u32 p_lo = (u32)active & ~mis_lo;
  const u32 p_hi = (u32)(active >> 32) & ~mis_hi;
  // lanes below me in the peer set: mbcnt counts mask bits of lower lanes directly
  below = __builtin_amdgcn_mbcnt_hi(p_hi, __builtin_amdgcn_mbcnt_lo(p_lo, 0u));
  total = (u32)__popc(p_lo) + (u32)__popc(p_hi);
}

// Tile order.  A block maps its blockIdx to a tile through the XCD-aware permutation of
// dev_common.hip.h (inside groups of 8 G blocks, block 8j + x takes tile
// ((j / G) * 8 + x) * G + j % G; G = OS_GROUP).  Workgroups are dealt round-robin to the 8 XCDs, so
// blocks x, x + 8, x + 16, ... -- and with them G CONSECUTIVE tiles -- land on one XCD
// at about the same time: the adjacent ~128-byte runs those tiles write into every
// digit bin meet in THAT XCD's L2, which then writes whole lines back instead of two
// partial lines per run (measured on 100M (key, end) rows: 0.471 -> 0.425 ms per pass).
// Only speed depends on placement and dispatch order; progress does not (k_onesweep).
#ifndef GIQL_OS_XCD_GROUP
#define GIQL_OS_XCD_GROUP 16  // consecutive tiles that run side by side on one XCD (4 / 8 / 12 / 16 / 24 / 32:
                              // sort 1.97 / 1.81 / 1.84 / 1.72 / 1.98 / 1.80 ms on the headline workload)
#endif
constexpr u32 OS_GROUPS = XCD_GROUPS;
constexpr u32 OS_GROUP = GIQL_OS_XCD_GROUP;

__device__ __forceinline__ u32 os_tile_of_ticket(u32 t, u32 n_tiles) { return xcd_tile_of_block<OS_GROUP>(t, n_tiles); }

// scatter store.  Kept temporal on purpose: the L2 merges the partial lines of adjacent
// runs; a non-temporal hint took the random-key pass from 0.47 to 0.74 ms.
__device__ __forceinline__ void os_store(u32* p, u32 v) { *p = v; }

// PAYLOAD bit 0: carry rid, bit 1: carry end.  0 = keys only, 1 = (key, rid),
// 2 = (key, end), 3 = (key, end, rid).
// FULL = every row of the tile is valid (all tiles but the last): no per-item
// bounds predicates.
// KEYGEN (first pass of a side in the aligned form, PAYLOAD 1 or 3): the keys are not read but
// built from the raw columns -- keys_in = the start column, and the chrom column in ends_in
// (PAYLOAD 1) or rids_in (PAYLOAD 3, where ends_in is the raw end column and the end keys are built
// the same way); key = abase[chrom] + start + start_off (u32, wrapping: exactly what the span
// pass's histogram counted, k_chrom_minmax<HIST>) -- so that side has no linearize pass.  The keys
// do not depend on the row being regular (end > start): the caller takes this pass only on the guess
// that every row is, and checks the guess afterwards.
template <int PAYLOAD, int OS_NT, int OS_ITEMS, bool FULL, bool KEYGEN = false>
__device__ __forceinline__ u32 onesweep_tile(
    const u32* __restrict__ keys_in, const u32* __restrict__ ends_in, const u32* __restrict__ rids_in,
    u32* __restrict__ keys_out, u32* __restrict__ ends_out, u32* __restrict__ rids_out, u32 n_valid,
    u32 tile, u32 tile_base, int shift, const u32* __restrict__ gbase, u32* __restrict__ status,
    DevMeta* __restrict__ meta, u32* s_buf, u32 (*s_wcnt)[OS_BINS], u32* s_dstart, u32* s_goff,
    u32* s_scan, u32* s_help, u32 help_after, u32 n_total, const u32* s_abase = nullptr, u32 start_off = 0,
    u32 end_off = 0, const bool unstable = false) {
  constexpr int OS_NW = OS_NT / WAVE;
  const u32 tid = threadIdx.x;
  const u32 lane = lane_id();
  const u32 w = wave_id();

  // wave-striped: item i of lane l of wave w is row w*(ITEMS*64) + i*64 + l
  // Only the keys are live during the ranking; each payload array is loaded one
  // phase before its staging round (64-VGPR budget = two 1024-thread blocks / CU).
  u32 key[OS_ITEMS];
  const u32 wbase = w * (OS_ITEMS * WAVE);
  const u32* kin = keys_in + tile_base;  // block-uniform bases: 32-bit lane offsets
  const u32* ein = (PAYLOAD & 2) ? ends_in + tile_base : nullptr;
  // KEYGEN: the row ids are synthesised (a first pass); with the end payload the chrom column comes in
  // through rids_in, without it through ends_in
  const u32* rin = ((PAYLOAD & 1) && rids_in && !KEYGEN) ? rids_in + tile_base : nullptr;
  const int* cin = KEYGEN ? reinterpret_cast<const int*>((PAYLOAD & 2) ? rids_in : ends_in) + tile_base : nullptr;
  if (KEYGEN) {
    u32 cc[OS_ITEMS];
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 r = wbase + i * WAVE + lane;
      const bool ok = FULL || r < n_valid;
      cc[i] = ok ? (u32)cin[r] : 0u;
      key[i] = ok ? kin[r] : 0u;
    }
    __syncthreads();  // the block's LDS is ready (counters zeroed, s_abase filled): see k_onesweep
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 r = wbase + i * WAVE + lane;
      const bool ok = FULL || r < n_valid;
      key[i] = ok ? s_abase[cc[i] & 31u] + key[i] + start_off : U32_MAX;
    }
  } else {
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 r = wbase + i * WAVE + lane;
      const bool ok = FULL || r < n_valid;
      key[i] = ok ? kin[r] : U32_MAX;
    }
    __syncthreads();  // the block's LDS is ready (counters zeroed): the loads above fly meanwhile
  }

  GIQL_TL(tile, 1);  // key loads issued
#if defined(GIQL_OS_TIMELINE)
  __builtin_amdgcn_s_waitcnt(0);  // ... and arrived
  GIQL_TL(tile, 2);
#endif
  // stable rank inside the wave (peers = lanes with the same digit)
  u32 rank[OS_ITEMS];
  if (unstable) {
    // UNSTABLE ranking (round 4; a FIRST pass whose caller does not need the rows of equal keys in input order --
    // the INNER join's sides): one returning LDS atomic on the wave's own counter is a row's rank among the wave's
    // rows of its digit.  The ballot ranking below costs ~45 VALU operations per row, and these passes are bound by
    // their instruction count (a wave64 VALU op holds its SIMD for 4 cycles), not by HBM.
    u32* wcnt = s_wcnt[w];
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 r = wbase + i * WAVE + lane;
      const bool ok = FULL || r < n_valid;
      rank[i] = ok ? atomicAdd(&wcnt[(key[i] >> shift) & 0xFFu], 1u) : 0u;
    }
  } else {
    u32* wcnt = s_wcnt[w];
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 r = wbase + i * WAVE + lane;
      const bool ok = FULL || r < n_valid;
      const u32 d = (key[i] >> shift) & 0xFFu;
      const u64 active = FULL ? ~0ull : __ballot(ok);
      u32 below, total;
#if defined(GIQL_ABLATE) && GIQL_ABLATE == 2  // timing-only build: no ranking
      below = lane;
      total = 64;
      (void)active;
#else
      wave_match8(d, active, below, total);
#endif
      rank[i] = 0;
      if (ok) {
        const u32 pre = wcnt[d];
        rank[i] = pre + below;
        // every peer has read `pre` (one wave, in-order LDS) before the leader adds
        if (below == 0) wcnt[d] = pre + total;
      }
    }
  }
  GIQL_TL(tile, 3);  // ranked
  __syncthreads();
  GIQL_TL(tile, 4);  // every wave ranked

  // threads 0..255 own one digit each: wave bases, tile digit starts, and the tile's count
  // published at once.  The look-back WALK comes later, after the keys are staged: a tile spends
  // microseconds waiting for its predecessors (the 64 tiles of a dispatch group start together;
  // tools/os_timeline.py), and the staging needs none of what the walk returns.
  u32 lb_count = 0;
#if defined(GIQL_OS_FAKE_SEG)
  u32 lb_fake = 0;
#endif
  if (tid < OS_BINS) {
    u32 run = 0;
#pragma unroll
    for (int k = 0; k < OS_NW; k++) {
      const u32 c = s_wcnt[k][tid];
      s_wcnt[k][tid] = run;
      run += c;
    }
    const u32 count = run;
    lb_count = count;
    GIQL_TL(tile, 5);  // wave bases done
    // exclusive scan of the 256 digit counts (4 waves)
    const u32 incl = wave_incl_scan(count);
    if (lane == WAVE - 1) s_scan[w] = incl;
#if !(defined(GIQL_ABLATE) && GIQL_ABLATE == 1)
#if defined(GIQL_OS_FAKE_SEG)  // timing-only build (results WRONG): the look-back chain restarts every GIQL_OS_FAKE_SEG tiles,
                               // as it would with per-segment digit bases known up front -- what would that buy?
    // (a segment's first tile starts from the EXPECTED prefix of a shuffled input -- its bin's size x the share of the
    // tiles before it -- so that the destinations spread as the real ones do: without it the segments overwrite each
    // other's lines in L2 and the build flatters itself)
    {
      const u32 n_tiles_ = (n_total + (u32)(OS_NT * OS_ITEMS) - 1u) / (u32)(OS_NT * OS_ITEMS);
      const u32 bin_ = (tid == OS_BINS - 1 ? n_total : gbase[tid + 1]) - gbase[tid];
      const u32 fake_ = (u32)((u64)bin_ * tile / n_tiles_);
      const bool head_ = (tile % GIQL_OS_FAKE_SEG) == 0;
      lb_fake = head_ ? fake_ : 0u;
      __hip_atomic_store(status + (size_t)tile * OS_BINS + tid,
                         (head_ ? OS_FLAG_PREFIX : OS_FLAG_AGG) | ((count + lb_fake) & OS_VALUE_MASK), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
#else
    __hip_atomic_store(status + (size_t)tile * OS_BINS + tid, (tile == 0 ? OS_FLAG_PREFIX : OS_FLAG_AGG) | count,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
#endif
    s_dstart[tid] = incl - count;  // wave-local exclusive; wave base added below
  }
  __syncthreads();
  if (tid < OS_BINS) {
    u32 wb = 0;
#pragma unroll
    for (int k = 0; k < OS_BINS / WAVE; k++)
      if (k < (int)w) wb += s_scan[k];
    s_dstart[tid] += wb;
  }
  __syncthreads();
  GIQL_TL(tile, 6);

  // in-tile sorted position of every item
  u32 pos[OS_ITEMS];
#pragma unroll
  for (int i = 0; i < OS_ITEMS; i++) {
    const u32 d = (key[i] >> shift) & 0xFFu;
    pos[i] = s_dstart[d] + s_wcnt[w][d] + rank[i];
  }
  // first payload array: issue its loads now, they fly under the key round
  u32 pay[OS_ITEMS];
  if (PAYLOAD & 2) {
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 r = wbase + i * WAVE + lane;
      if (KEYGEN)  // the end key from the raw columns (the chrom words were read a moment ago: L2)
        pay[i] = (FULL || r < n_valid) ? s_abase[(u32)cin[r] & 31u] + ein[r] + end_off : 0u;
      else
        pay[i] = (FULL || r < n_valid) ? ein[r] : 0u;
    }
  } else if (PAYLOAD & 1) {
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 r = wbase + i * WAVE + lane;
      pay[i] = (FULL || r < n_valid) ? (rin ? rin[r] : tile_base + r) : 0u;
    }
  }
  // keys into LDS in sorted order (the read-back below waits for the walk's barrier)
#pragma unroll
  for (int i = 0; i < OS_ITEMS; i++) {
    const u32 r = wbase + i * WAVE + lane;
    if (FULL || r < n_valid) s_buf[pos[i]] = key[i];
  }
  GIQL_TL(tile, 7);  // keys staged
  // One payload array: ALL the LDS work comes before the walk -- sorted keys and payload end up
  // in registers, and only the global stores are left once the walk has returned.
  constexpr bool PRESTAGE = GIQL_OS_PRESTAGE && (PAYLOAD == 1 || PAYLOAD == 2);
  u32 ks[PRESTAGE ? OS_ITEMS : 1], ps[PRESTAGE ? OS_ITEMS : 1];
  if (PRESTAGE) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 p = i * OS_NT + tid;
      ks[i] = (FULL || p < n_valid) ? s_buf[p] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 r = wbase + i * WAVE + lane;
      if (FULL || r < n_valid) s_buf[pos[i]] = pay[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 p = i * OS_NT + tid;
      ps[i] = (FULL || p < n_valid) ? s_buf[p] : 0u;
    }
  }
  GIQL_TL(tile, 8);  // every LDS round done (one-payload sorts)
  if (tid < OS_BINS) {
    u32 excl = 0;
#if defined(GIQL_OS_FAKE_SEG)
    excl = lb_fake;
#endif
#if defined(GIQL_ABLATE) && GIQL_ABLATE == 1  // timing-only build: no look-back
    excl = tile * 32;
#else
#if defined(GIQL_OS_FAKE_SEG)
    if ((tile % GIQL_OS_FAKE_SEG) != 0) {
#else
    if (tile != 0) {
#endif
      // look back: the OS_LB_WIDTH nearest predecessors are polled together (independent loads
      // in flight).  The 64 tiles of a dispatch group start together, so a walk crosses ~25
      // aggregate-only tiles (its distance from the group's first tile) before it meets a prefix.
      // 8 words per round: 2 are 5 % slower, 4 are 1 % slower; 16 and 32 spill registers and
      // were 37 % / 78 % slower when the walk still came before the staging.
      u32* st = status + (size_t)tile * OS_BINS + tid;
      const u32 count = lb_count;
      u32 t = tile;  // predecessors t-1, t-2, ...
      u32 spins = 0;
      bool done = false;
#if defined(GIQL_OS_TIMELINE)
      u32 tl_polls = 0;
#endif
      while (!done) {
#if defined(GIQL_OS_TIMELINE)
        tl_polls++;
#endif
        u32 v[OS_LB_WIDTH];
#pragma unroll
        for (int j = 0; j < OS_LB_WIDTH; j++) {
          const u32 tj = t > (u32)j ? t - 1 - j : 0u;  // clamped; tile 0 always holds a PREFIX
          v[j] = __hip_atomic_load(status + (size_t)tj * OS_BINS + tid, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        // consume in order while the words are ready
        int used = 0;
#pragma unroll
        for (int j = 0; j < OS_LB_WIDTH; j++) {
          if (done || used != j) continue;
          if (t <= (u32)j) {  // ran past tile 0 (its PREFIX ended the walk already)
            done = true;
            continue;
          }
          const u32 f = v[j] >> 30;
          if (f == 0) continue;  // not published yet: re-poll from here
          excl += v[j] & OS_VALUE_MASK;
          used = j + 1;
          if (f == 2u) done = true;
        }
        t -= (u32)used;
        if (!done && used == 0) {
          if (++spins > help_after) {
            // predecessor t-1 has published nothing for too long: whatever the reason (its
            // block may not even have been dispatched yet), this block computes it itself
            atomicMin(s_help, t - 1u);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (t == 0) done = true;
      }
      if (done)  // not when the walk was abandoned for a helping round
        __hip_atomic_store(st, OS_FLAG_PREFIX | ((excl + count) & OS_VALUE_MASK), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
#if defined(GIQL_OS_TIMELINE)
      if (tid == 0 && tile < OS_TL_TILES) {
        g_os_tl[(size_t)tile * 16 + 12] = tl_polls;   // poll rounds of digit 0's walk
        g_os_tl[(size_t)tile * 16 + 13] = tile - t;   // predecessors consumed
      }
#endif
    }
#endif
    s_goff[tid] = gbase[tid] + excl - s_dstart[tid];  // global dst = s_goff[d] + in-tile position
  }
  GIQL_TL(tile, 9);  // walk done (thread 0's digit)
  __syncthreads();
  {
    const u32 help = *s_help;  // block-uniform; nothing of this tile has been written yet
    if (help != OS_NO_TILE) return help;
  }
#if defined(GIQL_ABLATE) && GIQL_ABLATE == 3  // timing-only build: no stores
  if (key[0] != 0x12345u) return OS_NO_TILE;
#endif
  if (PRESTAGE) {
    u32* pout = (PAYLOAD & 2) ? ends_out : rids_out;
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 p = i * OS_NT + tid;
      if (FULL || p < n_valid) {
        const u32 d = s_goff[(ks[i] >> shift) & 0xFFu] + p;
        os_store(keys_out + d, ks[i]);
        os_store(pout + d, ps[i]);
      }
    }
    GIQL_TL(tile, 10);  // every store issued
#if defined(GIQL_OS_TIMELINE)
    __builtin_amdgcn_s_waitcnt(0);
    GIQL_TL(tile, 11);  // ... and acknowledged (wave 0's)
#endif
    return OS_NO_TILE;
  }
  // keys out of LDS (staged above, before the walk); remember each output slot's global destination
  u32 dst[OS_ITEMS];
#pragma unroll
  for (int i = 0; i < OS_ITEMS; i++) {
    const u32 p = i * OS_NT + tid;
    dst[i] = 0;
    if (FULL || p < n_valid) {
      const u32 k = s_buf[p];
      dst[i] = s_goff[(k >> shift) & 0xFFu] + p;
      os_store(keys_out + dst[i], k);
    }
  }
  GIQL_TL(tile, 9);  // key round: staged, stores issued
  if (PAYLOAD == 3) {
    // second payload array (rid) loaded while the end round runs
    u32 pay2[OS_ITEMS];
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 r = wbase + i * WAVE + lane;
      pay2[i] = (FULL || r < n_valid) ? (rin ? rin[r] : tile_base + r) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 r = wbase + i * WAVE + lane;
      if (FULL || r < n_valid) s_buf[pos[i]] = pay[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 p = i * OS_NT + tid;
      if (FULL || p < n_valid) os_store(ends_out + dst[i], s_buf[p]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 r = wbase + i * WAVE + lane;
      if (FULL || r < n_valid) s_buf[pos[i]] = pay2[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 p = i * OS_NT + tid;
      if (FULL || p < n_valid) os_store(rids_out + dst[i], s_buf[p]);
    }
  } else if (PAYLOAD != 0) {
    u32* out = (PAYLOAD & 2) ? ends_out : rids_out;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 r = wbase + i * WAVE + lane;
      if (FULL || r < n_valid) s_buf[pos[i]] = pay[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
      const u32 p = i * OS_NT + tid;
      if (FULL || p < n_valid) os_store(out + dst[i], s_buf[p]);
    }
  }
  GIQL_TL(tile, 10);  // every store issued
#if defined(GIQL_OS_TIMELINE)
  __builtin_amdgcn_s_waitcnt(0);
  GIQL_TL(tile, 11);  // ... and acknowledged (wave 0's)
#endif
  return OS_NO_TILE;
}

// OS_NT threads x OS_ITEMS rows per thread = one tile.  Default 1024 x 8: 16 waves
// halve the per-wave serial work of 512 x 16 at the same 8192-row tile.
//
// order 2 (default): tile = xcd_tile_of_block(blockIdx) -- no ticket atomic, and the
// XCD-aware order above.  order 0: tile = atomic ticket.  Neither relies on dispatch
// order for PROGRESS: a block whose look-back finds a predecessor silent for
// help_after polls abandons its own tile (nothing of it has been written), computes
// that predecessor itself -- recursively the earliest silent one -- and then starts
// over on its own tile.  A tile computed twice (by a helper and, later, by its own
// block) publishes the same status values and writes the same bytes to the same
// addresses, so the duplicate is harmless.
// One tile with the block's LDS prepared (counters zeroed, s_help reset); returns
// OS_NO_TILE when the tile is finished, else the silent predecessor to compute first.
template <int PAYLOAD, int OS_NT, int OS_ITEMS, bool KEYGEN = false>
__device__ __forceinline__ u32 os_run_tile(
    const u32* __restrict__ keys_in, const u32* __restrict__ ends_in, const u32* __restrict__ rids_in,
    u32* __restrict__ keys_out, u32* __restrict__ ends_out, u32* __restrict__ rids_out, u32 n, u32 tile,
    int shift, const u32* __restrict__ gbase, u32* __restrict__ status, DevMeta* __restrict__ meta,
    u32* s_buf, u32 (*s_wcnt)[OS_BINS], u32* s_dstart, u32* s_goff, u32* s_scan, u32* s_help,
    u32 help_after, const u32* s_abase = nullptr, u32 start_off = 0, u32 end_off = 0, const bool unstable = false) {
  constexpr int OS_TILE = OS_NT * OS_ITEMS;
  const u32 tile_base = tile * OS_TILE;
  const u32 n_valid = (n - tile_base) < (u32)OS_TILE ? (n - tile_base) : (u32)OS_TILE;
  if (n_valid == (u32)OS_TILE)
    return onesweep_tile<PAYLOAD, OS_NT, OS_ITEMS, true, KEYGEN>(keys_in, ends_in, rids_in, keys_out, ends_out,
                                                                 rids_out, n_valid, tile, tile_base, shift, gbase,
                                                                 status, meta, s_buf, s_wcnt, s_dstart, s_goff,
                                                                 s_scan, s_help, help_after, n, s_abase, start_off, end_off,
                                                                 unstable);
  return onesweep_tile<PAYLOAD, OS_NT, OS_ITEMS, false, KEYGEN>(keys_in, ends_in, rids_in, keys_out, ends_out,
                                                                rids_out, n_valid, tile, tile_base, shift, gbase,
                                                                status, meta, s_buf, s_wcnt, s_dstart, s_goff,
                                                                s_scan, s_help, help_after, n, s_abase, start_off, end_off,
                                                                unstable);
}

// The cold path: compute the silent predecessor `need` (recursively the earliest silent
// one), then start over on the block's own tile, until that is finished.  Out of line
// so that the hot path of k_onesweep keeps its register allocation.
template <int PAYLOAD, int OS_NT, int OS_ITEMS, bool KEYGEN = false>
__device__ __noinline__ void os_help_loop(
    const u32* __restrict__ keys_in, const u32* __restrict__ ends_in, const u32* __restrict__ rids_in,
    u32* __restrict__ keys_out, u32* __restrict__ ends_out, u32* __restrict__ rids_out, u32 n, u32 own,
    u32 need, int shift, const u32* __restrict__ gbase, u32* __restrict__ status,
    DevMeta* __restrict__ meta, u32* s_buf, u32 (*s_wcnt)[OS_BINS], u32* s_dstart, u32* s_goff,
    u32* s_scan, u32* s_help, u32 help_after, const u32* s_abase = nullptr, u32 start_off = 0, u32 end_off = 0,
    const bool unstable = false) {
  constexpr int OS_NW = OS_NT / WAVE;
  const u32 tid = threadIdx.x;
  u32 cur = need;
  for (u32 rounds = 0;; rounds++) {
    __syncthreads();  // the LDS of the abandoned / finished tile is reused
#pragma unroll
    for (int k = tid; k < OS_NW * OS_BINS; k += OS_NT) (&s_wcnt[0][0])[k] = 0;
    if (tid == 0) *s_help = OS_NO_TILE;
    __syncthreads();
    const u32 r = os_run_tile<PAYLOAD, OS_NT, OS_ITEMS, KEYGEN>(keys_in, ends_in, rids_in, keys_out, ends_out,
                                                               rids_out, n, cur, shift, gbase, status, meta, s_buf,
                                                               s_wcnt, s_dstart, s_goff, s_scan, s_help, help_after,
                                                               s_abase, start_off, end_off, unstable);
    if (r == OS_NO_TILE) {
      if (cur == own) return;
      cur = own;  // the helped tile is done: start over on this block's own tile
    } else {
      cur = r;
      if (rounds > (1u << 20)) {  // cannot happen: every helping round completes a tile
        if (tid == 0) meta->status = -2;
        return;
      }
    }
  }
}

// OS_NT threads x OS_ITEMS rows per thread = one tile.  Default 1024 x 8: 16 waves
// halve the per-wave serial work of 512 x 16 at the same 8192-row tile.
//
// order 2 (default): tile = xcd_tile_of_block(blockIdx) -- no ticket atomic, and the
// XCD-aware order above.  order 0: tile = atomic ticket.  Neither relies on dispatch
// order for PROGRESS: a block whose look-back finds a predecessor silent for
// help_after polls abandons its own tile (nothing of it has been written), computes
// that predecessor itself -- recursively the earliest silent one -- and then starts
// over on its own tile (os_help_loop).  A tile computed twice (by a helper and, later,
// by its own block) publishes the same status values and writes the same bytes to the
// same addresses, so the duplicate is harmless.
template <int PAYLOAD, int OS_NT, int OS_ITEMS, bool KEYGEN = false>
__global__ __launch_bounds__(OS_NT, (OS_NT == 1024 && (PAYLOAD != 3 || OS_ITEMS <= 8)) ? 8 : 1) void k_onesweep(
    const u32* __restrict__ keys_in, const u32* __restrict__ ends_in, const u32* __restrict__ rids_in,
    u32* __restrict__ keys_out, u32* __restrict__ ends_out, u32* __restrict__ rids_out, u32 n,
    int shift, const u32* __restrict__ gbase, u32* __restrict__ status, u32* __restrict__ ticket,
    DevMeta* __restrict__ meta, int order, u32 help_after, const u32* __restrict__ abase = nullptr,
    u32 start_off = 0, u32 end_off = 0, int unstable = 0) {
  constexpr int OS_TILE = OS_NT * OS_ITEMS;
  constexpr int OS_NW = OS_NT / WAVE;
  __shared__ u32 s_buf[OS_TILE];          // staging, reused for key / end / rid
  __shared__ u32 s_wcnt[OS_NW][OS_BINS];  // per-wave digit counters -> bases
  __shared__ u32 s_dstart[OS_BINS];
  __shared__ u32 s_goff[OS_BINS];
  __shared__ u32 s_scan[OS_BINS / WAVE + 1];
  __shared__ u32 s_tile;
  __shared__ u32 s_help;
  __shared__ u32 s_abase[KEYGEN ? 32 : 1];

  const u32 tid = threadIdx.x;
  if (KEYGEN && tid < 32) s_abase[tid] = abase[tid];
  if (tid == 0) {
    if (order != 2) s_tile = atomicAdd(ticket, 1u);
    s_help = OS_NO_TILE;
  }
#pragma unroll
  for (int k = tid; k < OS_NW * OS_BINS; k += OS_NT) (&s_wcnt[0][0])[k] = 0;
  // order 2: every thread knows its tile from blockIdx, so the tile's key loads are issued
  // BEFORE the barrier that makes this LDS set-up visible (onesweep_tile holds that barrier,
  // between issuing the loads and first using them): ~1 us per tile
  if (order != 2) __syncthreads();
  const u32 own = order == 2 ? xcd_tile_of_block<OS_GROUP>(blockIdx.x, gridDim.x) : s_tile;
  if (own * OS_TILE >= n) return;  // block-uniform (cannot happen: grid = n_tiles)
  GIQL_TL(own, 0);  // block started, LDS counters zeroed
  const u32 need = os_run_tile<PAYLOAD, OS_NT, OS_ITEMS, KEYGEN>(keys_in, ends_in, rids_in, keys_out, ends_out,
                                                                rids_out, n, own, shift, gbase, status, meta, s_buf,
                                                                s_wcnt, s_dstart, s_goff, s_scan, &s_help, help_after,
                                                                s_abase, start_off, end_off, unstable != 0);
  if (need != OS_NO_TILE)
    os_help_loop<PAYLOAD, OS_NT, OS_ITEMS, KEYGEN>(keys_in, ends_in, rids_in, keys_out, ends_out, rids_out, n, own,
                                                   need, shift, gbase, status, meta, s_buf, s_wcnt, s_dstart,
                                                   s_goff, s_scan, &s_help, help_after, s_abase, start_off, end_off,
                                                   unstable != 0);
}

// A side that arrives SORTED on the linear axis (coordinate-sorted BED / BAM-derived tables; the span pass
// tells: DevMeta::unsorted_a / _b) takes no scatter pass at all: one streaming pass writes what the sort would
// have left in buffer 0 -- the keys (and end keys) in input order and the identity row ids.  KEYGEN layout
// (2^24-aligned bases, k_onesweep<.., KEYGEN>): key = abase[chrom] + start + start_off.
template <int PAYLOAD>
__global__ __launch_bounds__(256) void k_keygen_stream(const int* __restrict__ chrom, const int* __restrict__ start,
                                                        const int* __restrict__ end, u32 n,
                                                        const u32* __restrict__ abase, u32 start_off, u32 end_off,
                                                        u32* __restrict__ keys, u32* __restrict__ ends,
                                                        u32* __restrict__ rids) {
  __shared__ u32 s_abase[32];
  if (threadIdx.x < 32) s_abase[threadIdx.x] = abase[threadIdx.x];
  __syncthreads();
  const u32 stride = gridDim.x * 256u;
  for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += stride) {
    const u32 b = s_abase[(u32)chrom[i] & 31u];
    keys[i] = b + (u32)start[i] + start_off;
    if (PAYLOAD & 2) ends[i] = b + (u32)end[i] + end_off;
    if (PAYLOAD & 1) rids[i] = i;
  }
}

// ... and a side whose keys the linearize pass already wrote in input order only needs its row ids.
__global__ __launch_bounds__(256) void k_iota(u32* __restrict__ out, u32 n) {
  const u32 stride = gridDim.x * 256u;
  for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += stride) out[i] = i;
}

}  // namespace giql
