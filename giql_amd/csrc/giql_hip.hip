// giql_amd/csrc/giql_hip.hip -- C ABI of libgiql_hip.so (see include/giql_hip.h).
//
// Host-side orchestration only: arena carving, kernel launches on the caller's
// stream, one pinned-memory readback per join (the output size).  All arithmetic
// is in the *.hip.h kernels.  gfx950 only; there is no CPU fallback: without a
// HIP device every entry point fails with GIQL_ERR_HIP.
#include <hip/hip_runtime.h>
#include <chrono>
#include <atomic>
#include <sys/mman.h>
#include <utility>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <mutex>
#include <thread>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/giql_hip.h"
#include "aux_kernels.hip.h"
#include "take_kernels.hip.h"
#include "select_kernels.hip.h"
#include "cluster_kernels.hip.h"
#include "dev_common.hip.h"
#include "join_kernels.hip.h"
#include "onesweep.hip.h"
#include "bucket_sort.hip.h"
#include "radix_sort.hip.h"
#include "scan.hip.h"
#include "probe_kernels.hip.h"

using namespace giql;

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

static int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return set_err(GIQL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                     __FILE__, __LINE__);                                                 \
  } while (0)

#define GIQL_TRY(expr)      \
  do {                      \
    int _rc = (expr);       \
    if (_rc != GIQL_OK) return _rc; \
  } while (0)

// tile shapes per class: class 1 = many queries with ~0.5 match each (B rows as
// queries), class 2 = fewer queries with tens of matches each (A rows)
#ifndef GIQL_RC_ITEMS
#define GIQL_RC_ITEMS 1
#endif
constexpr int RC_ITEMS_C2 = GIQL_RC_ITEMS;  // queries per thread of a count block
#ifndef GIQL_FILL_ITEMS
#define GIQL_FILL_ITEMS 16
#endif
constexpr int FILL_ITEMS_C2 = GIQL_FILL_ITEMS;  // pairs per thread of a fill block (tile = FILL_NT x this)

// grid cap of the gather-bound grid-stride kernels (take, mark, segment sum, checksum).  Unlike the
// streaming min/max and linearize passes (best at ONE resident wave of blocks), these were 2-5 %
// faster with 16384 blocks than with 2048.
#ifndef GIQL_STREAM_GRID
#define GIQL_STREAM_GRID 16384u
#endif

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

// ----------------------------------------------------------------- context
struct SortBufs {
  u32* key[2];
  u32* end[2];
  u32* rid[2];
};

// Device pointers kept between inner_plan and inner_fill (all inside the arena).
struct InnerState {
  SortBufs sa, sb;   // sorted (key, end, rid) of A / B in buffer 0
  u32 nt1 = 0, nt2 = 0;
  int c1_items = C1_ITEMS_MAX;  // class-1 rows per thread of this plan (k_c1_count / k_c1_emit)
  u32* wlo1 = nullptr;   // class-1 S-window starts per block
  u64* c1_base = nullptr;  // class-1 output base per block
  // small B sides: class 1 takes the same count -> scan -> merge-path fill as class 2 (per-row arrays are
  // cheap there, and the fill stays coalesced when rows have many matches; k_c1_emit does not)
  bool c1_fill = false;
  u32 *lo1 = nullptr, *cnt1 = nullptr;
  u64* off1 = nullptr;
  u32* wlo2 = nullptr;
  u32* lo2 = nullptr;    // class-2 first matching B index per A row
  u32* cnt2 = nullptr;   // ... and the number of matches (kept for giql_hip_inner_plan_export_dev)
  u64* off2 = nullptr;   // class-2 exclusive output offsets
  // uniform-length form: 0 = general two-class join; 1 = B is uniform (queries =
  // A rows); 2 = A is uniform (queries = B rows)
  int uniform = 0;
};

struct giql_hip_ctx {
  int device = 0;
  char* arena = nullptr;
  size_t arena_cap = 0;
  DevMeta* d_meta = nullptr;
  DevMeta* h_meta = nullptr;  // pinned
  u32* part = nullptr;        // fill partition (grown on demand)
  void* stage_out = nullptr;  // device staging of the host-buffer entry points' outputs (grown on demand)
  size_t stage_out_cap = 0;
  size_t part_cap = 0;
  u64* d_scratch64 = nullptr;  // small device scratch (checksum)

  bool classic_sort = false;  // GIQL_HIP_SORT=classic: three-launch radix passes
  int os_variant = 0;         // GIQL_HIP_OS_VARIANT: onesweep block shape (tuning)
  int c1_items = 0;           // GIQL_HIP_C1_ITEMS: class-1 rows per thread, 2 or 8 (0 = by size)
  size_t c1_small_rows = (size_t)4 << 20;  // class-1 sides up to this many rows take 2 rows per thread
  bool no_uniform = false;    // GIQL_HIP_NO_UNIFORM=1: always run the general two-class join
  int n_cu = 256;             // compute units of the device
  int os_order = 2;           // onesweep tile order (GIQL_HIP_OS_ORDER, see k_onesweep)
  u32 os_help_after = OS_HELP_AFTER;  // look-back polls before a block helps (GIQL_HIP_OS_HELP_AFTER)
  // fused join (giql_hip_inner_join_dev): outputs offered to the plan for a fill launched
  // before the host has learned the pair count
  int32_t* fuse_a = nullptr;
  int32_t* fuse_b = nullptr;
  u64 fuse_cap = 0;
  bool fuse_done = false;
  bool no_c1_fill = false;  // GIQL_HIP_NO_C1_FILL=1: class 1 always through k_c1_count / k_c1_emit
  bool no_keygen_general = false;  // GIQL_HIP_NO_KEYGEN_GENERAL=1: the general form always linearises both sides
  bool swapped = false;  // the last INNER plan ran with the sides exchanged (giql_hip_inner_plan_dev_impl)
  bool no_swap = false;  // GIQL_HIP_NO_SWAP=1: plan the sides as given
  bool last_no_irr = false;   // the previous plan met no irregular row
  bool nearest_two_sorts = false;  // NEAREST: a B table with long equal-start runs was seen
  // NEAREST k = 1: both sides sorted straight from their raw columns (digits counted in the span pass, aligned layout;
  // no linearize pass) -- -1: not known yet (the next call probes the layout), 0: this data does not take the aligned
  // layout, 1: the previous call did (a guess, validated at the read-back)
  int nearest_aligned = -1;
  bool spec_valid = false;    // INNER: the previous plan's form decision, speculated on next time
  int spec_form = 0;
  i64 spec_len = 0;
  int spec_misses = 0;
  // per-row operators (SEMI / ANTI / COUNT): the fixed length of B seen by the previous call
  // (0 = B was not fixed-length), speculated on like the INNER form
  bool row_spec_valid = false;
  i64 row_spec_len = 0;
  bool spec_aligned = false;  // ... and whether the aligned layout (histogram in the span pass) held
  bool no_span_hist = false;  // GIQL_HIP_NO_SPAN_HIST=1: always linearize the fixed-length side (A/B aid)
  int inject_timeout = 0;     // test hook: report one look-back timeout
  int order_fallbacks = 0;    // calls repeated in the ticket order after a timeout
  // three-stage sort (two global passes on bits 16-31 + the in-LDS bucket sort, bucket_sort.hip.h)
  // for sides of at least local_min_rows rows (smaller sides have too few rows per bucket to pay for a block each); switched off for good on a context once a bucket
  // turned out larger than the LDS sort holds (GIQL_HIP_NO_LOCAL_SORT=1: never)
  // second stream: the smaller side's linearize + sort run beside the larger side's when that side is
  // small enough to be latency-bound (a chain of ~10 short launches); GIQL_HIP_NO_OVERLAP=1: never
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  u64 overlap_max_rows = 4u << 20;
  bool overlap_large = true;  // GIQL_HIP_OVERLAP_LARGE=0: the small side's chain runs beside the other side's only when that is small too
  int overlap_mask = 3;  // GIQL_HIP_OVERLAP_MASK: 1 = the sides' sort chains, 2 = the two count classes
  int row_skip_digits = -1;    // GIQL_HIP_ROW_SKIP_DIGITS: low digits the per-row operators leave unsorted on their query side (-1: by density, row_skip())
  u64 last_span = 0;           // linearised span of the context's last call: the density guess of row_skip() / sort_is_local()
  double local_max_bucket_rows = 2800.0;  // three-stage sort only while a 16-bit bucket holds at most this many rows on average
  double local_min_bucket_rows = 300.0;   // ... and at least this many (below: a block per bucket is mostly overhead)
  // denser tables keep the form with NARROWER buckets (15 / 14 / 13 key bits, three global passes): sort_local_bits().
  // GIQL_HIP_NO_NARROW_BUCKETS=1: 16 bits or nothing (round 3); GIQL_HIP_LOCAL_BITS=w: every three-stage sort takes w (tests)
  bool no_narrow = false;
  int force_bits = 0;
  int index_bits = 16;  // the width of the index being built (force_local > 0)
  bool no_skip_digit = false;  // GIQL_HIP_NO_SKIP_DIGIT=1: query sides are sorted on every digit
  // a sort's FIRST pass may rank its rows with LDS atomics (unstable: rows of equal digits in any order) when the caller
  // does not need equal keys in input order -- the INNER join's sides (onesweep.hip.h); GIQL_HIP_NO_UNSTABLE_FIRST=1: never
  bool first_unstable = false;
  bool no_unstable_first = false;
  bool no_dual_span = false;   // GIQL_HIP_NO_DUAL_SPAN=1: one span launch per side (round 3)
  bool no_coarse_b = false;    // GIQL_HIP_NO_COARSE_B=1: the fixed-length B of SEMI / ANTI / COUNT is sorted on every digit
  double coarse_max_group_rows = 8.0;  // ... and coarsely only while the rows sharing their upper 24 key bits are at most this many on average
  bool local_sort = true;
  int force_local = 0;         // +1 while a table index is built (giql_hip_index_create_dev): the three-stage sort whatever the guesses say; -1: global passes only
  u64 local_min_rows = 1u << 21;  // (round 4: a floor only; the density bounds above decide)
  int local_resorts = 0;      // calls repeated with the four-pass sort
  bool last_sort_local = false;  // the call in flight sorted at least one side in three stages
  int last_local_bits = 16;      // ... the key bits of its buckets (the last bucket stage launched)
  // fused range count (fixed-length INNER form whose sorted side takes the three-stage sort): the bucket
  // sort answers the queries' bounds from LDS, the sorted keys never return to HBM (bucket_sort.hip.h)
  // sorted inputs: a side the span pass found in (chrom id, start) order is not sorted again.  The answer of the
  // previous plan is the guess of the next (validated at the read-back); plan labels (after the exchange of sides)
  bool spec_sorted[2] = {false, false};
  bool used_sorted[2] = {false, false};  // the call in flight skipped that side's sort
  bool no_sorted = false;       // GIQL_HIP_NO_SORTED_INPUT=1: every side is sorted whatever its order
  bool no_keygen_q = false;     // GIQL_HIP_NO_KEYGEN_Q=1: the fixed-length form always linearizes its query side
  u32* span_hist_dirty[2] = {nullptr, nullptr};  // histograms the span pass of the call in flight counted into (a side that is linearized after all must zero its own again)
  bool prezeroed = false;       // the call in flight zeroed its histograms and status words in ONE memset up front
  bool no_fuse_count = false;   // GIQL_HIP_NO_FUSE_COUNT=1: the separate count kernel always
  int fuse_q_skip = 2;          // GIQL_HIP_Q_SKIP_DIGITS: low digits the fused form leaves unsorted on the query side
  bool spec_fuse_len_ok = false;  // the previous plan's query rows were all short enough for the fused windows
  bool count_fused = false;     // the call in flight answered its bounds in the bucket sort
  // the join itself in the bucket stage (bucket_sort.hip.h, FUSE == 2): one-call form only, on the same guesses as the
  // early fill; the pairs leave from the bucket blocks, no sorted id / bound / offset array is written
  bool no_bucket_join = false;  // GIQL_HIP_NO_BUCKET_JOIN=1: bounds from the bucket sort, then scan + fill as before
  bool bucket_join = false;     // the call in flight emitted its pairs from the bucket stage
  bool plan_is_join = false;    // ... and so left no plan arrays behind (fill / export need a plan of their own)
  u32* bucket_qwin = nullptr;   // [2 * BS_MAX_BUCKETS] query window per bucket
  u32* bucket_bnd = nullptr;  // [BS_MAX_BUCKETS + 1] bucket boundaries of the sort in flight
  u32* bucket_big = nullptr;  // [1 + BS_MAX_BUCKETS] buckets too large for LDS, queued for k_bucket_sort_big ([0] = count)
  char* xplan = nullptr;      // scratch of giql_hip_fill_from_plan_dev (offsets + scan partials), grown on demand
  size_t xplan_cap = 0;

  // profiling
  int profiling = 0;          // 0 off, 1 every phase, 2 only profile_phase (the dominant kernel's)
  int profile_phase = GIQL_PH_SORT_SCATTER;
  std::vector<hipEvent_t> ev_pool;
  struct Span {
    int phase;
    hipEvent_t a, b;
  };
  std::vector<Span> spans;
  size_t ev_used = 0;
  giql_hip_stats stats;

  // state kept between inner_plan and inner_fill
  bool planned = false;
  giql_side side_a, side_b;
  u32 n_a = 0, n_b = 0;
  int n_chrom = 0;
  u64 n_reg = 0, n_irr = 0, n_c1 = 0;
  InnerState inner;
  u32* irr_a_list = nullptr;
  u32* irr_b_list = nullptr;
  u64* irr_off = nullptr;
};

struct Carver {
  char* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

static int ensure_arena(giql_hip_ctx* ctx, size_t bytes, hipStream_t stream) {
  if (bytes <= ctx->arena_cap) return GIQL_OK;
  HIP_TRY(hipStreamSynchronize(stream));
  if (ctx->arena) HIP_TRY(hipFree(ctx->arena));
  ctx->arena = nullptr;
  ctx->arena_cap = 0;
  ctx->planned = false;
  const size_t want = align_up(bytes + bytes / 8, (size_t)1 << 20);
  hipError_t e = hipMalloc((void**)&ctx->arena, want);
  if (e != hipSuccess)
    return set_err(GIQL_ERR_NOMEM, "hipMalloc(%zu bytes) for the workspace failed: %s", want,
                   hipGetErrorString(e));
  ctx->arena_cap = want;
  if (getenv("GIQL_HIP_DEBUG_ADDR")) fprintf(stderr, "[giql_hip] arena %p + %zu bytes\n", (void*)ctx->arena, want);
  return GIQL_OK;
}

// ---------------------------------------------------------------- profiling
struct Phase {
  giql_hip_ctx* ctx;
  hipStream_t stream;
  int phase;
  hipEvent_t a = nullptr, b = nullptr;
  Phase(giql_hip_ctx* c, hipStream_t s, int ph, int launches = 1) : ctx(c), stream(s), phase(ph) {
    ctx->stats.phase_launches[ph] += launches;
    if (!ctx->profiling || (ctx->profiling == 2 && ph != ctx->profile_phase)) return;
    while (ctx->ev_used + 2 > ctx->ev_pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      ctx->ev_pool.push_back(e);
    }
    a = ctx->ev_pool[ctx->ev_used++];
    b = ctx->ev_pool[ctx->ev_used++];
    (void)hipEventRecord(a, stream);
  }
  ~Phase() {
    if (!a) return;
    (void)hipEventRecord(b, stream);
    ctx->spans.push_back({phase, a, b});
  }
};

static void reset_stats(giql_hip_ctx* ctx) {
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  ctx->last_sort_local = false;
  ctx->last_local_bits = 16;
  ctx->count_fused = false;
  ctx->bucket_join = false;
  ctx->used_sorted[0] = ctx->used_sorted[1] = false;
  ctx->spans.clear();
  ctx->ev_used = 0;
}

// Call only after the stream has been synchronised.
static void collect_spans(giql_hip_ctx* ctx) {
  if (!ctx->profiling) return;
  for (auto& s : ctx->spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) ctx->stats.phase_ms[s.phase] += ms;
  }
  ctx->spans.clear();
  ctx->ev_used = 0;
  float t = 0.f;
  for (int i = 0; i < GIQL_PH_N; i++) t += ctx->stats.phase_ms[i];
  ctx->stats.total_ms = t;
  ctx->stats.profiled = 1;
}

// ------------------------------------------------------------ small helpers
static int check_side(const giql_side* s, const char* name) {
  if (!s) return set_err(GIQL_ERR_INVALID, "side %s is NULL", name);
  if (s->n < 0 || s->n > 0x7FFFFFF0ll)
    return set_err(GIQL_ERR_INVALID, "side %s: n=%lld outside [0, 2^31-16]", name, (long long)s->n);
  if (s->n > 0 && (!s->chrom || !s->start || !s->end))
    return set_err(GIQL_ERR_INVALID, "side %s: NULL column buffer", name);
  if (s->start_off < -1 || s->start_off > 0 || s->end_off < -1 || s->end_off > 1)
    return set_err(GIQL_ERR_INVALID, "side %s: canonical offsets (%d,%d) outside {0,-1}x{+1,0,-1}",
                   name, s->start_off, s->end_off);
  return GIQL_OK;
}

static SideView view_of(const giql_side& s) {
  SideView v;
  v.chrom = s.chrom;
  v.start = s.start;
  v.end = s.end;
  v.n = (u32)s.n;
  v.start_off = s.start_off;
  v.end_off = s.end_off;
  return v;
}

static int post_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return set_err(GIQL_ERR_HIP, "kernel launch failed in %s: %s", what, hipGetErrorString(e));
  return GIQL_OK;
}

// exclusive scan of u32 counts; TOut in {u32,u64}; in-place allowed for u32.
// total_out2: a second place for the total (DevMeta::n_out ...), written by the spine kernel itself.
template <typename TOut>
static int run_scan(giql_hip_ctx* ctx, hipStream_t st, int phase, const u32* in, u64 n, TOut* out,
                    u64* bsums, u64* total_out, u64* total_out2 = nullptr) {
  if (n == 0) {
    if (total_out) HIP_TRY(hipMemsetAsync(total_out, 0, sizeof(u64), st));
    if (total_out2) HIP_TRY(hipMemsetAsync(total_out2, 0, sizeof(u64), st));
    return GIQL_OK;
  }
  const u32 nb = cdiv(n, SCAN_TILE);
  Phase ph(ctx, st, phase, 3);
  hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_NT), 0, st, in, n, bsums);
  hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, st, bsums, nb, total_out, total_out2);
  hipLaunchKernelGGL((k_scan_down<TOut>), dim3(nb), dim3(SCAN_NT), 0, st, in, n, bsums, out);
  return post_launch("scan");
}

// The same over counts given as two bounds per row (cnt = hi - lo below the regular prefix, 0 past it):
// what the fused range count leaves.  The down-sweep rewrites `hi` as the counts.
static int run_scan_diff(giql_hip_ctx* ctx, hipStream_t st, int phase, u32* hi, u32* lo, u64 n, const u32* n_irr,
                         u64* out, u64* bsums, u64* total_out, u64* total_out2) {
  const u32 nb = cdiv(n, SCAN_TILE);
  Phase ph(ctx, st, phase, 3);
  hipLaunchKernelGGL(k_scan_reduce_diff, dim3(nb), dim3(SCAN_NT), 0, st, hi, lo, n, n_irr, bsums);
  hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, st, bsums, nb, total_out, total_out2);
  hipLaunchKernelGGL(k_scan_down_diff, dim3(nb), dim3(SCAN_NT), 0, st, hi, lo, n, n_irr, bsums, out);
  return post_launch("scan (bounds)");
}

// One launch: chained scan of the same bounds (status + ticket zeroed by the caller's previous kernel); out[n] =
// total, also to total_out2; part (optional): the fill's merge-path partition for tiles of 2^tile_log2 outputs.
static int run_scan_chain(giql_hip_ctx* ctx, hipStream_t st, int phase, const u32* hi, const u32* lo, u64 n,
                          const u32* n_irr, u64* out, u64* chain_status, u64* total_out2, u32* part, u32 tile_log2,
                          u32 part_cap) {
  const u32 nb = cdiv(n, SC_TILE);
  Phase ph(ctx, st, phase, 1);
  hipLaunchKernelGGL(k_scan_chain_diff, dim3(nb), dim3(SC_NT), 0, st, hi, lo, (u32)n, n_irr, chain_status,
                     reinterpret_cast<u32*>(chain_status + nb + 1), out, total_out2, part, tile_log2, part_cap);
  return post_launch("scan (chained)");
}

// Spans + linearised keys for both sides.
struct LinBufs {
  int* gmin;
  int* gmax;
  i64* chrom_base;
  u32* chrom_first;
  int* len_part;  // [2 sides][MM_MAX_BLOCKS][min,max]
  // histogram-in-the-span-pass form (INNER plan only; NULL elsewhere)
  u32* abase = nullptr;        // [MM_HIST_CHROMS] 2^24-aligned chromosome bases
  u32* top_partial = nullptr;  // [LIN_HIST_REPLICAS][MM_HIST_CHROMS][256]
  // the OTHER side's (the query side of the fixed-length form, when it is sorted from its raw columns too)
  u32* top_partial2 = nullptr;
  u32* hist_partial2 = nullptr;
};

static inline int sort_local_bits(const giql_hip_ctx* ctx, size_t n);
static inline bool sort_is_local(const giql_hip_ctx* ctx, size_t n) { return sort_local_bits(ctx, n) != 0; }

// hist_side = 0 / 1 (with hist_partial and lb.abase / lb.top_partial): that side's span pass
// also counts the digits of its aligned keys (k_chrom_minmax<true>) and the chromosome bases
// are laid out 2^24-aligned when they fit (meta->aligned_ok).
static int run_spans(giql_hip_ctx* ctx, hipStream_t st, const giql_side& a, const giql_side& b,
                     int n_chrom, const LinBufs& lb, int hist_side = -1, u32* hist_partial = nullptr) {
  const bool hist = hist_side >= 0 && hist_partial && lb.abase && lb.top_partial &&
                    n_chrom <= MM_HIST_CHROMS && (hist_side ? b.n : a.n) > 0;
  // both sides count their digits (lb.hist_partial2 / top_partial2 for the other one; prezeroed plans only)
  const bool hist2 = hist && lb.hist_partial2 && lb.top_partial2 && ctx->prezeroed && (hist_side ? a.n : b.n) > 0;
  if (hist && !ctx->prezeroed) {
    HIP_TRY(hipMemsetAsync(hist_partial, 0, (size_t)LIN_HIST_REPLICAS * 1024 * sizeof(u32), st));
    HIP_TRY(hipMemsetAsync(lb.top_partial, 0, (size_t)LIN_HIST_REPLICAS * MM_TOP_WORDS * sizeof(u32), st));
  }
  Phase ph(ctx, st, GIQL_PH_SPAN, 4);
  hipLaunchKernelGGL(k_init_minmax, dim3(cdiv((u64)n_chrom + 1, 256)), dim3(256), 0, st, lb.gmin,
                     lb.gmax, n_chrom, ctx->d_meta);
  const size_t lds = n_chrom <= MM_LDS_CHROMS ? (size_t)n_chrom * 2 * sizeof(int) : 0;
  const giql_side* sides[2] = {&a, &b};
  int nblk[2] = {0, 0};
  if (a.n > 0 && b.n > 0 && !ctx->no_dual_span) {
    // both sides in ONE launch: the grid (one resident wave of 512-thread blocks) is shared in proportion to the rows
    MmSide ms[2];
    const double tot = (double)a.n + (double)b.n;
    for (int k = 0; k < 2; k++) {
      const giql_side& s = *sides[k];
      const bool with_hist = hist && (k == hist_side || hist2);
      u32* const hp = k == hist_side ? hist_partial : lb.hist_partial2;
      if (with_hist) ctx->span_hist_dirty[k] = hp;
      u32 grid = (u32)((double)MM_MAX_BLOCKS * (double)s.n / tot + 0.5);
      const u32 need = cdiv((u64)s.n, (u64)MM_NT_HIST * 4);   // a block per tile at most
      if (grid > need) grid = need;
      if (grid < 1) grid = 1;
      nblk[k] = (int)grid;
      ms[k].chrom = s.chrom;
      ms[k].start = s.start;
      ms[k].end = s.end;
      ms[k].n = (i64)s.n;
      ms[k].len_bias = s.end_off - s.start_off;
      ms[k].start_off = with_hist ? s.start_off : 0;
      // (16-bit buckets: the low digits are sorted in LDS and not counted; narrower ones need the bits 8-15 digit)
      ms[k].hist = !with_hist ? 0 : (sort_local_bits(ctx, (size_t)s.n) == 16 ? 2 : 1);
      ms[k].nblk = grid;
      ms[k].hist_partial = with_hist ? hp : nullptr;
      ms[k].top_partial = with_hist ? (k == hist_side ? lb.top_partial : lb.top_partial2) : nullptr;
    }
    hipLaunchKernelGGL((k_chrom_minmax2<MM_NT_HIST>), dim3(ms[0].nblk + ms[1].nblk), dim3(MM_NT_HIST), lds, st, ms[0], ms[1],
                       n_chrom, lb.gmin, lb.gmax, ctx->d_meta, lb.len_part);
  } else
  for (int k = 0; k < 2; k++) {
    const giql_side& s = *sides[k];
    if (s.n == 0) continue;
    const bool with_hist = hist && (k == hist_side || hist2);
    u32* const hp = k == hist_side ? hist_partial : lb.hist_partial2;
    if (with_hist) ctx->span_hist_dirty[k] = hp;
    u32* const tp = k == hist_side ? lb.top_partial : lb.top_partial2;
    u32 grid = cdiv((u64)s.n, (u64)(with_hist ? MM_NT_HIST : MM_NT) * MM_ITEMS);
    if (grid > (u32)MM_MAX_BLOCKS) grid = MM_MAX_BLOCKS;
    nblk[k] = (int)grid;
    if (with_hist && sort_local_bits(ctx, (size_t)s.n) == 16)  // the low digits are sorted in LDS: not counted
      hipLaunchKernelGGL((k_chrom_minmax<2, MM_NT_HIST>), dim3(grid), dim3(MM_NT_HIST), lds, st, s.chrom, s.start, s.end,
                         (i64)s.n, n_chrom, lb.gmin, lb.gmax, ctx->d_meta, s.end_off - s.start_off, k,
                         lb.len_part, s.start_off, hp, tp);
    else if (with_hist)
      hipLaunchKernelGGL((k_chrom_minmax<1, MM_NT_HIST>), dim3(grid), dim3(MM_NT_HIST), lds, st, s.chrom, s.start, s.end,
                         (i64)s.n, n_chrom, lb.gmin, lb.gmax, ctx->d_meta, s.end_off - s.start_off, k,
                         lb.len_part, s.start_off, hp, tp);
    else
      hipLaunchKernelGGL((k_chrom_minmax<0, MM_NT>), dim3(grid), dim3(MM_NT), lds, st, s.chrom, s.start, s.end,
                         (i64)s.n, n_chrom, lb.gmin, lb.gmax, ctx->d_meta, s.end_off - s.start_off, k,
                         lb.len_part, 0, (u32*)nullptr, (u32*)nullptr);
  }
  int omin = a.start_off, omax = a.start_off;
  const int offs[3] = {a.end_off, b.start_off, b.end_off};
  for (int k = 0; k < 3; k++) {
    if (offs[k] < omin) omin = offs[k];
    if (offs[k] > omax) omax = offs[k];
  }
  hipLaunchKernelGGL(k_chrom_offsets, dim3(1), dim3(256), 0, st, lb.gmin, lb.gmax, n_chrom, omin,
                     omax, lb.chrom_base, lb.chrom_first, ctx->d_meta, lb.len_part, nblk[0], nblk[1],
                     hist ? 1 : 0, lb.abase);
  return post_launch("spans");
}

// hist_partial / gbase non-NULL: also produce the onesweep digit offsets.
static int run_linearize(giql_hip_ctx* ctx, hipStream_t st, const giql_side& s, int n_chrom,
                         const LinBufs& lb, u32* keys, u32* ends, u32* irr_list, int which,
                         int keep_irregular, u32* hist_partial = nullptr, u32* gbase = nullptr,
                         u32* hist_end = nullptr, u32* gbase_end = nullptr, bool skip_end = false) {
  if (s.n == 0) return GIQL_OK;
  const int* end_col = skip_end ? (const int*)nullptr : s.end;  // a side known to hold no irregular row
  // (a plan zeroes its histograms once, up front -- but this one may hold the span pass's counts of a side that
  // turned out to need the linearize pass after all: layout or form guess not taken)
  bool dirty = false;
  for (int k = 0; k < 2; k++)
    if (hist_partial && ctx->span_hist_dirty[k] == hist_partial) {
      dirty = true;
      ctx->span_hist_dirty[k] = nullptr;
    }
  if (hist_partial && (!ctx->prezeroed || dirty))
    HIP_TRY(hipMemsetAsync(hist_partial, 0, (size_t)LIN_HIST_REPLICAS * 1024 * sizeof(u32), st));
  if (hist_end && !ctx->prezeroed)
    HIP_TRY(hipMemsetAsync(hist_end, 0, (size_t)LIN_HIST_REPLICAS * 1024 * sizeof(u32), st));
  Phase ph(ctx, st, GIQL_PH_LINEARIZE, hist_partial ? 2 : 1);
  u32 grid = cdiv((u64)s.n, LIN_NT);
  if (grid > (u32)LIN_MAX_BLOCKS) grid = LIN_MAX_BLOCKS;
  if (hist_end)
    hipLaunchKernelGGL((k_linearize<true>), dim3(grid), dim3(LIN_NT), 0, st, s.chrom, s.start, end_col,
                       (u32)s.n, s.start_off, s.end_off, n_chrom, lb.chrom_base, keys, ends, irr_list,
                       ctx->d_meta, which, keep_irregular, hist_partial, hist_end);
  else
    hipLaunchKernelGGL((k_linearize<false>), dim3(grid), dim3(LIN_NT), 0, st, s.chrom, s.start, end_col,
                       (u32)s.n, s.start_off, s.end_off, n_chrom, lb.chrom_base, keys, ends, irr_list,
                       ctx->d_meta, which, keep_irregular, hist_partial, (u32*)nullptr);
  if (hist_partial)
    hipLaunchKernelGGL(k_digit_offsets, dim3(4), dim3(256), 0, st, hist_partial,
                       (u32)LIN_HIST_REPLICAS, gbase);
  if (hist_end)
    hipLaunchKernelGGL(k_digit_offsets, dim3(4), dim3(256), 0, st, hist_end, (u32)LIN_HIST_REPLICAS,
                       gbase_end);
  return post_launch("linearize");
}

// Onesweep LSD sort (4 passes, one launch each); input and result in
// buffer 0.  Payload of a sort = which of end / rid buffers the SortBufs carries.
// status words of ONE pass: a {flag,count} word per (tile, digit) + the ticket word
static inline size_t os_pass_words(size_t n) {
  return (size_t)cdiv(n ? n : 1, OS_MIN_TILE) * OS_BINS + 16;
}
// ... and what a pass of this context really uses (the default 1024 x 8 block shape has 8192-row tiles: half
// of the worst case, half the bytes to zero); the ticket word is the stride's 16th-last
static inline size_t os_pass_stride(const giql_hip_ctx* ctx, size_t n) {
  return ctx->os_variant == 0 ? (size_t)cdiv(n ? n : 1, 8192) * OS_BINS + 16 : os_pass_words(n);
}

template <int NT, int ITEMS>
static void launch_onesweep(giql_hip_ctx* ctx, hipStream_t st, SortBufs& sb, int src, int dst, bool first,
                            u32 n, int shift, const u32* gbase, u32* status, DevMeta* meta, int unstable = 0) {
  const u32 grid = cdiv(n, NT * ITEMS);  // one block per tile
  u32* claim = status + os_pass_stride(ctx, n) - 16;  // the pass's ticket word
  const u32* rin = (first || !sb.rid[0]) ? (const u32*)nullptr : sb.rid[src];
  const int mode = (sb.rid[0] ? 1 : 0) | (sb.end[0] ? 2 : 0);
#define GIQL_OS_LAUNCH(M)                                                                            \
  hipLaunchKernelGGL((k_onesweep<M, NT, ITEMS>), dim3(grid), dim3(NT), 0, st, sb.key[src],           \
                     sb.end[0] ? sb.end[src] : (const u32*)nullptr, rin, sb.key[dst],                 \
                     sb.end[0] ? sb.end[dst] : (u32*)nullptr, sb.rid[0] ? sb.rid[dst] : (u32*)nullptr, \
                     n, shift, gbase, status, claim, meta, ctx->os_order, ctx->os_help_after, (const u32*)nullptr, 0u, 0u,  \
                     unstable)
  switch (mode) {
    case 0: GIQL_OS_LAUNCH(0); break;
    case 1: GIQL_OS_LAUNCH(1); break;
    case 2: GIQL_OS_LAUNCH(2); break;
    default: GIQL_OS_LAUNCH(3); break;
  }
#undef GIQL_OS_LAUNCH
}

// keep_rids: the rid buffer already holds row ids (second sort of a two-key sort).
// status: 4 * os_pass_words(n) words, zeroed here in one memset.
// keygen (with abase): the first pass builds the keys from that side's raw (chrom, start)
// columns instead of reading sb.key[0] (k_onesweep<.., KEYGEN>; (key, rid) sorts, default
// block shape only).
// Sides of ctx->local_min_rows rows and more take the three-stage form: global passes on bits
// 16-23 and 24-31 only, then every 16-bit bucket sorted on its low bits inside LDS, in place
// (bucket_sort.hip.h) -- three trips through HBM instead of four.
// the narrowest-needed bucket for a table of per_bucket rows per 65,536 keys on average (0: too dense for any)
static inline int density_bits(const giql_hip_ctx* ctx, double per_bucket) {
  int w = 16;
  while (per_bucket > ctx->local_max_bucket_rows && w > BS_MIN_WBITS && !ctx->no_narrow) {
    per_bucket *= 0.5;
    w--;
  }
  return per_bucket <= ctx->local_max_bucket_rows ? w : 0;
}

// Returns 0 (four global passes) or the key bits of a bucket: 16 (two global passes), or 15 / 14 / 13 for denser
// tables (three global passes -- bits 8-15, 16-23, 24-31 -- and buckets of 2^W keys: bucket_sort.hip.h).
static inline int sort_local_bits(const giql_hip_ctx* ctx, size_t n) {
  if (ctx->force_local > 0 && ctx->bucket_bnd && ctx->os_variant == 0) return ctx->index_bits;
  if (ctx->force_local < 0) return 0;
  if (!(ctx->local_sort && ctx->bucket_bnd && ctx->os_variant == 0 && n >= ctx->local_min_rows)) return 0;
  // The in-LDS stage holds 4096 rows per bucket; larger buckets go through a slow queue (one block each, two
  // more passes: 30M x 300M reads, ~6000 rows per bucket, spent 10.4 of 21 ms there).  So the form is taken
  // only while the AVERAGE bucket is comfortably below that -- by the span of the context's previous call
  // (a human-genome-sized axis until one has run); denser tables take the four global passes.  The value is
  // constant during a call (updated when it ends), so every decision of one call agrees.
  // Round 4: what decides is the DENSITY, not the row count -- a 12.5M-row shard of an 8-GPU run has the headline's
  // ~2,100 rows per bucket on an eighth of the axis and takes the same path (fused count, join in the bucket stage);
  // a 10M-row table over the whole genome (212 rows per bucket) does not: a block per bucket is mostly overhead there
  // (0.135 ms against 0.106 for the two passes it replaces).
  // Round 4, bucket width by density: past 2,800 rows per 65,536 keys (132M rows on a human-genome axis) the bucket
  // narrows -- 2^15, 2^14, 2^13 keys, up to ~1G rows -- instead of the form being given up.
  const double span = ctx->last_span ? (double)ctx->last_span : 3.2e9;
  double per_bucket = (double)n * 65536.0 / span;
  if (per_bucket < ctx->local_min_bucket_rows) return 0;
  if (ctx->force_bits) return ctx->force_bits;
  return density_bits(ctx, per_bucket);
}
static inline int local_passes(int wbits) { return wbits == 16 ? 2 : 3; }  // global passes before the bucket stage

// skip_digits = 1 (four-pass form only): the lowest digit is left unsorted -- rows come out ordered
// by key >> 8 and the result is left in buffer 0 by swapping the ping-pong pointers.  For QUERY
// sides whose order only serves locality (neighbouring rows search neighbouring ranges): three
// passes instead of four.
// fuse (three-stage (key, rid) sorts only): the bucket sort also answers the range bounds of the
// fixed-length INNER form's query rows and does not store the sorted keys (bucket_sort.hip.h).
struct FuseCount {
  BsFuse dev;
  u32 nq_total;
  const u32* irr_q;
  const u32* gbq3;      // the query sort's top-digit offsets
  u32 key_mask;         // 0xFFFFFF00 when the query side was sorted without its lowest digit
  const int* len_max_q; // DevMeta: longest regular query row
  u32* zero_ptr;        // words the bounds kernel zeroes on the way (the chained scan's status + ticket)
  u32 zero_words;
  bool join = false;    // FUSE == 2: the bucket blocks write the pairs themselves (dev.row_q / row_s / cap / cursor)
  bool general = false; // FUSE == 3: ... of the two-class join (rows of any length, (key, end, rid) sorts)
  const int* len_max_u = nullptr;  // DevMeta: longest row of the sorted side (general form: how far the windows reach up)
};

static int64_t bucket_stage_fused_bytes(u32 n, const FuseCount& fuse);
// The bucket stage of a fused (key, rid) sort: bounds only (FUSE 1) or the whole join (FUSE 2).  Three launches,
// each a phase of its own so that the stage's ONE heavy kernel can be timed alone: the bucket boundaries and query
// windows (COUNT), the bucket kernel (SORT_LOCAL), the queue of buckets it could not take (AUX; almost always an
// empty launch).
static void launch_bucket_stage_fused(giql_hip_ctx* ctx, hipStream_t st, SortBufs& sb, u32 n, const u32* gbase,
                                      const FuseCount& fuse_in, int wbits = 16) {
  FuseCount fuse = fuse_in;
  fuse.dev.wbits = (u32)wbits;
  const u32 BS_BUCKETS = bs_n_buckets((u32)wbits);  // (shadows the 16-bit constant: every launch below is per bucket)
  ctx->stats.phase_bytes[GIQL_PH_SORT_LOCAL] += bucket_stage_fused_bytes(n, fuse);
  ctx->count_fused = true;
  ctx->bucket_join = fuse.join;
  ctx->last_local_bits = wbits;
  {
    Phase ph(ctx, st, GIQL_PH_COUNT, 1);
    hipLaunchKernelGGL(k_bucket_bounds_fused, dim3(cdiv((u64)3 * BS_BUCKETS + 1, 256)), dim3(256), 0, st, sb.key[0], n,
                       gbase + 3 * OS_BINS, ctx->bucket_bnd, ctx->bucket_big, fuse.dev, fuse.nq_total, fuse.irr_q,
                       fuse.gbq3, fuse.key_mask, fuse.len_max_q, fuse.zero_ptr, fuse.zero_words, fuse.len_max_u);
  }
  const bool w16 = wbits == 16;
#define GIQL_BSF_LAUNCH(P, F, W)                                                                                    \
  hipLaunchKernelGGL((k_bucket_sort<P, F, W>), dim3(BS_BUCKETS), dim3(BS_NT), 0, st, sb.key[0],                      \
                     (P) == 3 ? sb.end[0] : (u32*)nullptr, sb.rid[0], ctx->bucket_bnd, ctx->d_meta, ctx->bucket_big, \
                     fuse.dev)
  if (fuse.general) {
    {
      Phase ph(ctx, st, GIQL_PH_SORT_LOCAL, 1);
      if (w16) GIQL_BSF_LAUNCH(3, 3, true); else GIQL_BSF_LAUNCH(3, 3, false);
    }
    Phase ph(ctx, st, GIQL_PH_AUX, 1);
    hipLaunchKernelGGL((k_bucket_sort_big<3, 3>), dim3(ctx->n_cu), dim3(BS_NT), 0, st, sb.key[0], sb.end[0], sb.rid[0],
                       sb.key[1], sb.end[1], sb.rid[1], ctx->bucket_bnd, ctx->bucket_big, fuse.dev);
    return;
  }
  {
    Phase ph(ctx, st, GIQL_PH_SORT_LOCAL, 1);
    if (fuse.join) {
      if (w16) GIQL_BSF_LAUNCH(1, 2, true); else GIQL_BSF_LAUNCH(1, 2, false);
    } else {
      if (w16) GIQL_BSF_LAUNCH(1, 1, true); else GIQL_BSF_LAUNCH(1, 1, false);
    }
  }
#undef GIQL_BSF_LAUNCH
  {
    Phase ph(ctx, st, GIQL_PH_AUX, 1);
    if (fuse.join)
      hipLaunchKernelGGL((k_bucket_sort_big<1, 2>), dim3(ctx->n_cu), dim3(BS_NT), 0, st, sb.key[0], (u32*)nullptr,
                         sb.rid[0], sb.key[1], (u32*)nullptr, sb.rid[1], ctx->bucket_bnd, ctx->bucket_big, fuse.dev);
    else
      hipLaunchKernelGGL((k_bucket_sort_big<1, 1>), dim3(ctx->n_cu), dim3(BS_NT), 0, st, sb.key[0], (u32*)nullptr,
                         sb.rid[0], sb.key[1], (u32*)nullptr, sb.rid[1], ctx->bucket_bnd, ctx->bucket_big, fuse.dev);
  }
}
// its algorithmic bytes up to the pairs (8 B each, added once the count is known): the rows' keys and ids read;
// bounds form: the ids written, the queries' keys and end keys read, two bounds each written; join form: the
// queries' keys, end keys and ids read
static int64_t bucket_stage_fused_bytes(u32 n, const FuseCount& fuse) {
  if (fuse.general) return (int64_t)12 * n + (int64_t)12 * fuse.nq_total;  // (key, end, rid) of both sides read
  return fuse.join ? (int64_t)8 * n + (int64_t)12 * fuse.nq_total : (int64_t)12 * n + (int64_t)16 * fuse.nq_total;
}

static int run_sort_onesweep(giql_hip_ctx* ctx, hipStream_t st, SortBufs& sb, u32 n,
                             const u32* gbase, u32* status, bool keep_rids = false,
                             const giql_side* keygen = nullptr, const u32* abase = nullptr,
                             int skip_digits = 0, const FuseCount* fuse = nullptr, bool presorted = false) {
  if (n == 0) return GIQL_OK;
  if (presorted) {
    // the side arrives sorted: no scatter pass, one streaming pass for what a sort would have left in buffer 0
    const int mode = (sb.rid[0] ? 1 : 0) | (sb.end[0] ? 2 : 0);
    const int wbits_pre = sort_local_bits(ctx, n);
    const bool local_fused = fuse && mode == (fuse->general ? 3 : 1) && wbits_pre != 0;
    {
      Phase ph(ctx, st, GIQL_PH_SORT_SCATTER, 1);
      u32 grid = cdiv(n, 256 * 8);
      if (grid > GIQL_STREAM_GRID) grid = GIQL_STREAM_GRID;
      if (keygen) {
        ctx->stats.phase_bytes[GIQL_PH_SORT_SCATTER] += (int64_t)4 * n * ((mode & 2 ? 3 : 2) + 1 + (mode & 1) + (mode & 2 ? 1 : 0));
        const u32 so = (u32)keygen->start_off, eo = (u32)keygen->end_off;
        switch (mode) {
          case 0: hipLaunchKernelGGL(k_keygen_stream<0>, dim3(grid), dim3(256), 0, st, keygen->chrom, keygen->start, keygen->end, n, abase, so, eo, sb.key[0], (u32*)nullptr, (u32*)nullptr); break;
          case 1: hipLaunchKernelGGL(k_keygen_stream<1>, dim3(grid), dim3(256), 0, st, keygen->chrom, keygen->start, keygen->end, n, abase, so, eo, sb.key[0], (u32*)nullptr, sb.rid[0]); break;
          case 2: hipLaunchKernelGGL(k_keygen_stream<2>, dim3(grid), dim3(256), 0, st, keygen->chrom, keygen->start, keygen->end, n, abase, so, eo, sb.key[0], sb.end[0], (u32*)nullptr); break;
          default: hipLaunchKernelGGL(k_keygen_stream<3>, dim3(grid), dim3(256), 0, st, keygen->chrom, keygen->start, keygen->end, n, abase, so, eo, sb.key[0], sb.end[0], sb.rid[0]); break;
        }
      } else if (sb.rid[0] && !keep_rids) {
        ctx->stats.phase_bytes[GIQL_PH_SORT_SCATTER] += (int64_t)4 * n;
        hipLaunchKernelGGL(k_iota, dim3(grid), dim3(256), 0, st, sb.rid[0], n);
      }
      GIQL_TRY(post_launch("sorted input (no sort)"));
    }
    if (!local_fused) return GIQL_OK;   // sorted already: the bucket stage only runs as the carrier of the fused count
    ctx->last_sort_local = true;
    launch_bucket_stage_fused(ctx, st, sb, n, gbase, *fuse, wbits_pre);
    return post_launch("bucket stage (sorted input, fused count)");
  }
  const int wbits = sort_local_bits(ctx, n);
  const bool local = wbits != 0;
  if (local) ctx->last_sort_local = true;
  if (local || ctx->no_skip_digit) skip_digits = 0;
  const int n_pass = local ? local_passes(wbits) : 4 - skip_digits;
  const int first_digit = local ? 4 - n_pass : skip_digits;
  const size_t per_pass = os_pass_stride(ctx, n);
  if (!ctx->prezeroed) HIP_TRY(hipMemsetAsync(status, 0, n_pass * per_pass * sizeof(u32), st));
  {
    // one event pair around the passes (an event record between two launches costs the
    // stream ~8 us of idle time; the per-launch time is phase time / launches)
    Phase ph(ctx, st, GIQL_PH_SORT_SCATTER, n_pass);
    for (int pass = 0; pass < n_pass; pass++) {
      const int src = pass & 1, dst = src ^ 1;
      const int digit = first_digit + pass;
      u32* stat = status + pass * per_pass;
      const bool first = pass == 0 && !keep_rids;
      const u32* gb = gbase + digit * OS_BINS;
      {
        // algorithmic bytes of this pass: every array it reads + every array it writes, 4 B per row
        // (KEYGEN reads chrom + start instead of the key; a first pass synthesises the row ids)
        const int w_out = 1 + (sb.rid[0] ? 1 : 0) + (sb.end[0] ? 1 : 0);
        const int w_in = (pass == 0 && keygen) ? 2 + (sb.end[0] ? 1 : 0)
                                               : 1 + ((sb.rid[0] && !first) ? 1 : 0) + (sb.end[0] ? 1 : 0);
        ctx->stats.phase_bytes[GIQL_PH_SORT_SCATTER] += (int64_t)4 * (w_in + w_out) * n;
      }
      const int unstable = (pass == 0 && !keep_rids && ctx->first_unstable && !ctx->no_unstable_first) ? 1 : 0;
      if (pass == 0 && keygen) {
        const u32 grid = cdiv(n, 1024 * 8);
        u32* claim = stat + per_pass - 16;
        if (sb.end[0])  // (key, end, rid): the end keys are built from the raw end column as well
          hipLaunchKernelGGL((k_onesweep<3, 1024, 8, true>), dim3(grid), dim3(1024), 0, st,
                             reinterpret_cast<const u32*>(keygen->start), reinterpret_cast<const u32*>(keygen->end),
                             reinterpret_cast<const u32*>(keygen->chrom), sb.key[dst], sb.end[dst], sb.rid[dst], n,
                             digit * 8, gb, stat, claim, ctx->d_meta, ctx->os_order, ctx->os_help_after, abase,
                             (u32)keygen->start_off, (u32)keygen->end_off, unstable);
        else
          hipLaunchKernelGGL((k_onesweep<1, 1024, 8, true>), dim3(grid), dim3(1024), 0, st,
                             reinterpret_cast<const u32*>(keygen->start), reinterpret_cast<const u32*>(keygen->chrom),
                             (const u32*)nullptr, sb.key[dst], (u32*)nullptr, sb.rid[dst], n, digit * 8, gb, stat, claim,
                             ctx->d_meta, ctx->os_order, ctx->os_help_after, abase, (u32)keygen->start_off, 0u, unstable);
        continue;
      }
      switch (ctx->os_variant) {  // block-shape sweep (tools/os_variants.py); default 1024 x 8
        case 1: launch_onesweep<512, 8>(ctx, st, sb, src, dst, first, n, digit * 8, gb, stat, ctx->d_meta, unstable); break;
        case 2: launch_onesweep<512, 16>(ctx, st, sb, src, dst, first, n, digit * 8, gb, stat, ctx->d_meta, unstable); break;
        case 3: launch_onesweep<256, 16>(ctx, st, sb, src, dst, first, n, digit * 8, gb, stat, ctx->d_meta, unstable); break;
        case 4: launch_onesweep<1024, 4>(ctx, st, sb, src, dst, first, n, digit * 8, gb, stat, ctx->d_meta, unstable); break;
        case 5: case 7: launch_onesweep<1024, 12>(ctx, st, sb, src, dst, first, n, digit * 8, gb, stat, ctx->d_meta, unstable); break;
        default: launch_onesweep<1024, 8>(ctx, st, sb, src, dst, first, n, digit * 8, gb, stat, ctx->d_meta, unstable); break;
      }
    }
  }
  if (n_pass & 1) {  // an odd number of passes ends in buffer 1: make that "buffer 0"
    u32* t;
    t = sb.key[0], sb.key[0] = sb.key[1], sb.key[1] = t;
    t = sb.end[0], sb.end[0] = sb.end[1], sb.end[1] = t;
    t = sb.rid[0], sb.rid[0] = sb.rid[1], sb.rid[1] = t;
  }
  if (local) {
    // the rows are back in buffer 0, ordered by key >> 16: bucket boundaries, then one block per bucket
    const int mode = (sb.rid[0] ? 1 : 0) | (sb.end[0] ? 2 : 0);
    if (fuse && mode == (fuse->general ? 3 : 1)) {
      // (key, rid) rows + the query side's bounds: keys and rids read, rids written, the query rows' keys
      // and ends read and their two bounds written -- the sorted keys never leave the CU
      launch_bucket_stage_fused(ctx, st, sb, n, gbase, *fuse, wbits);
      return post_launch("onesweep sort (fused count)");
    }
    BsFuse bw;  // the plain sort: only the bucket width travels
    bw.wbits = (u32)wbits;
    ctx->last_local_bits = wbits;
    const u32 BS_BUCKETS = bs_n_buckets((u32)wbits);
    HIP_TRY(hipMemsetAsync(ctx->bucket_big, 0, sizeof(u32), st));
    Phase ph(ctx, st, GIQL_PH_SORT_LOCAL, 3);
    ctx->stats.phase_bytes[GIQL_PH_SORT_LOCAL] +=
        (int64_t)8 * (1 + (sb.rid[0] ? 1 : 0) + (sb.end[0] ? 1 : 0)) * n;  // every array read once, written once
    hipLaunchKernelGGL(k_bucket_bounds, dim3(cdiv((u64)BS_BUCKETS + 1, 256)), dim3(256), 0, st, sb.key[0], n,
                       gbase + 3 * OS_BINS, ctx->bucket_bnd, (u32)wbits);
#define GIQL_BS_LAUNCH(M)                                                                              \
  if (wbits == 16)                                                                                     \
    hipLaunchKernelGGL((k_bucket_sort<M, 0, true>), dim3(BS_BUCKETS), dim3(BS_NT), 0, st, sb.key[0],   \
                       sb.end[0] ? sb.end[0] : (u32*)nullptr, sb.rid[0] ? sb.rid[0] : (u32*)nullptr,    \
                       ctx->bucket_bnd, ctx->d_meta, ctx->bucket_big, bw);                             \
  else                                                                                                 \
  hipLaunchKernelGGL((k_bucket_sort<M>), dim3(BS_BUCKETS), dim3(BS_NT), 0, st, sb.key[0],              \
                     sb.end[0] ? sb.end[0] : (u32*)nullptr, sb.rid[0] ? sb.rid[0] : (u32*)nullptr,      \
                     ctx->bucket_bnd, ctx->d_meta, ctx->bucket_big, bw);                               \
  hipLaunchKernelGGL((k_bucket_sort_big<M>), dim3(ctx->n_cu), dim3(BS_NT), 0, st, sb.key[0],           \
                     sb.end[0] ? sb.end[0] : (u32*)nullptr, sb.rid[0] ? sb.rid[0] : (u32*)nullptr,      \
                     sb.key[1], sb.end[0] ? sb.end[1] : (u32*)nullptr, sb.rid[0] ? sb.rid[1] : (u32*)nullptr, \
                     ctx->bucket_bnd, ctx->bucket_big, bw)
    switch (mode) {
      case 0: GIQL_BS_LAUNCH(0); break;
      case 1: GIQL_BS_LAUNCH(1); break;
      case 2: GIQL_BS_LAUNCH(2); break;
      default: GIQL_BS_LAUNCH(3); break;
    }
#undef GIQL_BS_LAUNCH
  }
  return post_launch("onesweep sort");
}

// A two-key sort's FIRST sort runs on a view of the buffers with key and end exchanged.  run_sort_onesweep leaves its
// result in what it then CALLS buffer 0 -- after an odd number of passes (three: a query side sorted without its lowest
// digit, a side with buckets narrower than 2^16 keys) that is the other physical buffer, and it says so by exchanging the
// view's pointers.  The owner of the buffers has to follow, or its second sort starts from the unsorted copy (round 4:
// found by the soak on a context forced to 8,192-key buckets -- ties among equal starts came out in input order).
static inline void adopt_by_end(SortBufs& owner, const SortBufs& by_end) {
  for (int k = 0; k < 2; k++) {
    owner.end[k] = by_end.key[k];
    owner.key[k] = by_end.end[k];
    owner.rid[k] = by_end.rid[k];
  }
}

// Stable LSD radix sort; input in buffer 0, result in buffer 0 (4 passes).
// rids: identity on the first pass.  Payload-less when bufs.end[0] == nullptr.
static int run_sort(giql_hip_ctx* ctx, hipStream_t st, SortBufs& sb, u32 n, u32* tile_hist,
                    u64* bsums) {
  if (n == 0) return GIQL_OK;
  const u32 n_tiles = cdiv(n, RS_TILE);
  for (int pass = 0; pass < 4; pass++) {
    const int src = pass & 1, dst = src ^ 1;
    const int shift = pass * 8;
    {
      Phase ph(ctx, st, GIQL_PH_SORT_HIST);
      hipLaunchKernelGGL(k_radix_hist, dim3(n_tiles), dim3(RS_NT), 0, st, sb.key[src], n, shift,
                         n_tiles, tile_hist);
    }
    GIQL_TRY(run_scan<u32>(ctx, st, GIQL_PH_SORT_SCAN, tile_hist, (u64)n_tiles * RS_BINS, tile_hist,
                           bsums, nullptr));
    {
      Phase ph(ctx, st, GIQL_PH_SORT_SCATTER);
      if (sb.end[0]) {
        hipLaunchKernelGGL((k_radix_scatter<true>), dim3(n_tiles), dim3(RS_NT), 0, st, sb.key[src],
                           sb.end[src], pass == 0 ? (const u32*)nullptr : sb.rid[src], sb.key[dst],
                           sb.end[dst], sb.rid[dst], n, shift, n_tiles, tile_hist);
      } else {
        hipLaunchKernelGGL((k_radix_scatter<false>), dim3(n_tiles), dim3(RS_NT), 0, st, sb.key[src],
                           (const u32*)nullptr, (const u32*)nullptr, sb.key[dst], (u32*)nullptr,
                           (u32*)nullptr, n, shift, n_tiles, tile_hist);
      }
    }
  }
  return post_launch("radix sort");
}

static int run_pmax(giql_hip_ctx* ctx, hipStream_t st, const u32* in, u32 n, u32* out, u32* bmax) {
  if (n == 0) return GIQL_OK;
  const u32 nb = cdiv(n, PM_TILE);
  Phase ph(ctx, st, GIQL_PH_AUX, 3);
  hipLaunchKernelGGL(k_pmax_reduce, dim3(nb), dim3(PM_NT), 0, st, in, n, bmax);
  hipLaunchKernelGGL(k_pmax_spine, dim3(1), dim3(1024), 0, st, bmax, nb);
  hipLaunchKernelGGL(k_pmax_down, dim3(nb), dim3(PM_NT), 0, st, in, n, bmax, out);
  return post_launch("prefix max");
}

static int read_meta(giql_hip_ctx* ctx, hipStream_t st) {
  HIP_TRY(hipMemcpyAsync(ctx->h_meta, ctx->d_meta, sizeof(DevMeta), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  int status = ctx->h_meta->status;
  // test hook (GIQL_HIP_INJECT_TIMEOUT=n): the n-th clean read-back of the context reports a timeout
  if (ctx->inject_timeout > 0 && ctx->os_order != 0 && status == 0 && --ctx->inject_timeout == 0)
    status = ctx->h_meta->status = GIQL_ERR_HIP;
  if (status == GIQL_STATUS_RESORT)  // internal: with_order_fallback repeats the call with the four-pass sort
    return set_err(GIQL_STATUS_RESORT, "a 16-bit key bucket holds more than %u rows", BS_CAP);
  if (status == GIQL_ERR_HIP)
    return set_err(GIQL_ERR_HIP, "onesweep look-back timed out (tile order %d)", ctx->os_order);
  if (status == GIQL_ERR_CHROM)
    return set_err(GIQL_ERR_CHROM, "a chrom id is outside [0, n_chrom)");
  if (status == GIQL_ERR_SPAN)
    return set_err(GIQL_ERR_SPAN,
                   "linearised coordinate span %llu does not fit 32 bits; shard the chromosomes "
                   "into groups (giql_amd.shard) and join each group",
                   (unsigned long long)ctx->h_meta->total_span);
  if (status != 0) return set_err(status, "device reported status %d", status);
  return GIQL_OK;
}

static void common_sizes(Carver& c, int n_chrom, LinBufs& lb) {
  lb.gmin = c.take<int>((size_t)n_chrom);
  lb.gmax = c.take<int>((size_t)n_chrom);
  lb.chrom_base = c.take<i64>((size_t)n_chrom);
  lb.chrom_first = c.take<u32>((size_t)n_chrom + 1);
  lb.len_part = c.take<int>((size_t)2 * MM_MAX_BLOCKS * 2);
}

static void sort_sizes(Carver& c, size_t n, SortBufs& sb, bool payload) {
  for (int k = 0; k < 2; k++) {
    sb.key[k] = c.take<u32>(n);
    sb.end[k] = payload ? c.take<u32>(n) : nullptr;
    sb.rid[k] = payload ? c.take<u32>(n) : nullptr;
  }
}

// Low digits the per-row operators leave unsorted on their QUERY side (its order only serves locality: the
// rows of a block should search neighbouring ranges of B).  One digit always; two (rows ordered by key >> 16
// only: two passes instead of three) when B is sparse enough that a 65536-key range of it -- what a block's
// bracket then covers at least -- still fits the LDS stage of k_nearest (cfg 5: 1.115 -> 1.063 ms).  The
// density comes from the span of the context's previous per-row call: a guess that only ever costs speed.
static inline int row_skip(const giql_hip_ctx* ctx, size_t nb) {
  if (ctx->row_skip_digits >= 0) return ctx->row_skip_digits;
  if (ctx->last_span == 0) return 1;
  return (double)nb * 65536.0 / (double)ctx->last_span <= 1024.0 ? 2 : 1;
}

// Fixed-length B of the per-row operators (SEMI / ANTI / COUNT): sorted without its lowest digit (three passes
// instead of four) when the rows sharing the upper 24 key bits are few enough to be looked at one by one
// (aux_kernels.hip.h, "sorted COARSELY") -- by the density of the context's previous call, like row_skip(): a guess
// that only costs speed.
static inline bool coarse_b_ok(const giql_hip_ctx* ctx, size_t nb) {
  if (ctx->no_coarse_b || ctx->no_skip_digit || ctx->last_span == 0 || sort_is_local(ctx, nb)) return false;
  return (double)nb * 256.0 / (double)ctx->last_span <= ctx->coarse_max_group_rows;
}

// Fork / join of the context's second stream.  A side of a few million rows is a chain of ~10 launches
// of 10-40 us each that do not fill the GPU (look-back latency, not bandwidth, bounds them): run beside
// the other side's chain it costs almost nothing.  Work given to stream() is ordered after everything
// already on the caller's stream; join() orders the caller's stream after it.  A call that returns
// early (an error) synchronises the second stream in the destructor, so no kernel outlives the arena.
struct SideChain {
  giql_hip_ctx* ctx;
  hipStream_t main;
  bool active = false, joined = false;
  // n_small / n_large: rows of the side given to the second stream / of the side that stays.  The small
  // side must be small (<= overlap_max_rows): its chain is then latency-bound and costs the other side's
  // kernels little (SEMI 1M x 10M: 0.378 -> 0.349 ms; 1M x 1M: 0.387 -> 0.351 ms).
  SideChain(giql_hip_ctx* c, hipStream_t m, size_t n_small, size_t n_large, int which = 1) : ctx(c), main(m) {
    if (!c->side_stream || !(c->overlap_mask & which) || n_small == 0 || n_small > c->overlap_max_rows ||
        (n_large > c->overlap_max_rows && !c->overlap_large))
      return;
    if (sort_is_local(c, n_small)) return;  // the bucket sort's boundary / queue buffers are one per context
    if (hipEventRecord(c->ev_fork, m) != hipSuccess) return;
    if (hipStreamWaitEvent(c->side_stream, c->ev_fork, 0) != hipSuccess) return;
    active = true;
  }
  hipStream_t stream() const { return active ? ctx->side_stream : main; }
  int join() {
    if (active && !joined) {
      joined = true;
      HIP_TRY(hipEventRecord(ctx->ev_join, ctx->side_stream));
      HIP_TRY(hipStreamWaitEvent(main, ctx->ev_join, 0));
    }
    return GIQL_OK;
  }
  ~SideChain() {
    if (active && !joined) (void)hipStreamSynchronize(ctx->side_stream);
  }
};

// Belt and braces around the sort: its look-back makes progress whatever the dispatch
// order (blocks compute silent predecessors themselves, k_onesweep), so a timeout status
// is never expected; should one be reported all the same, the call is repeated ONCE in
// the ticket order (order 0) and the context stays in that order.
template <typename F>
static int with_order_fallback(giql_hip_ctx* ctx, F&& call) {
  int rc = call();
  if (rc == GIQL_STATUS_RESORT && ctx && ctx->local_sort) {
    // a bucket too large for the in-LDS stage (bucket_sort.hip.h): this table wants the four-pass sort
    ctx->local_sort = false;
    ctx->local_resorts++;
    rc = call();
  }
  if (rc == GIQL_ERR_HIP && ctx && ctx->os_order != 0 && ctx->h_meta && ctx->h_meta->status == GIQL_ERR_HIP) {
    ctx->os_order = 0;
    ctx->order_fallbacks++;
    rc = call();
  }
  return rc;
}

// Fixed-length B side of the per-row operators: the canonical length every B row has (all of
// them regular), else 0.  From the span pass's length range.
static inline i64 uniform_len_b(const DevMeta& m) {
  return (m.len_min_b == m.len_max_b && m.len_max_b > 0) ? (i64)m.len_max_b : 0;
}

// Decide the form of a per-row operator after run_spans: a context that has run before assumes
// the previous call's answer (no stream sync here) and the caller validates it with
// row_form_settled() at the read-back the call ends with anyway; the first call reads the
// lengths back.
static int row_form_guess(giql_hip_ctx* ctx, hipStream_t st, i64& uni_len, bool& speculated) {
  uni_len = 0;
  speculated = false;
  if (ctx->no_uniform) return GIQL_OK;
  if (ctx->row_spec_valid) {
    uni_len = ctx->row_spec_len;
    speculated = true;
    return GIQL_OK;
  }
  GIQL_TRY(read_meta(ctx, st));
  uni_len = uniform_len_b(*ctx->h_meta);
  ctx->last_span = ctx->h_meta->total_span;  // known before anything is sorted: the sort form follows the real density
  return GIQL_OK;
}

// After the final read-back: remember the answer; false = the call assumed a fixed length that
// B does not have (its result is wrong: repeat it; the general form is right on any input, so a
// wrong "not fixed-length" guess only costs speed).
static bool row_form_settled(giql_hip_ctx* ctx, i64 uni_len, bool speculated) {
  if (ctx->no_uniform) return true;
  const i64 actual = uniform_len_b(*ctx->h_meta);
  const bool ok = !(speculated && uni_len != 0 && actual != uni_len);
  ctx->row_spec_valid = ok;
  ctx->row_spec_len = actual;
  return ok;
}

// ================================================================= C ABI
extern "C" {

int giql_hip_abi_version(void) { return GIQL_HIP_ABI_VERSION; }

const char* giql_hip_last_error(void) { return g_err; }

int giql_hip_device_count(int* n_devices) {
  if (!n_devices) return set_err(GIQL_ERR_INVALID, "n_devices is NULL");
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  *n_devices = n;
  return GIQL_OK;
}

int giql_hip_create(int device, giql_hip_ctx** out) {
  if (!out) return set_err(GIQL_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n)
    return set_err(GIQL_ERR_INVALID, "device %d not in [0,%d)", device, n);
  HIP_TRY(hipSetDevice(device));
  giql_hip_ctx* ctx = new (std::nothrow) giql_hip_ctx();
  if (!ctx) return set_err(GIQL_ERR_NOMEM, "out of host memory");
  ctx->device = device;
  {
    const char* e = getenv("GIQL_HIP_SORT");
    ctx->classic_sort = e && strcmp(e, "classic") == 0;
    const char* v = getenv("GIQL_HIP_OS_VARIANT");
    ctx->os_variant = v ? atoi(v) : 0;
    const char* nc1 = getenv("GIQL_HIP_NO_C1_FILL");
    ctx->no_c1_fill = nc1 && atoi(nc1) != 0;
    const char* nkg = getenv("GIQL_HIP_NO_KEYGEN_GENERAL");
    ctx->no_keygen_general = nkg && atoi(nkg) != 0;
    const char* nsw = getenv("GIQL_HIP_NO_SWAP");
    ctx->no_swap = nsw && atoi(nsw) != 0;
    const char* c1i = getenv("GIQL_HIP_C1_ITEMS");
    if (c1i && (atoi(c1i) == 2 || atoi(c1i) == C1_ITEMS_MAX)) ctx->c1_items = atoi(c1i);
    const char* o = getenv("GIQL_HIP_OS_ORDER");
    if (o) ctx->os_order = atoi(o);
    const char* ha = getenv("GIQL_HIP_OS_HELP_AFTER");
    if (ha) ctx->os_help_after = (u32)strtoul(ha, nullptr, 10);
    const char* it = getenv("GIQL_HIP_INJECT_TIMEOUT");
    if (it) ctx->inject_timeout = atoi(it);
    const char* u = getenv("GIQL_HIP_NO_UNIFORM");
    ctx->no_uniform = u && atoi(u) != 0;
    const char* nh = getenv("GIQL_HIP_NO_SPAN_HIST");
    ctx->no_span_hist = nh && atoi(nh) != 0;
    const char* nsd = getenv("GIQL_HIP_NO_SKIP_DIGIT");
    ctx->no_skip_digit = nsd && atoi(nsd) != 0;
    const char* nuf = getenv("GIQL_HIP_NO_UNSTABLE_FIRST");
    ctx->no_unstable_first = nuf && atoi(nuf) != 0;
    const char* nds = getenv("GIQL_HIP_NO_DUAL_SPAN");
    ctx->no_dual_span = nds && atoi(nds) != 0;
    const char* ncb = getenv("GIQL_HIP_NO_COARSE_B");
    ctx->no_coarse_b = ncb && atoi(ncb) != 0;
    const char* cmb = getenv("GIQL_HIP_COARSE_MAX_GROUP_ROWS");
    if (cmb && atof(cmb) > 0) ctx->coarse_max_group_rows = atof(cmb);
    const char* rsd = getenv("GIQL_HIP_ROW_SKIP_DIGITS");
    if (rsd && atoi(rsd) >= 0 && atoi(rsd) <= 3) ctx->row_skip_digits = atoi(rsd);
    const char* nfc = getenv("GIQL_HIP_NO_FUSE_COUNT");
    ctx->no_fuse_count = nfc && atoi(nfc) != 0;
    const char* nbj = getenv("GIQL_HIP_NO_BUCKET_JOIN");
    ctx->no_bucket_join = nbj && atoi(nbj) != 0;
    const char* nso = getenv("GIQL_HIP_NO_SORTED_INPUT");
    ctx->no_sorted = nso && atoi(nso) != 0;
    const char* nkq = getenv("GIQL_HIP_NO_KEYGEN_Q");
    ctx->no_keygen_q = nkq && atoi(nkq) != 0;
    const char* qsd = getenv("GIQL_HIP_Q_SKIP_DIGITS");
    if (qsd && atoi(qsd) >= 1 && atoi(qsd) <= 2) ctx->fuse_q_skip = atoi(qsd);
    const char* nl = getenv("GIQL_HIP_NO_LOCAL_SORT");
    if (nl && atoi(nl) != 0) ctx->local_sort = false;
    const char* lm = getenv("GIQL_HIP_LOCAL_MIN_ROWS");
    if (lm) {
      ctx->local_min_rows = strtoull(lm, nullptr, 10);
      ctx->local_max_bucket_rows = 1e30;  // a forced size threshold (tests, sweeps) is not second-guessed by density
      ctx->local_min_bucket_rows = 0.0;
    }
    const char* lnb = getenv("GIQL_HIP_LOCAL_MIN_BUCKET_ROWS");
    if (lnb && atof(lnb) >= 0) ctx->local_min_bucket_rows = atof(lnb);
    const char* lmb = getenv("GIQL_HIP_LOCAL_MAX_BUCKET_ROWS");
    if (lmb && atof(lmb) > 0) ctx->local_max_bucket_rows = atof(lmb);
    const char* nnb = getenv("GIQL_HIP_NO_NARROW_BUCKETS");
    ctx->no_narrow = nnb && atoi(nnb) != 0;
    const char* lbw = getenv("GIQL_HIP_LOCAL_BITS");
    if (lbw && atoi(lbw) >= BS_MIN_WBITS && atoi(lbw) <= 16) ctx->force_bits = atoi(lbw);
  }
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
      ctx->n_cu = prop.multiProcessorCount;
  }
  hipError_t e = hipMalloc((void**)&ctx->d_meta, sizeof(DevMeta));
  if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_meta, sizeof(DevMeta), hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_scratch64, 64);
  {
    const char* om = getenv("GIQL_HIP_OVERLAP_MASK");
    if (om) ctx->overlap_mask = atoi(om);
    const char* no = getenv("GIQL_HIP_NO_OVERLAP");
    if (!(no && atoi(no) != 0) && e == hipSuccess) {
      const char* omr = getenv("GIQL_HIP_OVERLAP_MAX_ROWS");
      if (omr && atoll(omr) > 0) ctx->overlap_max_rows = (u64)atoll(omr);
      const char* ol = getenv("GIQL_HIP_OVERLAP_LARGE");
      if (ol) ctx->overlap_large = atoi(ol) != 0;
      if (hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess)
        ctx->side_stream = nullptr;  // no second stream: everything stays on the caller's
    }
  }
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->bucket_bnd, ((size_t)BS_MAX_BUCKETS + 16) * sizeof(u32));
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->bucket_big, ((size_t)BS_MAX_BUCKETS + 16) * sizeof(u32));
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->bucket_qwin, ((size_t)2 * BS_MAX_BUCKETS + 16) * sizeof(u32));
  if (e != hipSuccess) {
    giql_hip_destroy(ctx);
    return set_err(GIQL_ERR_HIP, "context allocation failed: %s", hipGetErrorString(e));
  }
  *out = ctx;
  return GIQL_OK;
}

int giql_hip_destroy(giql_hip_ctx* ctx) {
  if (!ctx) return GIQL_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->arena) (void)hipFree(ctx->arena);
  if (ctx->part) (void)hipFree(ctx->part);
  if (ctx->stage_out) (void)hipFree(ctx->stage_out);
  if (ctx->d_meta) (void)hipFree(ctx->d_meta);
  if (ctx->d_scratch64) (void)hipFree(ctx->d_scratch64);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  if (ctx->bucket_bnd) (void)hipFree(ctx->bucket_bnd);
  if (ctx->bucket_big) (void)hipFree(ctx->bucket_big);
  if (ctx->bucket_qwin) (void)hipFree(ctx->bucket_qwin);
  if (ctx->xplan) (void)hipFree(ctx->xplan);
  if (ctx->h_meta) (void)hipHostFree(ctx->h_meta);
  delete ctx;
  return GIQL_OK;
}

int giql_hip_reserve(giql_hip_ctx* ctx, int64_t bytes) {
  if (!ctx || bytes < 0) return set_err(GIQL_ERR_INVALID, "bad ctx/bytes");
  HIP_TRY(hipSetDevice(ctx->device));
  return ensure_arena(ctx, (size_t)bytes, nullptr);
}

int giql_hip_set_profiling(giql_hip_ctx* ctx, int enabled) {
  if (!ctx) return set_err(GIQL_ERR_INVALID, "ctx is NULL");
  if (enabled >= 16 && enabled < 16 + GIQL_PH_N) {  // events around ONE phase only: 16 + its GIQL_PH_* number
    ctx->profiling = 2;
    ctx->profile_phase = enabled - 16;
    return GIQL_OK;
  }
  ctx->profiling = enabled < 0 ? 0 : (enabled > 2 ? 1 : enabled);
  ctx->profile_phase = GIQL_PH_SORT_SCATTER;
  return GIQL_OK;
}

int giql_hip_get_stats(giql_hip_ctx* ctx, giql_hip_stats* out) {
  if (!ctx || !out) return set_err(GIQL_ERR_INVALID, "ctx/out is NULL");
  if (ctx->profiling && !ctx->spans.empty()) {
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(ctx->spans.back().b));
    collect_spans(ctx);
  }
  ctx->stats.workspace_bytes = (int64_t)ctx->arena_cap;
  *out = ctx->stats;
  // byte 0: join form (+ bit 5: a side was sorted in three stages, bit 6: this context fell back to
  // the four-pass sort for good); byte 1: sort tile order in force; bits 16-26: order fallbacks so far; bits 27-28:
  // 16 - the key bits of the last bucket stage's buckets (0: 65,536-key buckets)
  out->reserved = (ctx->stats.reserved & 0x1F) | (ctx->last_sort_local ? 0x20 : 0) |
                  (ctx->local_resorts ? 0x40 : 0) | (ctx->swapped ? 0x80 : 0) | ((ctx->os_order & 0x7F) << 8) |
                  (ctx->count_fused ? 0x8000 : 0) | ((ctx->used_sorted[0] || ctx->used_sorted[1]) ? (int32_t)0x80000000u : 0) |
                  ((ctx->order_fallbacks & 0x7FF) << 16) | (((16 - ctx->last_local_bits) & 3) << 27) |
                  (ctx->bucket_join ? (1 << 29) : 0) |
                  (ctx->fuse_done ? (1 << 30) : 0);
  return GIQL_OK;
}

// ------------------------------------------------------------------ INNER
static int inner_plan_core(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b,
                           int32_t n_chrom, void* stream, int64_t* n_pairs) {
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  ctx->planned = false;
  ctx->fuse_done = false;
  reset_stats(ctx);
  ctx->stats.n_a = a->n;
  ctx->stats.n_b = b->n;
  ctx->side_a = *a;
  ctx->side_b = *b;
  ctx->n_a = (u32)a->n;
  ctx->n_b = (u32)b->n;
  ctx->n_chrom = n_chrom;
  ctx->n_reg = ctx->n_irr = ctx->n_c1 = 0;
  *n_pairs = 0;
  if (a->n == 0 || b->n == 0 || n_chrom == 0) {  // empty result (tests :4173-4229)
    ctx->planned = true;
    ctx->plan_is_join = false;
    return GIQL_OK;
  }
  const size_t na = (size_t)a->n, nb = (size_t)b->n, nq = na + nb;

  // ---- carve the arena (dry run for the size, then for real)
  LinBufs lb;
  InnerState& S = ctx->inner;
  u32 *tile_hist = nullptr, *cnt2 = nullptr, *irr_cnt = nullptr;
  u32 *hist_a = nullptr, *hist_b = nullptr, *gbase_a = nullptr, *gbase_b = nullptr;
  u32 *os_status = nullptr, *os_status2 = nullptr, *top_partial_q = nullptr;
  u64 *bsums = nullptr, *bsums1 = nullptr, *scan_chain = nullptr;
  size_t zero_off = 0, zero_end = 0;  // arena offsets of the region zeroed up front ([zero_off, zero_end + the larger side's status words))
  const bool onesweep = !ctx->classic_sort && na <= OS_MAX_ROWS && nb <= OS_MAX_ROWS;
  const size_t n_max = na > nb ? na : nb;
  const size_t n_tiles_max = cdiv(n_max, RS_TILE);
  const size_t scan_max = (nq > n_tiles_max * RS_BINS ? nq : n_tiles_max * RS_BINS);
  constexpr u32 TQ2 = RC_NT * RC_ITEMS_C2;
  S.c1_items = ctx->c1_items ? ctx->c1_items : (nb <= ctx->c1_small_rows ? 2 : C1_ITEMS_MAX);
  const u32 c1_tq = (u32)(C1_NT * S.c1_items);  // class-1 rows per block
  S.nt1 = cdiv(nb, c1_tq);
  S.c1_fill = !ctx->no_c1_fill && nb <= ctx->c1_small_rows;
  S.nt2 = cdiv(na, TQ2);
  auto carve = [&](char* base) {
    Carver c{base};
    common_sizes(c, n_chrom, lb);
    sort_sizes(c, na, S.sa, true);
    sort_sizes(c, nb, S.sb, true);
    if (onesweep) {
      // everything that has to start at zero lies back to back, the larger side's status words last: ONE memset
      // up front instead of one per histogram and per sort (5 launches of ~5 us at the headline sizes)
      c.off = align_up(c.off, 256);
      zero_off = c.off;
      hist_a = c.take<u32>((size_t)LIN_HIST_REPLICAS * 1024);
      hist_b = c.take<u32>((size_t)LIN_HIST_REPLICAS * 1024);
      lb.top_partial = c.take<u32>((size_t)LIN_HIST_REPLICAS * MM_TOP_WORDS);
      top_partial_q = c.take<u32>((size_t)LIN_HIST_REPLICAS * MM_TOP_WORDS);
      gbase_a = c.take<u32>(1024);
      gbase_b = c.take<u32>(1024);
      lb.abase = c.take<u32>(MM_HIST_CHROMS);
      os_status2 = c.take<u32>(4 * os_pass_words(na < nb ? na : nb));  // the smaller side's
      c.off = align_up(c.off, 256);
      zero_end = c.off;
      os_status = c.take<u32>(4 * os_pass_words(n_max));               // the larger side's
    } else {
      tile_hist = c.take<u32>(n_tiles_max * RS_BINS);
    }
    bsums = c.take<u64>(cdiv(scan_max, SCAN_TILE) + 2);
    S.wlo1 = c.take<u32>((size_t)S.nt1 + 2 + (S.c1_fill ? cdiv(nb, TQ2) : 0));
    S.c1_base = c.take<u64>((size_t)S.nt1 + 2);
    const size_t n1 = S.c1_fill ? nb : 0;
    S.lo1 = c.take<u32>(n1);
    S.cnt1 = c.take<u32>(n1);
    S.off1 = c.take<u64>(n1 + 1);
    bsums1 = c.take<u64>(cdiv(n1 ? n1 : 1, SCAN_TILE) + 2);
    const size_t nq2 = ctx->no_uniform ? na : n_max;  // the uniform form may query the other side
    S.wlo2 = c.take<u32>((size_t)cdiv(nq2, TQ2) + 2);
    cnt2 = c.take<u32>(nq2);
    S.cnt2 = cnt2;
    S.lo2 = c.take<u32>(nq2);
    S.off2 = c.take<u64>(nq2 + 1);
    scan_chain = c.take<u64>((size_t)cdiv(nq2, SCAN_TILE) + 4);  // chained scan: a status word per tile + the ticket
    ctx->irr_a_list = c.take<u32>(na);
    ctx->irr_b_list = c.take<u32>(nb);
    irr_cnt = c.take<u32>(nq);
    ctx->irr_off = c.take<u64>(nq + 1);
    return c.off;
  };
  const size_t need = carve(nullptr);
  GIQL_TRY(ensure_arena(ctx, need, st));
  carve(ctx->arena);
  SortBufs& sa = S.sa;
  SortBufs& sbb = S.sb;
  struct Prezero {  // the helpers skip their own memsets while this plan runs
    giql_hip_ctx* c;
    ~Prezero() {
      c->prezeroed = false;
      c->first_unstable = false;
    }
  } prezero_guard{ctx};
  ctx->first_unstable = true;  // an INNER join needs no order among rows of equal keys: first passes rank by LDS atomics
  ctx->prezeroed = false;
  ctx->span_hist_dirty[0] = ctx->span_hist_dirty[1] = nullptr;
  int big_passes_zeroed = 0;
  if (onesweep) {
    // the larger side takes at most 4 passes (2 in the three-stage form)
    big_passes_zeroed = sort_is_local(ctx, n_max) ? local_passes(sort_local_bits(ctx, n_max)) : 4;
    const size_t big_words = (size_t)big_passes_zeroed * os_pass_stride(ctx, n_max);
    HIP_TRY(hipMemsetAsync(ctx->arena + zero_off, 0, (zero_end - zero_off) + big_words * sizeof(u32), st));
    ctx->prezeroed = true;
  }

  // The larger side is the one the uniform form prefers as its fixed-length side: its span pass
  // also counts the digits of its keys (aligned layout), so that, if the form and the layout
  // hold, it is sorted straight from its raw columns with no linearize pass.  A context that
  // has planned before asks for it only when its last plan ended that way.
  const int big_side = nb >= na ? 1 : 0;
  bool want_hist = onesweep && !ctx->no_span_hist && ctx->os_variant == 0 &&
                   n_chrom <= MM_HIST_CHROMS;
  if (want_hist && ctx->spec_valid)  // ... or in the general form without irregular rows (its larger side)
    want_hist = ctx->spec_aligned && (ctx->spec_form == (big_side ? 1 : 2) ||
                                      (ctx->spec_form == 0 && ctx->last_no_irr && !ctx->no_keygen_general));
  // ... and the QUERY side of the fixed-length form too (round 3): its (key, end, rid) sort starts from the raw
  // columns as well (k_onesweep<3, .., KEYGEN>), so neither side has a linearize pass -- on the guesses that the
  // form and the layout hold AND that the query side has no irregular row (those carry the sentinel key and a
  // list entry, which only the linearize pass produces): validated at the read-back like the others.
  const bool want_hist_q = want_hist && ctx->spec_valid && ctx->spec_form == (big_side ? 1 : 2) && ctx->last_no_irr &&
                           !ctx->no_keygen_q && (big_side ? na : nb) > 0;
  if (want_hist_q) {
    lb.hist_partial2 = big_side ? hist_a : hist_b;
    lb.top_partial2 = top_partial_q;
  }
  GIQL_TRY(run_spans(ctx, st, *a, *b, n_chrom, lb, want_hist ? big_side : -1, big_side ? hist_b : hist_a));
  bool aligned = false;
  // Uniform-length side?  (fixed-length reads: min == max canonical length > 0 over ALL its
  // rows; a single irregular row makes the minimum 0).
  // One 100-byte readback; it also surfaces chrom / span errors before the sort.
  S.uniform = 0;
  i64 uni_len = 0;
  // The form is decided from the length ranges the min/max pass just produced.  Reading
  // them back costs a stream sync in the middle of the plan, so a context that has
  // planned before SPECULATES on its previous decision and validates it against the same
  // numbers at the read-back the plan ends with anyway; a wrong guess (the kernels are
  // memory-safe on any input) repeats the plan once without speculation.
  auto decide = [&](const DevMeta& m, int& form, i64& len) {
    const bool ub = m.len_min_b == m.len_max_b && m.len_max_b > 0;
    const bool ua = m.len_min_a == m.len_max_a && m.len_max_a > 0;
    form = 0;
    len = 0;
    if (ctx->no_uniform) return;  // GIQL_HIP_NO_UNIFORM: the general form whatever the lengths
    // sort the uniform side without its end; prefer the larger side when both are
    if (ub && (!ua || nb >= na)) {
      form = 1;
      len = m.len_max_b;
    } else if (ua) {
      form = 2;
      len = m.len_max_a;
    }
  };
  bool speculated = false;
  bool coarse_q = false;  // the query side was sorted without its lowest digit (a guess: no irregular rows)
  bool keygen_q = false;  // ... and from its raw columns (the same guess)
  if (onesweep) {
    if (ctx->spec_valid) {
      S.uniform = ctx->spec_form;
      uni_len = ctx->spec_len;
      aligned = want_hist;  // = ctx->spec_aligned when the form matches
      speculated = true;
    } else {
      GIQL_TRY(read_meta(ctx, st));
      decide(*ctx->h_meta, S.uniform, uni_len);
      aligned = want_hist && ctx->h_meta->aligned_ok != 0;
      // the span is known now, before anything is sorted: the sort form follows the table's real density
      // already on this first plan.  The span pass counted its digits for the form it expected: if the
      // larger side leaves the three-stage sort, its high-digits-only histogram is of no use and that side
      // is linearized (which counts all four digits)
      const size_t n_big = big_side ? nb : na;
      const int expected_bits = sort_local_bits(ctx, n_big);
      ctx->last_span = ctx->h_meta->total_span;
      // (the high-digits-only histogram serves 16-bit buckets alone: narrower ones sort on bits 8-15 too)
      if (expected_bits == 16 && sort_local_bits(ctx, n_big) != 16) aligned = false;
      const int passes_now = sort_is_local(ctx, n_max) ? local_passes(sort_local_bits(ctx, n_max)) : 4;
      if (big_passes_zeroed < passes_now) {
        // the larger side takes more global passes than expected: those need zeroed status words too
        const size_t stride = os_pass_stride(ctx, n_max);
        HIP_TRY(hipMemsetAsync(os_status + (size_t)big_passes_zeroed * stride, 0,
                               (size_t)(passes_now - big_passes_zeroed) * stride * sizeof(u32), st));
        big_passes_zeroed = passes_now;
      }
    }
  }
  // Sorted inputs: a side the span pass found in (chrom id, start) order (and free of irregular rows) skips its
  // sort -- from the read-back on a first plan, the previous plan's answer afterwards (validated below)
  bool pre_a = false, pre_b = false;
  if (onesweep && !ctx->no_sorted) {
    pre_a = speculated ? ctx->spec_sorted[0] : ctx->h_meta->unsorted_a == 0;
    pre_b = speculated ? ctx->spec_sorted[1] : ctx->h_meta->unsorted_b == 0;
  }
  ctx->used_sorted[0] = pre_a;
  ctx->used_sorted[1] = pre_b;
  const bool keygen = aligned && S.uniform == (big_side ? 1 : 2);
  // General form: the larger side's (key, end, rid) sort can start from the raw columns too, as long as
  // that side holds no irregular row (those carry the sentinel key, which depends on `end`; the span pass
  // counted digits of start alone).  Known from the read-back on a first plan, a guess afterwards.
  const bool keygen_g = aligned && S.uniform == 0 && !ctx->no_keygen_general &&
                        (speculated ? ctx->last_no_irr
                                    : (big_side ? ctx->h_meta->len_min_b : ctx->h_meta->len_min_a) > 0);
  const u32* irr_a = &ctx->d_meta->irr_a;
  const u32* irr_b = &ctx->d_meta->irr_b;
  if (S.uniform) {
    // ---- uniform-length form: queries Q (full rows) against points U (starts only)
    const bool q_is_a = S.uniform == 1;
    const giql_side& qs_ = q_is_a ? *a : *b;
    const giql_side& us_ = q_is_a ? *b : *a;
    SortBufs& sq = q_is_a ? sa : sbb;
    SortBufs& su = q_is_a ? sbb : sa;
    const size_t nqr = q_is_a ? na : nb, nu = q_is_a ? nb : na;
    su.end[0] = su.end[1] = nullptr;  // the uniform side carries (key, rid) only
    // Fused range count: when U takes the three-stage sort, its bucket sort answers the queries' bounds
    // (bucket_sort.hip.h) -- as long as no query row is longer than the windows allow for: known from the
    // read-back on a first plan, the previous plan's answer afterwards (validated below like the other guesses)
    const int q_len_max = q_is_a ? ctx->h_meta->len_max_a : ctx->h_meta->len_max_b;
    const bool fuse_cnt = !ctx->no_fuse_count && sort_is_local(ctx, nu) &&
                          (speculated ? ctx->spec_fuse_len_ok : q_len_max <= (int)BS_FUSE_WCAP);
    // the query side's chain (linearize + sort) beside the other side's when it is small (the fused count
    // needs the sorted queries before U's last stage: one stream)
    // the query side sorted from its raw columns too (its digits were counted in the span pass)
    keygen_q = keygen && want_hist_q;
    u32* const hist_q = q_is_a ? hist_a : hist_b;
    if (keygen) {
      // the span pass counted U's digits already: fold the per-chromosome top digits onto the
      // bases, scan, and let the first sort pass build the keys from (chrom, start)
      Phase ph(ctx, st, GIQL_PH_LINEARIZE, 2);
      u32* hist_u = q_is_a ? hist_b : hist_a;
      if (keygen_q) {  // both sides, one launch each
        hipLaunchKernelGGL(k_fold_top2, dim3(MM_HIST_CHROMS, 2), dim3(256), 0, st, lb.top_partial, lb.top_partial2,
                           lb.abase, hist_u, hist_q);
        hipLaunchKernelGGL(k_digit_offsets2, dim3(4, 2), dim3(256), 0, st, hist_u, hist_q, (u32)LIN_HIST_REPLICAS,
                           q_is_a ? gbase_b : gbase_a, q_is_a ? gbase_a : gbase_b);
      } else {
        hipLaunchKernelGGL(k_fold_top, dim3(MM_HIST_CHROMS), dim3(256), 0, st, lb.top_partial, lb.abase, hist_u);
        hipLaunchKernelGGL(k_digit_offsets, dim3(4), dim3(256), 0, st, hist_u, (u32)LIN_HIST_REPLICAS,
                           q_is_a ? gbase_b : gbase_a);
      }
      GIQL_TRY(post_launch("digit offsets (span histogram)"));
    }
    // (the fork comes AFTER the digit offsets above: the query side's sort on the second stream reads them)
    SideChain sc(ctx, st, (nqr <= nu && !fuse_cnt) ? nqr : 0, nu);
    if (!keygen_q) {
      GIQL_TRY(run_linearize(ctx, sc.stream(), qs_, n_chrom, lb, sq.key[0], sq.end[0],
                             q_is_a ? ctx->irr_a_list : ctx->irr_b_list, q_is_a ? 0 : 1, 0,
                             hist_q, q_is_a ? gbase_a : gbase_b));
    }
    if (!keygen) {
      GIQL_TRY(run_linearize(ctx, st, us_, n_chrom, lb, su.key[0], nullptr,
                             q_is_a ? ctx->irr_b_list : ctx->irr_a_list, q_is_a ? 1 : 0, 0,
                             q_is_a ? hist_b : hist_a, q_is_a ? gbase_b : gbase_a, nullptr, nullptr,
                             /*skip_end=*/true));  // uniform => no irregular row: `end` is not read
    }
    // The query side's order only serves locality (its bounds are searched per row, its pairs are laid
    // out by the scan), so its lowest digit stays unsorted: three passes.  Irregular rows would break
    // that (their sentinel keys must end up LAST, after every row of the top 256-key block): taken
    // only on the context's guess that there are none, validated with the other guesses below.
    // With the fused count the queries only have to be grouped by the bucket their key falls into (the windows
    // of k_bucket_bounds_fused are computed under the same mask and then cover whole query buckets): TWO digits
    // unsorted, two passes (GIQL_HIP_Q_SKIP_DIGITS=1 keeps three).
    int q_skip = (speculated && ctx->last_no_irr && !ctx->no_skip_digit && !sort_is_local(ctx, nqr)) ? 1 : 0;
    // (narrower buckets: the queries stay grouped by key >> 8 -- a window under the 16-bit mask would span 2-8 buckets)
    const int wb_u = sort_local_bits(ctx, nu);
    if (q_skip && fuse_cnt && ctx->fuse_q_skip > 1 && wb_u == 16) q_skip = ctx->fuse_q_skip;
    coarse_q = q_skip != 0;
    const u32 q_mask = q_skip == 2 ? 0xFFFF0000u : (q_skip == 1 ? 0xFFFFFF00u : 0xFFFFFFFFu);
    // (the smaller side's passes use the smaller status buffer: both were zeroed up front, neither is reused)
    u32* const stat_q = nqr <= nu ? os_status2 : os_status;
    u32* const stat_u = nqr <= nu ? os_status : os_status2;
    GIQL_TRY(run_sort_onesweep(ctx, sc.stream(), sq, (u32)nqr, q_is_a ? gbase_a : gbase_b,
                               stat_q, false, keygen_q ? &qs_ : nullptr, lb.abase, q_skip, nullptr,
                               q_is_a ? pre_a : pre_b));
    // One-call join (giql_hip_inner_join_dev): the caller's buffers are here and everything about this plan is a
    // guess that has held so far (same form as last time, no irregular rows), so the fill is launched inside the
    // plan, with a grid bounded by the capacity and the true count read on the device.
    constexpr u32 T2 = FILL_NT * FILL_ITEMS_C2;
    static_assert((T2 & (T2 - 1)) == 0, "fill tiles are a power of two (k_scan_chain_diff shifts)");
    const u64 nt_cap64 = (ctx->fuse_cap + T2 - 1) / T2;
    const bool early_fill = ctx->fuse_a && speculated && ctx->last_no_irr && ctx->fuse_cap > 0 &&
                            nt_cap64 <= 0x7FFFFFF0ull && (size_t)nt_cap64 + 2 <= ctx->part_cap;
    bool part_done = false;  // the scan wrote the fill's partition
    // ... and, when the other side's bucket stage answers the bounds anyway, the pairs are written right there
    // (bucket_sort.hip.h, FUSE == 2) -- as long as the query windows stay well inside what a block holds in
    // registers: a window covers the query buckets (coarsely grouped: whole 65536-key buckets) that a bucket's
    // reach -- its own 65536 keys, the fixed length above it, the longest query below it -- touches
    const double win_keys = q_skip == 2 ? 3.0 * 65536.0 + (double)uni_len
                                        : (double)(1u << (wb_u ? wb_u : 16)) + 512.0 + (double)uni_len +
                                              (double)(q_len_max > 0 ? q_len_max : 0);
    const bool join_in_buckets = fuse_cnt && !ctx->no_bucket_join && ctx->fuse_a && speculated && ctx->last_no_irr &&
                                 ctx->fuse_cap > 0 && ctx->last_span > 0 &&
                                 (double)nqr * win_keys / (double)ctx->last_span <= 0.75 * (double)BJ_WCAP;
    FuseCount fc;
    if (fuse_cnt) {
      if (join_in_buckets) {
        fc.join = true;
        fc.dev.qrid = sq.rid[0];
        fc.dev.row_q = q_is_a ? ctx->fuse_a : ctx->fuse_b;
        fc.dev.row_s = q_is_a ? ctx->fuse_b : ctx->fuse_a;
        fc.dev.cap = ctx->fuse_cap;
        fc.dev.cursor = reinterpret_cast<unsigned long long*>(&ctx->d_meta->n_out);  // zeroed by the span pass
      }
      fc.zero_ptr = reinterpret_cast<u32*>(scan_chain);
      fc.zero_words = (u32)(2 * (cdiv(nqr, SC_TILE) + 2));
      fc.dev.qkey = sq.key[0];
      fc.dev.qend = sq.end[0];
      fc.dev.qwin = ctx->bucket_qwin;
      fc.dev.lo_out = S.lo2;
      fc.dev.hi_out = cnt2;  // the scan below turns the upper bounds into counts in place
      fc.dev.lo_off = 1 - uni_len;  // u.start in [q.start - L + 1, q.end)
      fc.nq_total = (u32)nqr;
      fc.irr_q = q_is_a ? irr_a : irr_b;
      fc.gbq3 = (q_is_a ? gbase_a : gbase_b) + 3 * OS_BINS;
      fc.key_mask = q_mask;
      fc.len_max_q = q_is_a ? &ctx->d_meta->len_max_a : &ctx->d_meta->len_max_b;
    }
    GIQL_TRY(run_sort_onesweep(ctx, st, su, (u32)nu, q_is_a ? gbase_b : gbase_a, stat_u, false,
                               keygen ? &us_ : nullptr, lb.abase, 0, fuse_cnt ? &fc : nullptr,
                               q_is_a ? pre_b : pre_a));
    GIQL_TRY(sc.join());
    constexpr u32 TQ = RC_NT * RC_ITEMS_C2;
    S.nt2 = cdiv(nqr, TQ);
    if (ctx->bucket_join) {
      // the pairs are out already, counted in DevMeta::n_out
    } else if (ctx->count_fused) {
      u32 log2_t2 = 0;
      while ((1u << log2_t2) < T2) log2_t2++;
      GIQL_TRY(run_scan_chain(ctx, st, GIQL_PH_SCAN, cnt2, S.lo2, nqr, q_is_a ? irr_a : irr_b, S.off2, scan_chain,
                              &ctx->d_meta->n_out, early_fill ? ctx->part : nullptr, log2_t2, (u32)nt_cap64 + 1));
      part_done = early_fill;
    } else {
    {
      Phase ph(ctx, st, GIQL_PH_COUNT, 2);
      const u32* irr_q = q_is_a ? irr_a : irr_b;
      const u32* irr_u = q_is_a ? irr_b : irr_a;
      const i64 lo_off = 1 - uni_len;  // u.start in [q.start - L + 1, q.end)
      hipLaunchKernelGGL(k_count_partition, dim3(cdiv((u64)S.nt2 + 1, 256)), dim3(256), 0, st,
                         sq.key[0], (u32)nqr, irr_q, su.key[0], (u32)nu, irr_u, lo_off, TQ, S.nt2,
                         S.wlo2, q_mask);
      hipLaunchKernelGGL((k_range_count<RC_ITEMS_C2, RC_LDS_CAP>), dim3(S.nt2), dim3(RC_NT), 0, st,
                         sq.key[0], sq.end[0], (u32)nqr, irr_q, su.key[0], (u32)nu, irr_u, lo_off,
                         S.wlo2, S.lo2, cnt2);
      GIQL_TRY(post_launch("range count (uniform)"));
    }
    GIQL_TRY(run_scan<u64>(ctx, st, GIQL_PH_SCAN, cnt2, nqr, S.off2, bsums, S.off2 + nqr, &ctx->d_meta->n_out));
    }
    bool fused = false;
    {
      // The early fill: no stream sync between plan and fill.  Validated below; a wrong guess leaves the
      // buffers to the ordinary fill.
      if (ctx->bucket_join) {
        fused = true;
      } else if (early_fill) {
        const u32 nt_cap = (u32)nt_cap64;
        const u32* qrid = q_is_a ? sa.rid[0] : sbb.rid[0];
        const u32* srid = q_is_a ? sbb.rid[0] : sa.rid[0];
        int32_t* rq = q_is_a ? ctx->fuse_a : ctx->fuse_b;
        int32_t* rs = q_is_a ? ctx->fuse_b : ctx->fuse_a;
        if (!part_done) {
          Phase ph(ctx, st, GIQL_PH_PARTITION);
          hipLaunchKernelGGL(k_partition, dim3(cdiv((u64)nt_cap + 1, 256)), dim3(256), 0, st, S.off2,
                             (u32)nqr, (u64)0, T2, nt_cap, ctx->part, (const u64*)(S.off2 + nqr), ctx->fuse_cap);
        }
        {
          Phase ph(ctx, st, GIQL_PH_FILL);
          hipLaunchKernelGGL((k_fill<FILL_ITEMS_C2>), dim3(nt_cap), dim3(FILL_NT), 0, st, S.off2, S.lo2, qrid,
                             (u32)nqr, srid, ctx->part, (u64)0, (u64)0, rq, rs, (const u64*)(S.off2 + nqr), ctx->fuse_cap);
        }
        GIQL_TRY(post_launch("fused fill"));
        fused = true;
      }
    }
    GIQL_TRY(read_meta(ctx, st));
    ctx->n_c1 = 0;
    ctx->n_reg = ctx->h_meta->n_out;
    // the early fill stands only if every guess held and the pairs fitted
    ctx->fuse_done = fused && ctx->h_meta->irr_a + ctx->h_meta->irr_b == 0 && ctx->n_reg <= ctx->fuse_cap;
    if (ctx->bucket_join) ctx->stats.phase_bytes[GIQL_PH_SORT_LOCAL] += (int64_t)8 * (int64_t)ctx->n_reg;  // the pairs
  } else {
  // The join itself in B's bucket stage (bucket_sort.hip.h, FUSE == 3): one-call form, on the context's guesses (the
  // general form again, no irregular row, no row longer than the windows allow for), B the three-stage side, the A
  // rows -- fully sorted -- sparse enough for a bucket's window to stay in a block's registers.
  const bool general_join = onesweep && !ctx->no_bucket_join && ctx->fuse_a && speculated && ctx->last_no_irr &&
                            ctx->fuse_cap > 0 && nb >= na && sort_is_local(ctx, nb) && ctx->spec_fuse_len_ok &&
                            ctx->last_span > 0 &&
                            // (a window = the A rows within the bucket's 65536 keys + the longest rows of either side: the
                            // previous plan's maxima, like the guess they validate)
                            (double)na * ((double)(1u << (sort_local_bits(ctx, nb) ? sort_local_bits(ctx, nb) : 16)) +
                                          (double)ctx->h_meta->len_max_a + (double)ctx->h_meta->len_max_b) /
                                    (double)ctx->last_span <= 0.5 * (double)BJ_WCAP;
  // the smaller side's chain (linearize + sort) beside the larger side's when it is small (not in the form above:
  // B's last stage reads the sorted A)
  SideChain sc(ctx, st, (onesweep && !general_join) ? (na < nb ? na : nb) : 0, na < nb ? nb : na);
  const bool a_small = na < nb;
  hipStream_t st_a = a_small ? sc.stream() : st, st_b = a_small ? st : sc.stream();
  const bool kg_a = keygen_g && !big_side, kg_b = keygen_g && big_side;
  for (int k = 0; k < 2; k++) {
    const bool is_b = k == 1;
    hipStream_t stk = is_b ? st_b : st_a;
    if (is_b ? kg_b : kg_a) {
      // the span pass counted this side's digits: fold the top ones onto the bases and scan (no linearize pass)
      Phase ph(ctx, stk, GIQL_PH_LINEARIZE, 2);
      u32* hist_k = is_b ? hist_b : hist_a;
      hipLaunchKernelGGL(k_fold_top, dim3(MM_HIST_CHROMS), dim3(256), 0, stk, lb.top_partial, lb.abase, hist_k);
      hipLaunchKernelGGL(k_digit_offsets, dim3(4), dim3(256), 0, stk, hist_k, (u32)LIN_HIST_REPLICAS,
                         is_b ? gbase_b : gbase_a);
      GIQL_TRY(post_launch("digit offsets (span histogram, general form)"));
    } else if (is_b) {
      GIQL_TRY(run_linearize(ctx, st_b, *b, n_chrom, lb, sbb.key[0], sbb.end[0], ctx->irr_b_list, 1, 0,
                             hist_b, gbase_b));
    } else {
      GIQL_TRY(run_linearize(ctx, st_a, *a, n_chrom, lb, sa.key[0], sa.end[0], ctx->irr_a_list, 0, 0,
                             hist_a, gbase_a));
    }
  }
  if (onesweep) {
    GIQL_TRY(run_sort_onesweep(ctx, st_a, sa, (u32)na, gbase_a, a_small ? os_status2 : os_status,
                               false, kg_a ? a : nullptr, lb.abase, 0, nullptr, pre_a));
    FuseCount fg;
    if (general_join) {
      fg.join = fg.general = true;
      fg.zero_ptr = reinterpret_cast<u32*>(scan_chain);
      fg.zero_words = 0;
      fg.dev.qkey = sa.key[0];
      fg.dev.qend = sa.end[0];
      fg.dev.qrid = sa.rid[0];
      fg.dev.qwin = ctx->bucket_qwin;
      fg.dev.lo_out = fg.dev.hi_out = nullptr;
      fg.dev.lo_off = 1;  // class 2: b.start in (a.start, a.end)
      fg.dev.row_q = ctx->fuse_a;
      fg.dev.row_s = ctx->fuse_b;
      fg.dev.cap = ctx->fuse_cap;
      fg.dev.cursor = reinterpret_cast<unsigned long long*>(&ctx->d_meta->n_out);  // zeroed by the span pass
      fg.nq_total = (u32)na;
      fg.irr_q = irr_a;
      fg.gbq3 = gbase_a + 3 * OS_BINS;
      fg.key_mask = 0xFFFFFFFFu;
      fg.len_max_q = &ctx->d_meta->len_max_a;
      fg.len_max_u = &ctx->d_meta->len_max_b;
    }
    GIQL_TRY(run_sort_onesweep(ctx, st_b, sbb, (u32)nb, gbase_b, a_small ? os_status : os_status2,
                               false, kg_b ? b : nullptr, lb.abase, 0, general_join ? &fg : nullptr, pre_b));
    GIQL_TRY(sc.join());
  } else {
    GIQL_TRY(run_sort(ctx, st, sa, (u32)na, tile_hist, bsums));
    GIQL_TRY(run_sort(ctx, st, sbb, (u32)nb, tile_hist, bsums));
  }
  if (ctx->bucket_join) {
    // the pairs are out already, counted in DevMeta::n_out
    GIQL_TRY(read_meta(ctx, st));
    ctx->n_c1 = 0;
    ctx->n_reg = ctx->h_meta->n_out;
    ctx->fuse_done = ctx->h_meta->irr_a + ctx->h_meta->irr_b == 0 && ctx->n_reg <= ctx->fuse_cap;
    ctx->stats.phase_bytes[GIQL_PH_SORT_LOCAL] += (int64_t)8 * (int64_t)ctx->n_reg;  // the pairs
  } else {
  // class 1 (count + the scan of its block totals) runs beside class 2 when both sides are small
  SideChain sc1(ctx, st, onesweep ? (na < nb ? na : nb) : 0, na < nb ? nb : na, 2);
  hipStream_t st1 = sc1.stream();
  {
    Phase ph(ctx, st, GIQL_PH_COUNT, 4);
    // class 1: queries = sorted B, points = sorted A starts, range [b.start, b.end);
    // only one total per block is kept (see k_c1_count)
    if (S.c1_fill) {
      const u32 nt1r = cdiv(nb, TQ2);
      hipLaunchKernelGGL(k_count_partition, dim3(cdiv((u64)nt1r + 1, 256)), dim3(256), 0, st1,
                         sbb.key[0], (u32)nb, irr_b, sa.key[0], (u32)na, irr_a, (i64)0, TQ2, nt1r, S.wlo1);
      hipLaunchKernelGGL((k_range_count<RC_ITEMS_C2, RC_LDS_CAP>), dim3(nt1r), dim3(RC_NT), 0, st1,
                         sbb.key[0], sbb.end[0], (u32)nb, irr_b, sa.key[0], (u32)na, irr_a, (i64)0, S.wlo1,
                         S.lo1, S.cnt1);
    } else {
    hipLaunchKernelGGL(k_count_partition, dim3(cdiv((u64)S.nt1 + 1, 256)), dim3(256), 0, st1,
                       sbb.key[0], (u32)nb, irr_b, sa.key[0], (u32)na, irr_a, (i64)0, c1_tq, S.nt1,
                       S.wlo1);
    if (S.c1_items == 2)
      hipLaunchKernelGGL(k_c1_count<2>, dim3(S.nt1), dim3(C1_NT), 0, st1, sbb.key[0], sbb.end[0], (u32)nb,
                         irr_b, sa.key[0], (u32)na, irr_a, S.wlo1, S.c1_base);
    else
      hipLaunchKernelGGL(k_c1_count<C1_ITEMS_MAX>, dim3(S.nt1), dim3(C1_NT), 0, st1, sbb.key[0], sbb.end[0], (u32)nb,
                         irr_b, sa.key[0], (u32)na, irr_a, S.wlo1, S.c1_base);
    // class-1 block totals -> block bases (one block, in place); total -> n_out_c1
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, st1, S.c1_base, S.nt1,
                       &ctx->d_meta->n_out_c1);
    }
    // class 2: queries = sorted A, points = sorted B starts, range (a.start, a.end)
    hipLaunchKernelGGL(k_count_partition, dim3(cdiv((u64)S.nt2 + 1, 256)), dim3(256), 0, st,
                       sa.key[0], (u32)na, irr_a, sbb.key[0], (u32)nb, irr_b, (i64)1, TQ2, S.nt2, S.wlo2);
    hipLaunchKernelGGL((k_range_count<RC_ITEMS_C2, RC_LDS_CAP>), dim3(S.nt2), dim3(RC_NT), 0, st,
                       sa.key[0], sa.end[0], (u32)na, irr_a, sbb.key[0], (u32)nb, irr_b, (i64)1, S.wlo2,
                       S.lo2, cnt2);
    GIQL_TRY(post_launch("range count"));
  }
  if (S.c1_fill) {
    GIQL_TRY(run_scan<u64>(ctx, st1, GIQL_PH_SCAN, S.cnt1, nb, S.off1, bsums1, S.off1 + nb, &ctx->d_meta->n_out_c1));
  }
  GIQL_TRY(run_scan<u64>(ctx, st, GIQL_PH_SCAN, cnt2, na, S.off2, bsums, S.off2 + na, &ctx->d_meta->n_out));
  GIQL_TRY(sc1.join());
  GIQL_TRY(read_meta(ctx, st));
  ctx->n_c1 = ctx->h_meta->n_out_c1;
  ctx->n_reg = ctx->h_meta->n_out + ctx->n_c1;
  }
  }
  if (onesweep) {
    int form;
    i64 len;
    decide(*ctx->h_meta, form, len);
    const bool aligned_now = ctx->h_meta->aligned_ok != 0;
    const bool coarse_wrong = coarse_q && ctx->h_meta->irr_a + ctx->h_meta->irr_b > 0;
    // a side sorted from its raw columns in the general form must hold no irregular row (its length
    // range starts above 0); such a row was keyed as if regular, and never listed
    const bool keygen_wrong = (keygen_g && (big_side ? ctx->h_meta->len_min_b : ctx->h_meta->len_min_a) <= 0) ||
                              (keygen_q && (big_side ? ctx->h_meta->len_min_a : ctx->h_meta->len_min_b) <= 0);
    if (keygen_wrong) ctx->last_no_irr = false;
    // the fused count's windows allow for query rows up to BS_FUSE_WCAP long (the query side of the form just decided)
    const int q_len_now = form == 1 ? ctx->h_meta->len_max_a : ctx->h_meta->len_max_b;
    const bool fuse_len_ok_now = form != 0 ? q_len_now <= (int)BS_FUSE_WCAP
                                           : (ctx->h_meta->len_max_a <= (int)BS_FUSE_WCAP &&
                                              ctx->h_meta->len_max_b <= (int)BS_FUSE_WCAP);  // (general form: both sides' rows)
    const bool fuse_wrong = ctx->count_fused && !fuse_len_ok_now;
    ctx->spec_fuse_len_ok = fuse_len_ok_now;
    // a side taken as sorted must be: an out-of-order row was left where it was
    const bool sorted_wrong = (ctx->used_sorted[0] && ctx->h_meta->unsorted_a != 0) ||
                              (ctx->used_sorted[1] && ctx->h_meta->unsorted_b != 0);
    ctx->spec_sorted[0] = ctx->h_meta->unsorted_a == 0;
    ctx->spec_sorted[1] = ctx->h_meta->unsorted_b == 0;
    // a join in the bucket stage that did not stand (the pairs did not fit the caller's buffers, or a guess failed)
    // left no plan arrays for the ordinary fill: plan again, unspeculated -- which never takes that form
    const bool join_wrong = ctx->bucket_join && !ctx->fuse_done;
    if (join_wrong && getenv("GIQL_HIP_DEBUG_JOIN"))
      fprintf(stderr, "[giql_hip] join in the bucket stage did not stand: %llu pairs, capacity %llu, irregular %u + %u\n",
              (unsigned long long)ctx->n_reg, (unsigned long long)ctx->fuse_cap, ctx->h_meta->irr_a, ctx->h_meta->irr_b);
    if (speculated && (form != S.uniform || len != uni_len || (want_hist && !aligned_now) || coarse_wrong || keygen_wrong || fuse_wrong || sorted_wrong || join_wrong)) {
      ctx->spec_valid = false;  // wrong guess: plan again from the numbers just read
      ctx->spec_misses++;
      ctx->fuse_done = false;
      return inner_plan_core(ctx, a, b, n_chrom, stream, n_pairs);
    }
    ctx->spec_valid = true;
    ctx->spec_form = form;
    ctx->spec_len = len;
    // the layout is probed only when the span histogram ran; a plan that skipped it for a
    // form mismatch leaves the last answer (a later form change re-plans unspeculated anyway)
    if (want_hist) ctx->spec_aligned = aligned_now;
  }
  ctx->stats.n_irregular_a = ctx->h_meta->irr_a;
  ctx->stats.n_irregular_b = ctx->h_meta->irr_b;
  ctx->stats.span = (int64_t)ctx->h_meta->total_span;
  ctx->last_span = ctx->h_meta->total_span;
  ctx->last_no_irr = ctx->h_meta->irr_a + ctx->h_meta->irr_b == 0;

  if (ctx->h_meta->irr_a + ctx->h_meta->irr_b > 0) {
    {
      Phase ph(ctx, st, GIQL_PH_IRREGULAR);
      hipLaunchKernelGGL(k_irr_count, dim3(cdiv(nq, 256)), dim3(256), 0, st, view_of(*a), view_of(*b),
                         ctx->irr_a_list, ctx->irr_b_list, ctx->d_meta, irr_cnt);
      GIQL_TRY(post_launch("irregular count"));
    }
    GIQL_TRY(run_scan<u64>(ctx, st, GIQL_PH_IRREGULAR, irr_cnt, nq, ctx->irr_off, bsums,
                           ctx->irr_off + nq));
    HIP_TRY(hipMemcpyAsync(&ctx->d_meta->n_out_irr, ctx->irr_off + nq, sizeof(u64),
                           hipMemcpyDeviceToDevice, st));
    GIQL_TRY(read_meta(ctx, st));
    ctx->n_irr = ctx->h_meta->n_out_irr;
  }
  collect_spans(ctx);
  // which form ran: 0 general, 1 B uniform, 2 A uniform; bit 4: the fixed-length side was sorted
  // from its raw columns (histogram in the span pass, no linearize pass)
  ctx->stats.reserved = S.uniform | ((keygen || keygen_g) ? 0x10 : 0);
  ctx->stats.n_out = (int64_t)(ctx->n_reg + ctx->n_irr);
  *n_pairs = (int64_t)(ctx->n_reg + ctx->n_irr);
  ctx->planned = true;
  ctx->plan_is_join = ctx->bucket_join;
  return GIQL_OK;
}

// The plan works with "B = the larger side": the general form keeps per-row arrays (bounds, counts, 64-bit
// offsets) for its class-2 queries, the A rows, and only block totals for class 1, the B rows, so the
// sides' roles are not symmetric in cost (10M x 100M reads: 4.4 ms; the same tables as (100M, 10M):
// 12.9 ms before this swap).  INTERSECTS is symmetric, so a call with the larger table first is planned
// with the sides exchanged and everything that leaves the context (pairs, stats, the exported plan) is
// labelled back; only the order of the pairs -- never part of the contract -- differs.
static int giql_hip_inner_plan_dev_impl(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b,
                                        int32_t n_chrom, void* stream, int64_t* n_pairs) {
  if (!ctx || !n_pairs) return set_err(GIQL_ERR_INVALID, "ctx/n_pairs is NULL");
  GIQL_TRY(check_side(a, "a"));
  GIQL_TRY(check_side(b, "b"));
  if (n_chrom < 0) return set_err(GIQL_ERR_INVALID, "n_chrom < 0");
  const bool swap = !ctx->no_swap && a->n > b->n;
  ctx->swapped = swap;
  if (!swap) return inner_plan_core(ctx, a, b, n_chrom, stream, n_pairs);
  // The offered output buffers follow the sides for the duration of THIS attempt only: with_order_fallback
  // may run this function again (a resort, a look-back timeout), and an exchange left in place would then
  // be undone by the second one -- the retry's early fill writing B ids into row_a.
  int32_t* const fa = ctx->fuse_a;
  int32_t* const fb = ctx->fuse_b;
  ctx->fuse_a = fb;
  ctx->fuse_b = fa;
  const int rc = inner_plan_core(ctx, b, a, n_chrom, stream, n_pairs);
  ctx->fuse_a = fa;
  ctx->fuse_b = fb;
  // stats in the caller's labels
  giql_hip_stats& stt = ctx->stats;
  const int64_t tn = stt.n_a;
  stt.n_a = stt.n_b;
  stt.n_b = tn;
  const int64_t ti = stt.n_irregular_a;
  stt.n_irregular_a = stt.n_irregular_b;
  stt.n_irregular_b = ti;
  const int form = stt.reserved & 0xF;
  if (form == 1 || form == 2) stt.reserved = (stt.reserved & ~0xF) | (3 - form);
  return rc;
}

int giql_hip_inner_plan_dev(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                            void* stream, int64_t* n_pairs) {
  return with_order_fallback(ctx, [&] { return giql_hip_inner_plan_dev_impl(ctx, a, b, n_chrom, stream, n_pairs); });
}

int giql_hip_inner_fill_dev(giql_hip_ctx* ctx, int32_t* row_a, int32_t* row_b, int64_t capacity,
                            void* stream) {
  if (!ctx) return set_err(GIQL_ERR_INVALID, "ctx is NULL");
  if (!ctx->planned) return set_err(GIQL_ERR_STATE, "inner_fill without a successful inner_plan");
  const u64 total = ctx->n_reg + ctx->n_irr;
  if (total == 0) return GIQL_OK;
  if (ctx->plan_is_join)
    return set_err(GIQL_ERR_STATE, "the last giql_hip_inner_join_dev wrote its pairs itself and kept no plan: "
                                   "call giql_hip_inner_plan_dev first");
  if (!row_a || !row_b) return set_err(GIQL_ERR_INVALID, "row_a/row_b is NULL");
  if (capacity < 0 || (u64)capacity < total)
    return set_err(GIQL_ERR_CAPACITY, "capacity %lld < %llu pairs", (long long)capacity,
                   (unsigned long long)total);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  if (ctx->swapped) {  // planned with the sides exchanged: the plan's "A" rows are the caller's B rows
    int32_t* t = row_a;
    row_a = row_b;
    row_b = t;
  }
  const u32 nq = ctx->n_a + ctx->n_b;
  InnerState& S = ctx->inner;
  const u32* irr_a = &ctx->d_meta->irr_a;
  const u32* irr_b = &ctx->d_meta->irr_b;
  const u64 p1 = ctx->n_c1, p2 = ctx->n_reg - ctx->n_c1;
  // who plays "query" in the range-fill: A rows (general class 2, or B uniform),
  // or B rows (A uniform)
  const bool q_is_a = S.uniform != 2;
  const u32 nq2 = q_is_a ? ctx->n_a : ctx->n_b;
  u32 nt2 = 0, nt1f = 0;
  const bool c1_fill = S.uniform == 0 && S.c1_fill && p1 > 0;
  if (p2 > 0 || c1_fill) {
    constexpr u32 T2 = FILL_NT * FILL_ITEMS_C2;
    const u64 nt2_64 = (p2 + T2 - 1) / T2, nt1_64 = c1_fill ? (p1 + T2 - 1) / T2 : 0;
    if (nt2_64 + nt1_64 > 0x7FFFFFF0ull) return set_err(GIQL_ERR_INVALID, "output too large");
    nt2 = (u32)nt2_64;
    nt1f = (u32)nt1_64;
    const size_t part_need = (size_t)nt2 + 2 + (c1_fill ? (size_t)nt1f + 2 : 0);
    if (part_need > ctx->part_cap) {
      HIP_TRY(hipStreamSynchronize(st));
      if (ctx->part) HIP_TRY(hipFree(ctx->part));
      ctx->part = nullptr;
      ctx->part_cap = 0;
      const size_t want = part_need + part_need / 4;
      HIP_TRY(hipMalloc((void**)&ctx->part, want * sizeof(u32)));
      ctx->part_cap = want;
    }
    Phase ph(ctx, st, GIQL_PH_PARTITION);
    if (p2 > 0)
      hipLaunchKernelGGL(k_partition, dim3(cdiv((u64)nt2 + 1, 256)), dim3(256), 0, st, S.off2, nq2,
                         (u64)0, T2, nt2, ctx->part);
    if (c1_fill)  // class 1's tiles behind class 2's in the partition array
      hipLaunchKernelGGL(k_partition, dim3(cdiv((u64)nt1f + 1, 256)), dim3(256), 0, st, S.off1, ctx->n_b,
                         (u64)0, T2, nt1f, ctx->part + nt2 + 2);
  }
  {
    Phase ph(ctx, st, GIQL_PH_FILL, 2);
    // class 1 -> outputs [0, p1): query = B row, matches = A rows
    if (c1_fill)
      hipLaunchKernelGGL((k_fill<FILL_ITEMS_C2>), dim3(nt1f), dim3(FILL_NT), 0, st, S.off1, S.lo1, S.sb.rid[0],
                         ctx->n_b, S.sa.rid[0], ctx->part + nt2 + 2, (u64)0, p1, row_b, row_a);
    else if (p1 > 0 && S.c1_items == 2)
      hipLaunchKernelGGL(k_c1_emit<2>, dim3(S.nt1), dim3(C1_NT), 0, st, S.sb.key[0], S.sb.end[0],
                         S.sb.rid[0], ctx->n_b, irr_b, S.sa.key[0], S.sa.rid[0], ctx->n_a, irr_a,
                         S.wlo1, S.c1_base, (u64)0, row_b, row_a);
    else if (p1 > 0)
      hipLaunchKernelGGL(k_c1_emit<C1_ITEMS_MAX>, dim3(S.nt1), dim3(C1_NT), 0, st, S.sb.key[0], S.sb.end[0],
                         S.sb.rid[0], ctx->n_b, irr_b, S.sa.key[0], S.sa.rid[0], ctx->n_a, irr_a,
                         S.wlo1, S.c1_base, (u64)0, row_b, row_a);
    // range fill -> outputs [p1, p1 + p2)
    if (p2 > 0) {
      const u32* qrid = q_is_a ? S.sa.rid[0] : S.sb.rid[0];
      const u32* srid = q_is_a ? S.sb.rid[0] : S.sa.rid[0];
      int32_t* rq = (q_is_a ? row_a : row_b) + p1;
      int32_t* rs = (q_is_a ? row_b : row_a) + p1;
      hipLaunchKernelGGL((k_fill<FILL_ITEMS_C2>), dim3(nt2), dim3(FILL_NT), 0, st, S.off2, S.lo2, qrid,
                         nq2, srid, ctx->part, (u64)0, p2, rq, rs);
    }
    GIQL_TRY(post_launch("fill"));
  }
  if (ctx->n_irr > 0) {
    Phase ph(ctx, st, GIQL_PH_IRREGULAR);
    hipLaunchKernelGGL(k_irr_fill, dim3(cdiv(nq, 256)), dim3(256), 0, st, view_of(ctx->side_a),
                       view_of(ctx->side_b), ctx->irr_a_list, ctx->irr_b_list, ctx->d_meta,
                       ctx->irr_off, row_a + ctx->n_reg, row_b + ctx->n_reg);
    GIQL_TRY(post_launch("irregular fill"));
  }
  return GIQL_OK;
}

// Plan + fill in one call into caller-owned buffers.  When the context's guesses hold (same
// join form as its previous plan, no irregular rows, the pairs fit `capacity`) the fill has
// already been launched inside the plan, with no stream sync between the two; otherwise the
// ordinary fill runs here.  GIQL_ERR_CAPACITY leaves the plan valid: *n_pairs tells the
// size to offer to giql_hip_inner_fill_dev.
int giql_hip_inner_join_dev(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                            int32_t* row_a, int32_t* row_b, int64_t capacity, void* stream,
                            int64_t* n_pairs) {
  if (!ctx || !n_pairs) return set_err(GIQL_ERR_INVALID, "ctx/n_pairs is NULL");
  if (capacity < 0 || (capacity > 0 && (!row_a || !row_b)))
    return set_err(GIQL_ERR_INVALID, "bad output buffers");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  {  // the merge-path partition array must exist before the plan can launch the fill
    constexpr u32 T2 = FILL_NT * FILL_ITEMS_C2;
    const u64 nt_cap = ((u64)capacity + T2 - 1) / T2;
    const size_t part_need = (size_t)nt_cap + 2;
    if (nt_cap <= 0x7FFFFFF0ull && part_need > ctx->part_cap) {
      HIP_TRY(hipStreamSynchronize(st));
      if (ctx->part) HIP_TRY(hipFree(ctx->part));
      ctx->part = nullptr;
      ctx->part_cap = 0;
      const size_t want = part_need + part_need / 4;
      HIP_TRY(hipMalloc((void**)&ctx->part, want * sizeof(u32)));
      ctx->part_cap = want;
    }
  }
  ctx->fuse_a = row_a;
  ctx->fuse_b = row_b;
  ctx->fuse_cap = (u64)capacity;
  const int rc = giql_hip_inner_plan_dev(ctx, a, b, n_chrom, stream, n_pairs);
  ctx->fuse_a = ctx->fuse_b = nullptr;
  ctx->fuse_cap = 0;
  if (rc != GIQL_OK) return rc;
  if (ctx->fuse_done) return GIQL_OK;
  if (*n_pairs > capacity)
    return set_err(GIQL_ERR_CAPACITY, "capacity %lld < %lld pairs", (long long)capacity, (long long)*n_pairs);
  return giql_hip_inner_fill_dev(ctx, row_a, row_b, capacity, stream);
}

// ------------------------------------------------------------- table index
// The reference tells its users to CREATE INDEX ... (chrom, start, "end") on both join sides
// (docs/transpilation/performance.rst:111-130): what the engine keeps between queries.  Here: the properties of a
// TABLE that every join over it recomputes -- span and chromosome bases, the fixed length, the (key, rid) rows
// grouped by 65,536-key bucket and sorted -- kept in HBM as an explicit object.  A join against an index
// (giql_hip_inner_join_indexed_dev) is the other side's span pass + sort + the bucket stage: the larger table's
// span pass (1.2 GB read at 100M rows) and its two global sort passes (3.2 GB) are not repeated.  An extra, never
// the headline: bench.py times it as its own workload beside the build time of the index.
struct giql_hip_index {
  int device = 0;
  u32 n = 0;
  int n_chrom = 0;
  bool general = false;    // rows of any length: (key, end, rid); else fixed length: (key, rid)
  i64 uni_len = 0;         // the fixed canonical length (0 in the general form)
  int len_max = 0;         // the longest row
  u64 span = 0;
  u32 sentinel = 0;        // one past the largest key of the axis
  int wbits = 16;          // key bits of a bucket (16; 15 / 14 / 13 for tables past ~2,800 rows per 65,536 positions)
  u32 *key = nullptr, *end = nullptr, *rid = nullptr;   // [n] sorted by key (every bucket sorted in place)
  u32* small = nullptr;    // the sort's digit offsets gbase[4][256] | first[n_chrom + 1]: chromosome c owns keys [first[c], first[c + 1])
  size_t bytes = 0;
};

int giql_hip_index_destroy(giql_hip_index* idx) {
  if (!idx) return GIQL_OK;
  (void)hipSetDevice(idx->device);
  if (idx->key) (void)hipFree(idx->key);
  if (idx->end) (void)hipFree(idx->end);
  if (idx->rid) (void)hipFree(idx->rid);
  if (idx->small) (void)hipFree(idx->small);
  delete idx;
  return GIQL_OK;
}

int giql_hip_index_info(const giql_hip_index* idx, int64_t* n_rows, int64_t* bytes, int32_t* general, int64_t* span) {
  if (!idx) return set_err(GIQL_ERR_INVALID, "index is NULL");
  if (n_rows) *n_rows = idx->n;
  if (bytes) *bytes = (int64_t)idx->bytes;
  if (general) *general = idx->general ? 1 : 0;
  if (span) *span = (int64_t)idx->span;
  return GIQL_OK;
}

int giql_hip_index_create_dev(giql_hip_ctx* ctx, const giql_side* side, int32_t n_chrom, void* stream,
                              giql_hip_index** out) {
  if (!ctx || !out) return set_err(GIQL_ERR_INVALID, "ctx/out is NULL");
  *out = nullptr;
  GIQL_TRY(check_side(side, "side"));
  if (n_chrom < 1 || n_chrom > MM_HIST_CHROMS)
    return set_err(GIQL_ERR_STATE, "a table index takes 1..%d chromosomes (the 2^24-aligned axis), got %d", MM_HIST_CHROMS, n_chrom);
  if (side->n < 1 || (size_t)side->n > OS_MAX_ROWS) return set_err(GIQL_ERR_STATE, "a table index takes 1..2^30-1 rows");
  if (ctx->classic_sort || ctx->os_variant != 0 || !ctx->bucket_bnd)
    return set_err(GIQL_ERR_STATE, "this context cannot run the three-stage sort (GIQL_HIP_SORT / GIQL_HIP_OS_VARIANT)");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  ctx->planned = false;
  reset_stats(ctx);
  const size_t n = (size_t)side->n;
  LinBufs lb;
  SortBufs scratch;  // buffer 1 of the sort (buffer 0 = the index's own arrays)
  u32 *hist = nullptr, *gbase = nullptr, *status = nullptr;
  size_t zero_off = 0, zero_end = 0;
  auto carve = [&](char* base) {
    Carver c{base};
    common_sizes(c, n_chrom, lb);
    scratch.key[1] = c.take<u32>(n);
    scratch.end[1] = c.take<u32>(n);
    scratch.rid[1] = c.take<u32>(n);
    c.off = align_up(c.off, 256);
    zero_off = c.off;
    hist = c.take<u32>((size_t)LIN_HIST_REPLICAS * 1024);
    lb.top_partial = c.take<u32>((size_t)LIN_HIST_REPLICAS * MM_TOP_WORDS);
    gbase = c.take<u32>(1024);
    lb.abase = c.take<u32>(MM_HIST_CHROMS);
    status = c.take<u32>(3 * os_pass_words(n));
    c.off = align_up(c.off, 256);
    zero_end = c.off;
    return c.off;
  };
  GIQL_TRY(ensure_arena(ctx, carve(nullptr), st));
  carve(ctx->arena);
  giql_hip_index* idx = new (std::nothrow) giql_hip_index();
  if (!idx) return set_err(GIQL_ERR_NOMEM, "out of host memory");
  struct Fail {  // released on every early way out
    giql_hip_index* p;
    giql_hip_ctx* c;
    ~Fail() {
      c->force_local = 0;
      c->index_bits = 16;
      c->prezeroed = false;
      if (p) giql_hip_index_destroy(p);
    }
  } guard{idx, ctx};
  idx->device = ctx->device;
  idx->n = (u32)n;
  idx->n_chrom = n_chrom;
  HIP_TRY(hipMemsetAsync(ctx->arena + zero_off, 0, zero_end - zero_off, st));
  ctx->prezeroed = true;
  ctx->force_local = 1;
  ctx->index_bits = 15;  // (not 16: the span pass counts the bits 8-15 digit too -- the density is not known yet)
  ctx->span_hist_dirty[0] = ctx->span_hist_dirty[1] = nullptr;
  giql_side none = *side;
  none.n = 0;
  none.chrom = none.start = none.end = nullptr;
  // the span pass of the ordinary plan, this table as its side B: per-chromosome range, length range, the digits
  // of the aligned keys (all but the lowest matter: buckets narrower than 2^16 keys sort on bits 8-15 as well)
  GIQL_TRY(run_spans(ctx, st, none, *side, n_chrom, lb, 1, hist));
  GIQL_TRY(read_meta(ctx, st));
  const DevMeta& m = *ctx->h_meta;
  if (!m.aligned_ok)
    return set_err(GIQL_ERR_STATE, "the table does not take the aligned axis (a negative coordinate, or more than 255 2^24-position blocks)");
  if (m.len_min_b <= 0) return set_err(GIQL_ERR_STATE, "the table holds irregular rows (canonical end <= start): not indexable");
  if (m.len_max_b > (int)BS_FUSE_WCAP)
    return set_err(GIQL_ERR_STATE, "a row of %d positions is longer than the bucket stage's windows allow (%u)", m.len_max_b, BS_FUSE_WCAP);
  const double per_bucket = (double)n * 65536.0 / (double)(m.total_span ? m.total_span : 1);
  idx->wbits = ctx->force_bits ? ctx->force_bits : density_bits(ctx, per_bucket);
  if (idx->wbits == 0)
    return set_err(GIQL_ERR_STATE, "%.0f rows per 65,536 positions: too dense for the in-LDS bucket stage (at most %.0f per %d)",
                   per_bucket, ctx->local_max_bucket_rows, 1 << BS_MIN_WBITS);
  ctx->index_bits = idx->wbits;
  idx->general = ctx->no_uniform || m.len_min_b != m.len_max_b;
  idx->uni_len = idx->general ? 0 : m.len_max_b;
  idx->len_max = m.len_max_b;
  idx->span = m.total_span;
  idx->sentinel = m.sentinel;
  HIP_TRY(hipMalloc((void**)&idx->key, n * sizeof(u32)));
  HIP_TRY(hipMalloc((void**)&idx->rid, n * sizeof(u32)));
  if (idx->general) HIP_TRY(hipMalloc((void**)&idx->end, n * sizeof(u32)));
  HIP_TRY(hipMalloc((void**)&idx->small, (1024 + (size_t)n_chrom + 1) * sizeof(u32)));
  idx->bytes = n * sizeof(u32) * (idx->general ? 3 : 2) + (1024 + (size_t)n_chrom + 1) * sizeof(u32);
  {
    Phase ph(ctx, st, GIQL_PH_LINEARIZE, 2);
    hipLaunchKernelGGL(k_fold_top, dim3(MM_HIST_CHROMS), dim3(256), 0, st, lb.top_partial, lb.abase, hist);
    hipLaunchKernelGGL(k_digit_offsets, dim3(4), dim3(256), 0, st, hist, (u32)LIN_HIST_REPLICAS, gbase);
    GIQL_TRY(post_launch("digit offsets (index)"));
  }
  SortBufs sb = scratch;
  sb.key[0] = idx->key;
  sb.rid[0] = idx->rid;
  sb.end[0] = idx->general ? idx->end : nullptr;
  if (!idx->general) sb.end[1] = nullptr;
  if (local_passes(idx->wbits) & 1) {
    // three global passes end in the buffer they did not start from (run_sort_onesweep then calls THAT "buffer 0"):
    // the index's own arrays start as buffer 1
    u32* t;
    t = sb.key[0], sb.key[0] = sb.key[1], sb.key[1] = t;
    t = sb.rid[0], sb.rid[0] = sb.rid[1], sb.rid[1] = t;
    t = sb.end[0], sb.end[0] = sb.end[1], sb.end[1] = t;
  }
  // two (three) global passes from the raw columns + every bucket sorted in LDS, in place: the result is in buffer 0
  GIQL_TRY(run_sort_onesweep(ctx, st, sb, (u32)n, gbase, status, false, side, lb.abase, 0, nullptr, false));
  if (sb.key[0] != idx->key) return set_err(GIQL_ERR_HIP, "internal: the sort did not end in the index's buffers");
  HIP_TRY(hipMemcpyAsync(idx->small, gbase, 1024 * sizeof(u32), hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(idx->small + 1024, lb.chrom_first, ((size_t)n_chrom + 1) * sizeof(u32), hipMemcpyDeviceToDevice, st));
  {
    const int rc = read_meta(ctx, st);
    if (rc == GIQL_STATUS_RESORT)
      return set_err(GIQL_ERR_STATE, "a %d-key bucket of the table holds more rows than the bucket stage sorts (%u)", 1 << idx->wbits, BS_BIG_MAX);
    if (rc != GIQL_OK) return rc;
  }
  collect_spans(ctx);
  ctx->stats.n_b = side->n;
  ctx->stats.span = (int64_t)idx->span;
  guard.p = nullptr;
  *out = idx;
  return GIQL_OK;
}

// INNER join of `a` against an indexed table: pairs (row of a, row of the indexed table).  Per call: a's span pass
// (lengths only), its keys on the index's axis, its sort (grouped by bucket: two passes) and the bucket stage over
// the index's rows -- the one-call join's last stage, reading the index instead of a freshly sorted side.
// GIQL_ERR_STATE: `a` holds irregular rows or rows too long for the windows (the ordinary join answers those);
// GIQL_ERR_CAPACITY with *n_pairs set when the buffers are short.
int giql_hip_inner_join_indexed_dev(giql_hip_ctx* ctx, const giql_hip_index* idx, const giql_side* a,
                                    int32_t* row_a, int32_t* row_idx, int64_t capacity, void* stream,
                                    int64_t* n_pairs) {
  if (!ctx || !idx || !n_pairs) return set_err(GIQL_ERR_INVALID, "ctx/index/n_pairs is NULL");
  GIQL_TRY(check_side(a, "a"));
  if (idx->device != ctx->device) return set_err(GIQL_ERR_INVALID, "the index lives on device %d, the context on %d", idx->device, ctx->device);
  if (capacity < 0 || (capacity > 0 && (!row_a || !row_idx))) return set_err(GIQL_ERR_INVALID, "bad output buffers");
  *n_pairs = 0;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  ctx->planned = false;
  ctx->fuse_done = false;
  reset_stats(ctx);
  ctx->stats.n_a = a->n;
  ctx->stats.n_b = idx->n;
  if (a->n == 0) return GIQL_OK;
  const size_t na = (size_t)a->n, nb = (size_t)idx->n;
  if (na > OS_MAX_ROWS) return set_err(GIQL_ERR_INVALID, "side larger than 2^30 rows");
  SortBufs sa, sq;  // a's sort; buffer 1 of the index side (only a queued bucket's block ever touches it)
  u32 *hist = nullptr, *gbase = nullptr, *status = nullptr, *flags = nullptr;
  size_t zero_off = 0, zero_end = 0;
  auto carve = [&](char* base) {
    Carver c{base};
    sort_sizes(c, na, sa, true);
    sq.key[1] = c.take<u32>(nb);
    sq.rid[1] = c.take<u32>(nb);
    sq.end[1] = idx->general ? c.take<u32>(nb) : nullptr;
    c.off = align_up(c.off, 256);
    zero_off = c.off;
    hist = c.take<u32>((size_t)LIN_HIST_REPLICAS * 1024);
    gbase = c.take<u32>(1024);
    flags = c.take<u32>(64);
    status = c.take<u32>(4 * os_pass_words(na));
    c.off = align_up(c.off, 256);
    zero_end = c.off;
    return c.off;
  };
  GIQL_TRY(ensure_arena(ctx, carve(nullptr), st));
  carve(ctx->arena);
  struct Guard {
    giql_hip_ctx* c;
    ~Guard() { c->prezeroed = false; }
  } guard{ctx};
  HIP_TRY(hipMemsetAsync(ctx->arena + zero_off, 0, zero_end - zero_off, st));
  ctx->prezeroed = true;
  const u32* first = idx->small + 1024;
  u32* const dead = flags;           // rows of a that cannot match (sorted last, skipped by the windows)
  u32* const irregular = flags + 1;
  int* const len_max_q = reinterpret_cast<int*>(flags + 2);
  int* const len_max_u = reinterpret_cast<int*>(flags + 3);
  {
    Phase ph(ctx, st, GIQL_PH_LINEARIZE, 3);
    hipLaunchKernelGGL(k_init_minmax, dim3(1), dim3(256), 0, st, (int*)nullptr, (int*)nullptr, 0, ctx->d_meta);
    u32 grid = cdiv((u64)na, LIN_NT);
    if (grid > (u32)LIN_MAX_BLOCKS) grid = LIN_MAX_BLOCKS;
    hipLaunchKernelGGL(k_index_query_keys, dim3(grid), dim3(LIN_NT), 0, st, a->chrom, a->start, a->end, (u32)na,
                       a->start_off, a->end_off, idx->n_chrom, first, idx->sentinel, sa.key[0], sa.end[0], dead, irregular,
                       len_max_q, hist);
    hipLaunchKernelGGL(k_digit_offsets, dim3(4), dim3(256), 0, st, hist, (u32)LIN_HIST_REPLICAS, gbase);
    GIQL_TRY(post_launch("query keys (index)"));
  }
  HIP_TRY(hipMemcpyAsync(len_max_u, &idx->len_max, sizeof(int), hipMemcpyHostToDevice, st));
  // grouped by bucket only (two passes): the windows are computed under the same mask.  The general form ranks a
  // window's keys against the bucket rows and wants them fully sorted.
  // (narrower buckets: grouped by key >> 8, three passes)
  const int q_skip = idx->general ? 0 : (idx->wbits == 16 ? 2 : 1);
  const u32 q_mask = q_skip == 2 ? 0xFFFF0000u : (q_skip == 1 ? 0xFFFFFF00u : 0xFFFFFFFFu);
  ctx->force_local = -1;  // a's own sort: global passes only (its bucket stage would reuse the context's boundary arrays)
  const int rc_sort = run_sort_onesweep(ctx, st, sa, (u32)na, gbase, status, false, nullptr, nullptr, q_skip, nullptr, false);
  ctx->force_local = 0;
  GIQL_TRY(rc_sort);
  FuseCount fc;
  fc.join = true;
  fc.general = idx->general;
  fc.zero_ptr = flags + 8;
  fc.zero_words = 0;
  fc.dev.qkey = sa.key[0];
  fc.dev.qend = sa.end[0];
  fc.dev.qrid = sa.rid[0];
  fc.dev.qwin = ctx->bucket_qwin;
  fc.dev.lo_out = fc.dev.hi_out = nullptr;
  fc.dev.lo_off = idx->general ? 1 : 1 - idx->uni_len;
  fc.dev.row_q = row_a;
  fc.dev.row_s = row_idx;
  fc.dev.cap = (u64)capacity;
  fc.dev.cursor = reinterpret_cast<unsigned long long*>(&ctx->d_meta->n_out);  // zeroed by k_init_minmax
  fc.nq_total = (u32)na;
  fc.irr_q = dead;
  fc.gbq3 = gbase + 3 * OS_BINS;
  fc.key_mask = q_mask;
  fc.len_max_q = len_max_q;
  fc.len_max_u = idx->general ? len_max_u : nullptr;
  SortBufs sb = sq;
  sb.key[0] = idx->key;
  sb.rid[0] = idx->rid;
  sb.end[0] = idx->end;
  ctx->last_sort_local = true;
  launch_bucket_stage_fused(ctx, st, sb, (u32)nb, idx->small, fc, idx->wbits);
  GIQL_TRY(post_launch("bucket stage (index)"));
  u32 h_flags[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(h_flags, flags, sizeof(h_flags), hipMemcpyDeviceToHost, st));
  {
    const int rc = read_meta(ctx, st);   // (synchronises the stream: h_flags has arrived too)
    if (rc == GIQL_STATUS_RESORT) return set_err(GIQL_ERR_STATE, "a bucket of the index is too large for the bucket stage");
    if (rc != GIQL_OK) return rc;
  }
  collect_spans(ctx);
  if (h_flags[1]) return set_err(GIQL_ERR_STATE, "the query table holds irregular rows (canonical end <= start): use the ordinary join");
  if ((int)h_flags[2] > (int)BS_FUSE_WCAP)
    return set_err(GIQL_ERR_STATE, "a query row of %d positions is longer than the bucket stage's windows allow (%u): use the ordinary join",
                   (int)h_flags[2], BS_FUSE_WCAP);
  const u64 total = ctx->h_meta->n_out;
  *n_pairs = (int64_t)total;
  ctx->stats.n_out = (int64_t)total;
  ctx->stats.span = (int64_t)idx->span;
  ctx->stats.reserved = idx->general ? 0 : 1;
  ctx->stats.phase_bytes[GIQL_PH_SORT_LOCAL] += (int64_t)8 * (int64_t)total;
  if (total > (u64)capacity)
    return set_err(GIQL_ERR_CAPACITY, "capacity %lld < %llu pairs", (long long)capacity, (unsigned long long)total);
  return GIQL_OK;
}

// Scratch shared by the single-output operators: histogram replicas, digit bases,
// onesweep status words (+ tile-claim state) for one side at a time.
struct OsScratch {
  u32 *hist = nullptr, *gbase = nullptr, *hist_e = nullptr, *gbase_e = nullptr;
  u32* status = nullptr;
};

// ONE memset for everything two sides' histograms and sort passes need at zero, instead of one per histogram and per
// sort (four launches of ~5 us on a path of ~60): `first` is carved right before `second`, so the range runs from
// first's histograms to the end of second's four passes of status words.  Only for calls that sort each side ONCE (a
// second sort through the same status words must zero them again: ctx->prezeroed makes the helpers skip theirs).
struct PrezeroGuard {
  giql_hip_ctx* c;
  ~PrezeroGuard() { c->prezeroed = false; }
};
static int prezero_row_scratch(giql_hip_ctx* ctx, hipStream_t st, const OsScratch& first, const OsScratch& second,
                               size_t n_second) {
  char* const lo = reinterpret_cast<char*>(first.hist);
  char* const hi = reinterpret_cast<char*>(second.status + 4 * os_pass_stride(ctx, n_second));
  if (!first.hist || hi <= lo) return GIQL_OK;
  HIP_TRY(hipMemsetAsync(lo, 0, (size_t)(hi - lo), st));
  ctx->prezeroed = true;
  return GIQL_OK;
}

static void os_scratch_sizes(Carver& c, size_t n_max, OsScratch& o) {
  o.hist = c.take<u32>((size_t)LIN_HIST_REPLICAS * 1024);
  o.hist_e = c.take<u32>((size_t)LIN_HIST_REPLICAS * 1024);
  o.gbase = c.take<u32>(1024);
  o.gbase_e = c.take<u32>(1024);
  o.status = c.take<u32>(4 * os_pass_words(n_max));
}

// -------------------------------------------------------------- SEMI / ANTI
static int giql_hip_semi_anti_dev_impl(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b,
                           int32_t n_chrom, int anti, int32_t* rows_out, int64_t* n_out,
                           void* stream) {
  if (!ctx || !n_out) return set_err(GIQL_ERR_INVALID, "ctx/n_out is NULL");
  GIQL_TRY(check_side(a, "a"));
  GIQL_TRY(check_side(b, "b"));
  if (n_chrom < 0) return set_err(GIQL_ERR_INVALID, "n_chrom < 0");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  ctx->planned = false;
  ctx->prezeroed = false;  // (a repeated call -- a guess that missed -- starts over: the caller's frame may still hold its guard)
  reset_stats(ctx);
  ctx->stats.n_a = a->n;
  ctx->stats.n_b = b->n;
  *n_out = 0;
  if (a->n == 0) return GIQL_OK;
  if (!rows_out) return set_err(GIQL_ERR_INVALID, "rows_out is NULL");
  const size_t na = (size_t)a->n, nb = (size_t)b->n;
  if (na > OS_MAX_ROWS || nb > OS_MAX_ROWS) return set_err(GIQL_ERR_INVALID, "side larger than 2^30 rows");
  const int nch = n_chrom > 0 ? n_chrom : 1;

  LinBufs lb;
  SortBufs sa, sbb;
  OsScratch os, os_a;
  u32 *flag = nullptr, *pmax = nullptr, *bmax = nullptr, *dummy_irr = nullptr;
  u64 *bsums = nullptr, *off = nullptr;
  auto carve = [&](char* base) {
    Carver c{base};
    common_sizes(c, nch, lb);
    sort_sizes(c, na, sa, true);
    for (int k = 0; k < 2; k++) {  // B: (key, end), no row ids
      sbb.key[k] = c.take<u32>(nb ? nb : 1);
      sbb.end[k] = c.take<u32>(nb ? nb : 1);
      sbb.rid[k] = nullptr;
    }
    os_scratch_sizes(c, na, os_a);   // A's own histogram / status words: its chain may run beside B's
    os_scratch_sizes(c, nb ? nb : 1, os);  // (right behind A's: prezero_row_scratch)
    bsums = c.take<u64>(cdiv(na, SCAN_TILE) + 2);
    flag = c.take<u32>(na);
    off = c.take<u64>(na + 1);
    pmax = c.take<u32>(nb ? nb : 1);
    bmax = c.take<u32>(cdiv(nb ? nb : 1, PM_TILE) + 1);
    dummy_irr = c.take<u32>(16);
    return c.off;
  };
  GIQL_TRY(ensure_arena(ctx, carve(nullptr), st));
  carve(ctx->arena);
  PrezeroGuard prezero_guard{ctx};
  GIQL_TRY(prezero_row_scratch(ctx, st, os_a, os, nb ? nb : 1));  // each side is sorted once, in either form

  GIQL_TRY(run_spans(ctx, st, *a, *b, nch, lb));
  i64 uni_len = 0;
  bool speculated = false;
  if (nb > 0) GIQL_TRY(row_form_guess(ctx, st, uni_len, speculated));
  const bool coarse_b = nb > 0 && uni_len > 0 && coarse_b_ok(ctx, nb);
  {
    // the query side's chain (linearize + sort) beside B's when it is small
    SideChain sc(ctx, st, nb > 0 ? na : 0, nb);
    GIQL_TRY(run_linearize(ctx, sc.stream(), *a, nch, lb, sa.key[0], sa.end[0], dummy_irr + 8, 0, 1, os_a.hist, os_a.gbase));
    GIQL_TRY(run_sort_onesweep(ctx, sc.stream(), sa, (u32)na, os_a.gbase, os_a.status, false, nullptr, nullptr,
                               /*skip_digits=*/row_skip(ctx, nb)));  // the query side's order only serves locality; every row keeps its real key here
  if (nb > 0 && uni_len > 0) {
    // fixed-length B: keys only (its `end` column is not read again), no prefix max
    sbb.end[0] = sbb.end[1] = nullptr;
    GIQL_TRY(run_linearize(ctx, st, *b, nch, lb, sbb.key[0], nullptr, dummy_irr, 1, 1, os.hist, os.gbase,
                           nullptr, nullptr, /*skip_end=*/true));
    GIQL_TRY(run_sort_onesweep(ctx, st, sbb, (u32)nb, os.gbase, os.status, false, nullptr, nullptr,
                               /*skip_digits=*/coarse_b ? 1 : 0));
  } else if (nb > 0) {
    // every B row keeps its real key: the prefix-max test is exact for any row
    GIQL_TRY(run_linearize(ctx, st, *b, nch, lb, sbb.key[0], sbb.end[0], dummy_irr, 1, 1, os.hist,
                           os.gbase));
#if defined(GIQL_LIN_ABLATE)  // timing-only builds stop here: their keys / histograms are not valid
    GIQL_TRY(read_meta(ctx, st));
    collect_spans(ctx);
    return GIQL_OK;
#endif
    GIQL_TRY(run_sort_onesweep(ctx, st, sbb, (u32)nb, os.gbase, os.status));
    GIQL_TRY(run_pmax(ctx, st, sbb.end[0], (u32)nb, pmax, bmax));
  }
    GIQL_TRY(sc.join());
  }
  {
    Phase ph(ctx, st, GIQL_PH_COUNT);
    if (coarse_b)
      hipLaunchKernelGGL(k_semi_flags_uniform_coarse, dim3(cdiv(na, 256)), dim3(256), 0, st, sa.key[0], sa.end[0],
                         sa.rid[0], (u32)na, ctx->d_meta, sbb.key[0], (u32)nb, uni_len, anti, flag);
    else if (nb > 0 && uni_len > 0)
      hipLaunchKernelGGL(k_semi_flags_uniform, dim3(cdiv(na, 256)), dim3(256), 0, st, sa.key[0], sa.end[0],
                         sa.rid[0], (u32)na, ctx->d_meta, sbb.key[0], (u32)nb, uni_len, anti, flag);
    else
      hipLaunchKernelGGL(k_semi_flags, dim3(cdiv(na, 256)), dim3(256), 0, st, sa.key[0], sa.end[0],
                         sa.rid[0], (u32)na, ctx->d_meta, sbb.key[0], pmax, (u32)nb, anti, flag);
    GIQL_TRY(post_launch("semi flags"));
  }
  {
    // scan of the flags with the compaction in its down-sweep, the count straight into DevMeta::n_out
    // (three launches; were five: scan x 3, a device-to-device copy of the total, compact)
    const u32 nbk = cdiv(na, SCAN_TILE);
    Phase ph(ctx, st, GIQL_PH_SCAN, 3);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nbk), dim3(SCAN_NT), 0, st, flag, (u64)na, bsums);
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, st, bsums, nbk, off + na, &ctx->d_meta->n_out);
    hipLaunchKernelGGL(k_scan_down_compact, dim3(nbk), dim3(SCAN_NT), 0, st, flag, (u64)na, bsums, rows_out);
    GIQL_TRY(post_launch("scan + compact"));
  }
  GIQL_TRY(read_meta(ctx, st));
  if (nb > 0 && !row_form_settled(ctx, uni_len, speculated))  // B is not fixed-length after all
    return giql_hip_semi_anti_dev_impl(ctx, a, b, n_chrom, anti, rows_out, n_out, stream);
  collect_spans(ctx);
  ctx->stats.reserved = (uni_len > 0 ? 1 : 0) | (coarse_b ? 0x10 : 0);  // form: fixed-length B or general; bit 4: B sorted coarsely (without its lowest digit)
  *n_out = (int64_t)ctx->h_meta->n_out;
  ctx->stats.n_out = *n_out;
  ctx->stats.span = (int64_t)ctx->h_meta->total_span;
  ctx->last_span = ctx->h_meta->total_span;
  return GIQL_OK;
}

int giql_hip_semi_anti_dev(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                           int anti, int32_t* rows_out, int64_t* n_out, void* stream) {
  return with_order_fallback(ctx, [&] { return giql_hip_semi_anti_dev_impl(ctx, a, b, n_chrom, anti, rows_out, n_out, stream); });
}

// ------------------------------------------------------------------- COUNT
static int giql_hip_count_dev_impl(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                       int64_t* counts_out, void* stream) {
  if (!ctx) return set_err(GIQL_ERR_INVALID, "ctx is NULL");
  GIQL_TRY(check_side(a, "a"));
  GIQL_TRY(check_side(b, "b"));
  if (n_chrom < 0) return set_err(GIQL_ERR_INVALID, "n_chrom < 0");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  ctx->planned = false;
  ctx->prezeroed = false;  // (a repeated call -- a guess that missed -- starts over: the caller's frame may still hold its guard)
  reset_stats(ctx);
  ctx->stats.n_a = a->n;
  ctx->stats.n_b = b->n;
  if (a->n == 0) return GIQL_OK;
  if (!counts_out) return set_err(GIQL_ERR_INVALID, "counts_out is NULL");
  const size_t na = (size_t)a->n, nb = (size_t)b->n;
  if (na > OS_MAX_ROWS || nb > OS_MAX_ROWS) return set_err(GIQL_ERR_INVALID, "side larger than 2^30 rows");
  if (nb == 0 || n_chrom == 0) {
    HIP_TRY(hipMemsetAsync(counts_out, 0, na * sizeof(int64_t), st));
    return GIQL_OK;
  }
  LinBufs lb;
  SortBufs sa, sstart, send;
  OsScratch os, os_a;
  u32 *irr_a_list = nullptr, *irr_b_list = nullptr;
  auto carve = [&](char* base) {
    Carver c{base};
    common_sizes(c, n_chrom, lb);
    sort_sizes(c, na, sa, true);
    for (int k = 0; k < 2; k++) {  // two keys-only sorts of B: starts and ends
      sstart.key[k] = c.take<u32>(nb);
      sstart.end[k] = sstart.rid[k] = nullptr;
      send.key[k] = c.take<u32>(nb);
      send.end[k] = send.rid[k] = nullptr;
    }
    os_scratch_sizes(c, na, os_a);   // A's chain may run beside B's (SideChain)
    os_scratch_sizes(c, nb, os);     // (right behind A's: prezero_row_scratch)
    irr_a_list = c.take<u32>(na);
    irr_b_list = c.take<u32>(nb);
    return c.off;
  };
  GIQL_TRY(ensure_arena(ctx, carve(nullptr), st));
  carve(ctx->arena);

  GIQL_TRY(run_spans(ctx, st, *a, *b, n_chrom, lb));
  i64 uni_len = 0;
  bool speculated = false;
  GIQL_TRY(row_form_guess(ctx, st, uni_len, speculated));
  const bool coarse_b = uni_len > 0 && coarse_b_ok(ctx, nb);
  PrezeroGuard prezero_guard{ctx};
  if (uni_len > 0) GIQL_TRY(prezero_row_scratch(ctx, st, os_a, os, nb));  // (the general form sorts B twice through one set of status words)
  {
  SideChain sc(ctx, st, na, nb);
  GIQL_TRY(run_linearize(ctx, sc.stream(), *a, n_chrom, lb, sa.key[0], sa.end[0], irr_a_list, 0, 0, os_a.hist,
                         os_a.gbase));
  GIQL_TRY(run_sort_onesweep(ctx, sc.stream(), sa, (u32)na, os_a.gbase, os_a.status, false, nullptr, nullptr,
                             /*skip_digits=*/row_skip(ctx, nb)));  // locality only: k_count_rows tells irregular rows by their key
  if (uni_len > 0) {
    // fixed-length B: one sorted array (its sorted ends are its sorted starts + L)
    GIQL_TRY(run_linearize(ctx, st, *b, n_chrom, lb, sstart.key[0], nullptr, irr_b_list, 1, 0, os.hist,
                           os.gbase, nullptr, nullptr, /*skip_end=*/true));
    GIQL_TRY(run_sort_onesweep(ctx, st, sstart, (u32)nb, os.gbase, os.status, false, nullptr, nullptr,
                               /*skip_digits=*/coarse_b ? 1 : 0));
  } else {
    GIQL_TRY(run_linearize(ctx, st, *b, n_chrom, lb, sstart.key[0], send.key[0], irr_b_list, 1, 0,
                           os.hist, os.gbase, os.hist_e, os.gbase_e));
    GIQL_TRY(run_sort_onesweep(ctx, st, sstart, (u32)nb, os.gbase, os.status));
    GIQL_TRY(run_sort_onesweep(ctx, st, send, (u32)nb, os.gbase_e, os.status));
  }
  GIQL_TRY(sc.join());
  }
  {
    Phase ph(ctx, st, GIQL_PH_COUNT);
    hipLaunchKernelGGL(k_count_rows, dim3(cdiv(na, 256)), dim3(256), 0, st, sa.key[0], sa.end[0],
                       sa.rid[0], (u32)na, view_of(*a), view_of(*b), sstart.key[0], send.key[0], (u32)nb,
                       irr_b_list, ctx->d_meta, counts_out, uni_len, coarse_b ? 1 : 0);
    GIQL_TRY(post_launch("count rows"));
  }
  GIQL_TRY(read_meta(ctx, st));
  if (!row_form_settled(ctx, uni_len, speculated))  // B is not fixed-length after all
    return giql_hip_count_dev_impl(ctx, a, b, n_chrom, counts_out, stream);
  ctx->stats.reserved = (uni_len > 0 ? 1 : 0) | (coarse_b ? 0x10 : 0);
  ctx->stats.n_irregular_a = ctx->h_meta->irr_a;
  ctx->stats.n_irregular_b = ctx->h_meta->irr_b;
  if (ctx->h_meta->irr_a > 0) {
    Phase ph(ctx, st, GIQL_PH_IRREGULAR);
    hipLaunchKernelGGL(k_count_irregular, dim3(ctx->h_meta->irr_a), dim3(256), 0, st, view_of(*a),
                       view_of(*b), irr_a_list, counts_out);
    GIQL_TRY(post_launch("count irregular"));
    HIP_TRY(hipStreamSynchronize(st));
  }
  collect_spans(ctx);
  ctx->stats.n_out = a->n;
  ctx->stats.span = (int64_t)ctx->h_meta->total_span;
  ctx->last_span = ctx->h_meta->total_span;
  return GIQL_OK;
}

int giql_hip_count_dev(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                       int64_t* counts_out, void* stream) {
  return with_order_fallback(ctx, [&] { return giql_hip_count_dev_impl(ctx, a, b, n_chrom, counts_out, stream); });
}

// ----------------------------------------------------------------- NEAREST
// out32 (giql_hip_nearest32_dev): [n_a] {idx_b, distance} int32 records instead of the two arrays
static int giql_hip_nearest_dev_impl(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                         int is_signed, int64_t max_distance, int32_t* idx_b_out, int64_t* dist_out,
                         void* stream, int32_t* out32 = nullptr) {
  if (!ctx) return set_err(GIQL_ERR_INVALID, "ctx is NULL");
  GIQL_TRY(check_side(a, "a"));
  GIQL_TRY(check_side(b, "b"));
  if (n_chrom < 0) return set_err(GIQL_ERR_INVALID, "n_chrom < 0");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  ctx->planned = false;
  ctx->prezeroed = false;  // (a repeated call -- a guess that missed -- starts over: the caller's frame may still hold its guard)
  reset_stats(ctx);
  ctx->stats.n_a = a->n;
  ctx->stats.n_b = b->n;
  if (a->n == 0) return GIQL_OK;
  if (!out32 && (!idx_b_out || !dist_out)) return set_err(GIQL_ERR_INVALID, "idx_b_out/dist_out is NULL");
  if (out32 && ((uintptr_t)out32 & 7)) return set_err(GIQL_ERR_INVALID, "idx_dist_out must be 8-byte aligned");
  const size_t na = (size_t)a->n, nb = (size_t)b->n;
  if (na > OS_MAX_ROWS || nb > OS_MAX_ROWS) return set_err(GIQL_ERR_INVALID, "side larger than 2^30 rows");
  if (nb == 0 || n_chrom == 0) {
    if (out32) {
      hipLaunchKernelGGL(k_nearest32_none, dim3(cdiv(na, 256)), dim3(256), 0, st, reinterpret_cast<int2*>(out32), (u32)na);
      return post_launch("nearest32 (no target)");
    }
    HIP_TRY(hipMemsetAsync(idx_b_out, 0xFF, na * sizeof(int32_t), st));
    HIP_TRY(hipMemsetAsync(dist_out, 0, na * sizeof(int64_t), st));
    return GIQL_OK;
  }
  LinBufs lb;
  SortBufs sa, sbb;
  OsScratch os, os_a;
  u32 *pmax = nullptr, *bmax = nullptr, *dummy_irr = nullptr, *chrom_lo = nullptr;
  NearestRec* recs = nullptr;
  auto carve = [&](char* base) {
    Carver c{base};
    common_sizes(c, n_chrom, lb);
    sort_sizes(c, na, sa, true);
    sort_sizes(c, nb, sbb, true);
    os_scratch_sizes(c, na, os_a);   // A's chain may run beside B's (SideChain)
    lb.top_partial = c.take<u32>((size_t)LIN_HIST_REPLICAS * MM_TOP_WORDS);   // (inside the range prezero_row_scratch zeroes)
    lb.top_partial2 = c.take<u32>((size_t)LIN_HIST_REPLICAS * MM_TOP_WORDS);
    lb.abase = c.take<u32>(MM_HIST_CHROMS);
    os_scratch_sizes(c, nb, os);     // (right behind A's: prezero_row_scratch)
    recs = c.take<NearestRec>(out32 ? 0 : na);   // (the 32-bit form scatters its records straight into the output)
    pmax = c.take<u32>(nb);
    bmax = c.take<u32>(cdiv(nb, PM_TILE) + 1);
    chrom_lo = c.take<u32>((size_t)n_chrom + 2);
    dummy_irr = c.take<u32>(16);
    return c.off;
  };
  GIQL_TRY(ensure_arena(ctx, carve(nullptr), st));
  carve(ctx->arena);

  const bool two_sorts = ctx->nearest_two_sorts;
  PrezeroGuard prezero_guard{ctx};
  if (!two_sorts) GIQL_TRY(prezero_row_scratch(ctx, st, os_a, os, nb));  // (the two-sort plan reuses B's status words)
  // Both sides sorted from their raw columns (round 3): with every chromosome base a multiple of 2^24 the span pass
  // counts the digits of the keys itself (k_chrom_minmax<1>, "Histogram in the span pass"), and the first sort pass
  // of each side builds key and end key from (chrom, start, end) -- no linearize pass (2 x 49 us at 10M x 10M against
  // ~2 x 22 more in the span pass and the first sort pass).  Every NEAREST row keeps its real key, zero-length rows
  // included, so the only guess is the layout: taken when the previous call's data took it, validated at the
  // read-back; a first call probes it (digits counted for B only) and sorts the ordinary way.
  const bool hist_ok = !two_sorts && ctx->prezeroed && !ctx->no_span_hist && ctx->os_variant == 0 &&
                       n_chrom <= MM_HIST_CHROMS && !sort_is_local(ctx, na) && !sort_is_local(ctx, nb);
  const bool keygen = hist_ok && ctx->nearest_aligned == 1;
  const bool probe = hist_ok && ctx->nearest_aligned == -1;
  if (keygen) lb.hist_partial2 = os_a.hist;
  if (!(keygen || probe)) lb.abase = nullptr;  // (run_spans: no digit counting)
  GIQL_TRY(run_spans(ctx, st, *a, *b, n_chrom, lb, (keygen || probe) ? 1 : -1, os.hist));
  if (keygen) {
    Phase ph(ctx, st, GIQL_PH_LINEARIZE, 2);
    hipLaunchKernelGGL(k_fold_top2, dim3(MM_HIST_CHROMS, 2), dim3(256), 0, st, lb.top_partial, lb.top_partial2, lb.abase,
                       os.hist, os_a.hist);
    hipLaunchKernelGGL(k_digit_offsets2, dim3(4, 2), dim3(256), 0, st, os.hist, os_a.hist, (u32)LIN_HIST_REPLICAS,
                       os.gbase, os_a.gbase);
    GIQL_TRY(post_launch("digit offsets (span histogram, NEAREST)"));
  }
  SideChain sc(ctx, st, na, nb);
  if (!keygen)
    GIQL_TRY(run_linearize(ctx, sc.stream(), *a, n_chrom, lb, sa.key[0], sa.end[0], dummy_irr + 8, 0, 1, os_a.hist,
                           os_a.gbase));
  GIQL_TRY(run_sort_onesweep(ctx, sc.stream(), sa, (u32)na, os_a.gbase, os_a.status, false, keygen ? a : nullptr,
                             lb.abase, /*skip_digits=*/row_skip(ctx, nb)));  // the query side's order only serves locality; every row keeps its real key here
  if (!keygen)
    GIQL_TRY(run_linearize(ctx, st, *b, n_chrom, lb, sbb.key[0], sbb.end[0], dummy_irr, 1, 1, os.hist,
                           os.gbase, two_sorts ? os.hist_e : nullptr, two_sorts ? os.gbase_e : nullptr));
  // (an inverted B row -- NEAREST needs start <= end on both sides -- is reported by the span pass: DevMeta::inverted_b)
  if (two_sorts) {
    // (start, end) lexicographic order = stable sort by end, then stable sort by start
    SortBufs by_end = sbb;
    for (int k = 0; k < 2; k++) {
      by_end.key[k] = sbb.end[k];
      by_end.end[k] = sbb.key[k];
    }
    GIQL_TRY(run_sort_onesweep(ctx, st, by_end, (u32)nb, os.gbase_e, os.status));
    adopt_by_end(sbb, by_end);
    GIQL_TRY(run_sort_onesweep(ctx, st, sbb, (u32)nb, os.gbase, os.status, /*keep_rids=*/true));
  } else {
    // one sort by start; the (short) runs of equal starts are ordered by end in place
    GIQL_TRY(run_sort_onesweep(ctx, st, sbb, (u32)nb, os.gbase, os.status, false, keygen ? b : nullptr, lb.abase));
    Phase ph(ctx, st, GIQL_PH_AUX);
    hipLaunchKernelGGL(k_fix_start_ties, dim3(cdiv(nb, 256)), dim3(256), 0, st, sbb.key[0], sbb.end[0],
                       sbb.rid[0], (u32)nb, ctx->d_meta);
  }
  GIQL_TRY(run_pmax(ctx, st, sbb.end[0], (u32)nb, pmax, bmax));
  GIQL_TRY(sc.join());
  {
    Phase ph(ctx, st, GIQL_PH_COUNT, 3);
    hipLaunchKernelGGL(k_chrom_bounds, dim3(cdiv((u64)n_chrom + 1, 256)), dim3(256), 0, st,
                       lb.chrom_first, n_chrom, sbb.key[0], (u32)nb, chrom_lo);
    if (out32) {
      hipLaunchKernelGGL(k_nearest<true>, dim3(cdiv(na, NR_TQ)), dim3(NR_NT), 0, st, sa.key[0], sa.end[0], sa.rid[0],
                         (u32)na, n_chrom, lb.chrom_first, chrom_lo, sbb.key[0], pmax, sbb.rid[0], (u32)nb,
                         is_signed, (i64)max_distance, (NearestRec*)nullptr, ctx->d_meta, reinterpret_cast<int2*>(out32));
    } else {
      hipLaunchKernelGGL(k_nearest<false>, dim3(cdiv(na, NR_TQ)), dim3(NR_NT), 0, st, sa.key[0], sa.end[0], sa.rid[0],
                         (u32)na, n_chrom, lb.chrom_first, chrom_lo, sbb.key[0], pmax, sbb.rid[0], (u32)nb,
                         is_signed, (i64)max_distance, recs, ctx->d_meta, (int2*)nullptr);
      hipLaunchKernelGGL(k_nearest_unpack, dim3(cdiv(na, 256)), dim3(256), 0, st, recs, (u32)na, idx_b_out,
                         (i64*)dist_out);
    }
    GIQL_TRY(post_launch("nearest"));
  }
  GIQL_TRY(read_meta(ctx, st));
  if (keygen || probe) {
    const bool aligned_now = ctx->h_meta->aligned_ok != 0;
    ctx->nearest_aligned = aligned_now ? 1 : 0;
    if (keygen && !aligned_now)  // the layout did not hold for this data: its keys were built on bases that overlap
      return giql_hip_nearest_dev_impl(ctx, a, b, n_chrom, is_signed, max_distance, idx_b_out, dist_out, stream, out32);
  }
  if (ctx->h_meta->inverted_b) return set_err(GIQL_ERR_INVALID, "NEAREST: a target row has end < start");
  if (!two_sorts && ctx->h_meta->aux0 != 0) {
    // a long run of equal starts (pile-ups): this table wants the two-sort plan
    ctx->nearest_two_sorts = true;
    return giql_hip_nearest_dev_impl(ctx, a, b, n_chrom, is_signed, max_distance, idx_b_out, dist_out, stream, out32);
  }
  if (out32 && ctx->h_meta->aux1 != 0)
    return set_err(GIQL_ERR_INVALID, "NEAREST: a distance does not fit int32; use giql_hip_nearest_dev (int64 distances)");
  collect_spans(ctx);
  ctx->stats.n_out = a->n;
  ctx->stats.span = (int64_t)ctx->h_meta->total_span;
  ctx->last_span = ctx->h_meta->total_span;
  return GIQL_OK;
}

int giql_hip_nearest_dev(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                         int is_signed, int64_t max_distance, int32_t* idx_b_out, int64_t* dist_out,
                         void* stream) {
  return with_order_fallback(ctx, [&] { return giql_hip_nearest_dev_impl(ctx, a, b, n_chrom, is_signed, max_distance, idx_b_out, dist_out, stream); });
}


int giql_hip_nearest32_dev(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom, int is_signed,
                           int64_t max_distance, int32_t* idx_dist_out, void* stream) {
  if (!idx_dist_out && a && a->n > 0) return set_err(GIQL_ERR_INVALID, "idx_dist_out is NULL");
  return with_order_fallback(ctx, [&] { return giql_hip_nearest_dev_impl(ctx, a, b, n_chrom, is_signed, max_distance, nullptr, nullptr, stream, idx_dist_out); });
}

// ------------------------------------------------------------ NEAREST k > 1
static int giql_hip_nearest_k_dev_impl(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                                       int32_t k, int is_signed, int64_t max_distance, int32_t* idx_b_out,
                                       int64_t* dist_out, void* stream) {
  if (!ctx) return set_err(GIQL_ERR_INVALID, "ctx is NULL");
  GIQL_TRY(check_side(a, "a"));
  GIQL_TRY(check_side(b, "b"));
  if (n_chrom < 0) return set_err(GIQL_ERR_INVALID, "n_chrom < 0");
  if (k < 1 || k > NEAREST_K_MAX) return set_err(GIQL_ERR_INVALID, "k=%d outside [1, %d]", k, NEAREST_K_MAX);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  ctx->planned = false;
  reset_stats(ctx);
  ctx->stats.n_a = a->n;
  ctx->stats.n_b = b->n;
  if (a->n == 0) return GIQL_OK;
  if (!idx_b_out || !dist_out) return set_err(GIQL_ERR_INVALID, "idx_b_out/dist_out is NULL");
  const size_t na = (size_t)a->n, nb = (size_t)b->n;
  if (na > OS_MAX_ROWS || nb > OS_MAX_ROWS) return set_err(GIQL_ERR_INVALID, "side larger than 2^30 rows");
  if (na * (size_t)k > 0x7FFFFFF0ull) return set_err(GIQL_ERR_INVALID, "n_a * k does not fit 31 bits");
  if (nb == 0 || n_chrom == 0) {
    HIP_TRY(hipMemsetAsync(idx_b_out, 0xFF, na * k * sizeof(int32_t), st));
    HIP_TRY(hipMemsetAsync(dist_out, 0, na * k * sizeof(int64_t), st));
    return GIQL_OK;
  }
  LinBufs lb;
  SortBufs sa, sbb, se;
  OsScratch os;
  u32 *pmax = nullptr, *bmax = nullptr, *dummy_irr = nullptr, *chrom_lo = nullptr, *chrom_lo_e = nullptr;
  NearestRec* recs = nullptr;
  auto carve = [&](char* base) {
    Carver c{base};
    common_sizes(c, n_chrom, lb);
    sort_sizes(c, na, sa, true);
    sort_sizes(c, nb, sbb, true);
    sort_sizes(c, nb, se, true);   // the (end, start) order: key = end, "end" payload = start
    os_scratch_sizes(c, na > nb ? na : nb, os);
    recs = c.take<NearestRec>(na * (size_t)k);
    pmax = c.take<u32>(nb);
    bmax = c.take<u32>(cdiv(nb, PM_TILE) + 1);
    chrom_lo = c.take<u32>((size_t)n_chrom + 2);
    chrom_lo_e = c.take<u32>((size_t)n_chrom + 2);
    dummy_irr = c.take<u32>(16);
    return c.off;
  };
  GIQL_TRY(ensure_arena(ctx, carve(nullptr), st));
  carve(ctx->arena);

  GIQL_TRY(run_spans(ctx, st, *a, *b, n_chrom, lb));
  // Two sorted views of B from ONE linearize pass (it counts the digits of the starts and of the ends):
  //   by (start, end) and by (end, start).
  GIQL_TRY(run_linearize(ctx, st, *b, n_chrom, lb, sbb.key[0], sbb.end[0], dummy_irr, 1, 1, os.hist, os.gbase,
                         os.hist_e, os.gbase_e));
  const bool two_sorts = ctx->nearest_two_sorts;
  if (two_sorts) {
    // tables with long runs of equal starts or ends: stable two-key sorts -- (start, end) = by end, then stably by
    // start; (end, start) = that order stably re-sorted by end
    {
      SortBufs by_end = sbb;
      for (int i = 0; i < 2; i++) {
        by_end.key[i] = sbb.end[i];
        by_end.end[i] = sbb.key[i];
      }
      GIQL_TRY(run_sort_onesweep(ctx, st, by_end, (u32)nb, os.gbase_e, os.status));
      adopt_by_end(sbb, by_end);
      GIQL_TRY(run_sort_onesweep(ctx, st, sbb, (u32)nb, os.gbase, os.status, /*keep_rids=*/true));
    }
    HIP_TRY(hipMemcpyAsync(se.key[0], sbb.end[0], nb * sizeof(u32), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(se.end[0], sbb.key[0], nb * sizeof(u32), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(se.rid[0], sbb.rid[0], nb * sizeof(u32), hipMemcpyDeviceToDevice, st));
    GIQL_TRY(run_sort_onesweep(ctx, st, se, (u32)nb, os.gbase_e, os.status, /*keep_rids=*/true));
  } else {
    // ONE sort per view (eight passes instead of twelve): each view sorted on its first key from the linearized
    // columns, the -- short -- runs of equal first keys ordered by the second key in place (k_fix_start_ties; a run
    // too long for that flags the table for the two-key plan above, as in NEAREST k = 1)
    HIP_TRY(hipMemcpyAsync(se.key[0], sbb.end[0], nb * sizeof(u32), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(se.end[0], sbb.key[0], nb * sizeof(u32), hipMemcpyDeviceToDevice, st));
    GIQL_TRY(run_sort_onesweep(ctx, st, sbb, (u32)nb, os.gbase, os.status));
    GIQL_TRY(run_sort_onesweep(ctx, st, se, (u32)nb, os.gbase_e, os.status));
    Phase ph(ctx, st, GIQL_PH_AUX, 2);
    hipLaunchKernelGGL(k_fix_start_ties, dim3(cdiv(nb, 256)), dim3(256), 0, st, sbb.key[0], sbb.end[0], sbb.rid[0],
                       (u32)nb, ctx->d_meta);
    hipLaunchKernelGGL(k_fix_start_ties, dim3(cdiv(nb, 256)), dim3(256), 0, st, se.key[0], se.end[0], se.rid[0],
                       (u32)nb, ctx->d_meta);
  }
  GIQL_TRY(run_pmax(ctx, st, sbb.end[0], (u32)nb, pmax, bmax));
  GIQL_TRY(run_linearize(ctx, st, *a, n_chrom, lb, sa.key[0], sa.end[0], dummy_irr, 0, 1, os.hist, os.gbase));
  GIQL_TRY(run_sort_onesweep(ctx, st, sa, (u32)na, os.gbase, os.status, false, nullptr, nullptr,
                             /*skip_digits=*/row_skip(ctx, nb)));  // the query side's order only serves locality; every row keeps its real key here
  {
    Phase ph(ctx, st, GIQL_PH_COUNT, 4);
    hipLaunchKernelGGL(k_chrom_bounds, dim3(cdiv((u64)n_chrom + 1, 256)), dim3(256), 0, st, lb.chrom_first, n_chrom,
                       sbb.key[0], (u32)nb, chrom_lo);
    hipLaunchKernelGGL(k_chrom_bounds, dim3(cdiv((u64)n_chrom + 1, 256)), dim3(256), 0, st, lb.chrom_first, n_chrom,
                       se.key[0], (u32)nb, chrom_lo_e);
    if (k >= 16) {  // a row's k ids / distances are contiguous runs of 64+ / 128+ bytes: straight into the outputs
                     // (at k = 8 the two scattered 32 / 64-byte runs cost what the 128-byte records + the unpack pass do)
      hipLaunchKernelGGL(k_nearest_k<true>, dim3(cdiv(na, NR_TQ)), dim3(NR_NT), 0, st, sa.key[0], sa.end[0], sa.rid[0],
                         (u32)na, n_chrom, lb.chrom_first, chrom_lo, chrom_lo_e, sbb.key[0], sbb.end[0], pmax,
                         sbb.rid[0], se.key[0], se.end[0], se.rid[0], (u32)nb, (int)k, is_signed, (i64)max_distance,
                         recs, idx_b_out, (i64*)dist_out, ctx->d_meta);
    } else {
      hipLaunchKernelGGL(k_nearest_k<false>, dim3(cdiv(na, NR_TQ)), dim3(NR_NT), 0, st, sa.key[0], sa.end[0], sa.rid[0],
                         (u32)na, n_chrom, lb.chrom_first, chrom_lo, chrom_lo_e, sbb.key[0], sbb.end[0], pmax,
                         sbb.rid[0], se.key[0], se.end[0], se.rid[0], (u32)nb, (int)k, is_signed, (i64)max_distance,
                         recs, idx_b_out, (i64*)dist_out, ctx->d_meta);
      hipLaunchKernelGGL(k_nearest_unpack, dim3(cdiv(na * (size_t)k, 256)), dim3(256), 0, st, recs,
                         (u32)(na * (size_t)k), idx_b_out, (i64*)dist_out);
    }
    GIQL_TRY(post_launch("nearest k"));
  }
  GIQL_TRY(read_meta(ctx, st));
  if (!two_sorts && ctx->h_meta->aux0 != 0) {
    // a long run of equal starts or ends (pile-ups): this table wants the two-key sorts
    ctx->nearest_two_sorts = true;
    return giql_hip_nearest_k_dev_impl(ctx, a, b, n_chrom, k, is_signed, max_distance, idx_b_out, dist_out, stream);
  }
  if (ctx->h_meta->inverted_b) return set_err(GIQL_ERR_INVALID, "NEAREST: a target row has end < start");
  collect_spans(ctx);
  ctx->stats.n_out = a->n * (int64_t)k;
  ctx->stats.span = (int64_t)ctx->h_meta->total_span;
  ctx->last_span = ctx->h_meta->total_span;
  return GIQL_OK;
}

int giql_hip_nearest_k_dev(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom, int32_t k,
                           int is_signed, int64_t max_distance, int32_t* idx_b_out, int64_t* dist_out,
                           void* stream) {
  return with_order_fallback(ctx, [&] { return giql_hip_nearest_k_dev_impl(ctx, a, b, n_chrom, k, is_signed, max_distance, idx_b_out, dist_out, stream); });
}

// ---------------------------------------------------------- CLUSTER / MERGE
// Shared front half: keys + ends on the linear axis, sorted by start, inclusive
// prefix max of the ends, new-cluster flags and their exclusive scan.
struct ClusterBufs {
  LinBufs lb;
  SortBufs sb;
  OsScratch os;
  u32 *pmax = nullptr, *bmax = nullptr, *chrom_lo = nullptr, *flags = nullptr, *excl = nullptr;
  u32* dummy_irr = nullptr;
  u64 *bsums = nullptr, *total = nullptr;
  u32* head_pos = nullptr;
  u32* seg_end = nullptr;  // MERGE under a predicate: MAX(end) per region
};

static int cluster_front(giql_hip_ctx* ctx, hipStream_t st, const giql_side* s, int32_t n_chrom,
                         int64_t distance, bool want_rids, bool want_heads, ClusterBufs& cb,
                         const DevPreds* preds = nullptr) {
  const size_t n = (size_t)s->n;
  auto carve = [&](char* base) {
    Carver c{base};
    common_sizes(c, n_chrom, cb.lb);
    for (int k = 0; k < 2; k++) {
      cb.sb.key[k] = c.take<u32>(n);
      cb.sb.end[k] = c.take<u32>(n);
      cb.sb.rid[k] = want_rids ? c.take<u32>(n) : nullptr;
    }
    os_scratch_sizes(c, n, cb.os);
    cb.pmax = c.take<u32>(n);
    cb.bmax = c.take<u32>(cdiv(n, PM_TILE) + 1);
    cb.chrom_lo = c.take<u32>((size_t)n_chrom + 2);
    cb.flags = c.take<u32>(n);
    cb.excl = c.take<u32>(n + 1);
    cb.bsums = c.take<u64>(cdiv((u64)n, SCAN_TILE) + 1);
    cb.total = c.take<u64>(1);
    cb.dummy_irr = c.take<u32>(16);
    cb.head_pos = want_heads ? c.take<u32>(n + 1) : nullptr;
    cb.seg_end = want_heads && preds && preds->n > 0 ? c.take<u32>(n + 1) : nullptr;
    return c.off;
  };
  GIQL_TRY(ensure_arena(ctx, carve(nullptr), st));
  carve(ctx->arena);
  giql_side none;
  memset(&none, 0, sizeof(none));
  GIQL_TRY(run_spans(ctx, st, *s, none, n_chrom, cb.lb));
  GIQL_TRY(run_linearize(ctx, st, *s, n_chrom, cb.lb, cb.sb.key[0], cb.sb.end[0], cb.dummy_irr, 0, 1,
                         cb.os.hist, cb.os.gbase));
  {
    Phase ph(ctx, st, GIQL_PH_AUX);
    hipLaunchKernelGGL(k_check_not_inverted, dim3(cdiv(n, 256)), dim3(256), 0, st, view_of(*s),
                       ctx->d_meta);
  }
  GIQL_TRY(run_sort_onesweep(ctx, st, cb.sb, (u32)n, cb.os.gbase, cb.os.status));
  GIQL_TRY(run_pmax(ctx, st, cb.sb.end[0], (u32)n, cb.pmax, cb.bmax));
  HIP_TRY(hipMemsetAsync(cb.flags, 0, n * sizeof(u32), st));
  {
    Phase ph(ctx, st, GIQL_PH_COUNT, 3);
    hipLaunchKernelGGL(k_chrom_bounds, dim3(cdiv((u64)n_chrom + 1, 256)), dim3(256), 0, st,
                       cb.lb.chrom_first, n_chrom, cb.sb.key[0], (u32)n, cb.chrom_lo);
    hipLaunchKernelGGL(k_cluster_firsts, dim3(cdiv((u64)n_chrom, 256)), dim3(256), 0, st, cb.chrom_lo,
                       n_chrom, (u32)n, cb.flags);
    hipLaunchKernelGGL(k_cluster_flags, dim3(cdiv(n, 256)), dim3(256), 0, st, cb.sb.key[0], cb.pmax,
                       (u32)n, (u64)(distance > 0 ? distance : 0), cb.flags);
    if (preds && preds->n > 0)  // (needs the row ids: want_rids)
      hipLaunchKernelGGL(k_cluster_pred_flags, dim3(cdiv(n, 256)), dim3(256), 0, st, cb.sb.rid[0], (u32)n, *preds,
                         cb.flags);
    GIQL_TRY(post_launch("cluster flags"));
  }
  GIQL_TRY(run_scan<u32>(ctx, st, GIQL_PH_SCAN, cb.flags, (u64)n, cb.excl, cb.bsums, cb.total));
  return GIQL_OK;
}

static int check_cluster_args(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom) {
  if (!ctx) return set_err(GIQL_ERR_INVALID, "ctx is NULL");
  GIQL_TRY(check_side(s, "s"));
  if (n_chrom < 0) return set_err(GIQL_ERR_INVALID, "n_chrom < 0");
  if (s->start_off != 0 || s->end_off != 0)
    return set_err(GIQL_ERR_INVALID,
                   "CLUSTER / MERGE read the raw start / end columns (cluster.py:246-263): offsets must be 0");
  if ((size_t)s->n > OS_MAX_ROWS) return set_err(GIQL_ERR_INVALID, "side larger than 2^30 rows");
  return GIQL_OK;
}

static int cluster_status(giql_hip_ctx* ctx, hipStream_t st) {
  HIP_TRY(hipMemcpyAsync(ctx->h_meta, ctx->d_meta, sizeof(DevMeta), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (ctx->h_meta->status == -1)
    return set_err(GIQL_ERR_INVALID, "CLUSTER / MERGE need start <= end on every row");
  return read_meta(ctx, st);
}

static int giql_hip_cluster_dev_impl(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom, int64_t distance,
                         int64_t* cluster_id_out, void* stream, const DevPreds* preds = nullptr) {
  GIQL_TRY(check_cluster_args(ctx, s, n_chrom));
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  ctx->planned = false;
  reset_stats(ctx);
  ctx->stats.n_a = s->n;
  if (s->n == 0) return GIQL_OK;
  if (!cluster_id_out) return set_err(GIQL_ERR_INVALID, "cluster_id_out is NULL");
  if (n_chrom == 0) return set_err(GIQL_ERR_CHROM, "rows but n_chrom = 0");
  ClusterBufs cb;
  GIQL_TRY(cluster_front(ctx, st, s, n_chrom, distance, true, false, cb, preds));
  {
    Phase ph(ctx, st, GIQL_PH_FILL);
    hipLaunchKernelGGL(k_cluster_ids, dim3(cdiv((u64)s->n, 256)), dim3(256), 0, st, cb.sb.key[0],
                       cb.sb.rid[0], cb.flags, cb.excl, (u32)s->n, cb.lb.chrom_first, n_chrom,
                       cb.chrom_lo, (i64*)cluster_id_out);
    GIQL_TRY(post_launch("cluster ids"));
  }
  GIQL_TRY(cluster_status(ctx, st));
  collect_spans(ctx);
  ctx->stats.n_out = s->n;
  ctx->stats.span = (int64_t)ctx->h_meta->total_span;
  return GIQL_OK;
}

int giql_hip_cluster_dev(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom, int64_t distance,
                         int64_t* cluster_id_out, void* stream) {
  return with_order_fallback(ctx, [&] { return giql_hip_cluster_dev_impl(ctx, s, n_chrom, distance, cluster_id_out, stream); });
}

static int check_operand(const giql_operand& o, int k, const char* which);

// giql_pred[] -> the by-value kernel argument; uses[side] = a column of that side is read
// A well-formed postfix program: every operator finds its arguments, one value is left, the stack stays within
// SEL_X_STACK.  uses[side] as in convert_preds.
static int check_program(const giql_operand* nodes, int32_t n_nodes, int first, int count, int k, const char* which,
                         bool* uses) {
  if (!nodes || first < 0 || count < 1 || first + count > n_nodes)
    return set_err(GIQL_ERR_INVALID, "predicate %d %s: expression nodes [%d, %d) outside the %d given", k, which, first,
                   first + count, n_nodes);
  int sp = 0;
  for (int t = first; t < first + count; t++) {
    const giql_operand& nd = nodes[t];
    if (nd.side == GIQL_SIDE_A || nd.side == GIQL_SIDE_B || nd.side == GIQL_SIDE_LIT) {
      GIQL_TRY(check_operand(nd, k, which));
      if (uses && nd.side != GIQL_SIDE_LIT) uses[nd.side] = true;
      if (++sp > SEL_X_STACK) return set_err(GIQL_ERR_INVALID, "predicate %d %s: expression deeper than %d", k, which, SEL_X_STACK);
    } else if (nd.side == GIQL_X_NEG || nd.side == GIQL_X_ABS || nd.side == GIQL_X_ISNULL || nd.side == GIQL_X_NOTNULL ||
               nd.side == GIQL_X_NOT) {
      if (sp < 1) return set_err(GIQL_ERR_INVALID, "predicate %d %s: malformed expression", k, which);
    } else if ((nd.side >= GIQL_X_ADD && nd.side <= GIQL_X_GREATEST) || (nd.side >= GIQL_X_EQ && nd.side <= GIQL_X_GE) ||
               nd.side == GIQL_X_AND || nd.side == GIQL_X_OR) {
      if (sp < 2) return set_err(GIQL_ERR_INVALID, "predicate %d %s: malformed expression", k, which);
      sp--;
    } else {
      return set_err(GIQL_ERR_INVALID, "predicate %d %s: expression node kind %d", k, which, nd.side);
    }
  }
  if (sp != 1) return set_err(GIQL_ERR_INVALID, "predicate %d %s: malformed expression", k, which);
  return GIQL_OK;
}

static int convert_preds(const giql_pred* preds, int32_t n_preds, DevPreds& ps, bool* uses,
                         const giql_operand* nodes = nullptr, int32_t n_nodes = 0, bool* any_expr = nullptr) {
  static_assert(sizeof(giql_operand) == sizeof(DevOperand), "giql_operand layout");
  static_assert(sizeof(giql_pred) == sizeof(DevPred), "giql_pred layout");
  memset(&ps, 0, sizeof(ps));
  ps.n = n_preds;
  for (int k = 0; k < n_preds; k++) {
    const bool unary = preds[k].op == GIQL_OP_IS_NULL || preds[k].op == GIQL_OP_NOT_NULL || preds[k].op == GIQL_OP_IS_TRUE;
    if (preds[k].op < GIQL_OP_EQ || preds[k].op > GIQL_OP_IS_TRUE)
      return set_err(GIQL_ERR_INVALID, "predicate %d: operator %d", k, preds[k].op);
    if (preds[k].group < 0) return set_err(GIQL_ERR_INVALID, "predicate %d: group %d", k, preds[k].group);
    for (int w = 0; w < (unary ? 1 : 2); w++) {
      const giql_operand& o = w ? preds[k].rhs : preds[k].lhs;
      if (o.side == GIQL_SIDE_EXPR) {
        GIQL_TRY(check_program(nodes, n_nodes, (int)o.lit_i, o.type, k, w ? "rhs" : "lhs", uses));
        if (any_expr) *any_expr = true;
      } else {
        GIQL_TRY(check_operand(o, k, w ? "rhs" : "lhs"));
        if (uses && o.side != GIQL_SIDE_LIT) uses[o.side] = true;
      }
    }
    memcpy(&ps.p[k], &preds[k], sizeof(DevPred));
    if (unary) {  // the right operand is not read: make it a harmless literal
      memset(&ps.p[k].rhs, 0, sizeof(DevOperand));
      ps.p[k].rhs.side = GIQL_SIDE_LIT;
    }
  }
  return GIQL_OK;
}

int giql_hip_cluster_pred_dev(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom, int64_t distance,
                              const giql_pred* preds, int32_t n_preds, int64_t* cluster_id_out, void* stream) {
  if (n_preds < 0 || n_preds > SEL_MAX_PREDS || (n_preds && !preds))
    return set_err(GIQL_ERR_INVALID, "bad predicates (at most %d)", SEL_MAX_PREDS);
  DevPreds ps;
  GIQL_TRY(convert_preds(preds, n_preds, ps, nullptr));
  return with_order_fallback(ctx, [&] { return giql_hip_cluster_dev_impl(ctx, s, n_chrom, distance, cluster_id_out, stream, &ps); });
}

static int giql_hip_merge_dev_impl(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom, int64_t distance,
                       int32_t* out_chrom, int32_t* out_start, int32_t* out_end,
                       int64_t* out_count, int64_t capacity, int64_t* n_out, void* stream,
                       const DevPreds* preds = nullptr) {
  GIQL_TRY(check_cluster_args(ctx, s, n_chrom));
  if (!n_out) return set_err(GIQL_ERR_INVALID, "n_out is NULL");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  ctx->planned = false;
  reset_stats(ctx);
  ctx->stats.n_a = s->n;
  *n_out = 0;
  if (s->n == 0) return GIQL_OK;
  if (!out_chrom || !out_start || !out_end) return set_err(GIQL_ERR_INVALID, "output buffer is NULL");
  if (n_chrom == 0) return set_err(GIQL_ERR_CHROM, "rows but n_chrom = 0");
  ClusterBufs cb;
  const bool pred = preds && preds->n > 0;
  GIQL_TRY(cluster_front(ctx, st, s, n_chrom, distance, pred, true, cb, preds));
  u64 h_total = 0;
  HIP_TRY(hipMemcpyAsync(&h_total, cb.total, sizeof(u64), hipMemcpyDeviceToHost, st));
  GIQL_TRY(cluster_status(ctx, st));
  if ((int64_t)h_total > capacity)
    return set_err(GIQL_ERR_CAPACITY, "%llu merged regions, capacity %lld",
                   (unsigned long long)h_total, (long long)capacity);
  {
    Phase ph(ctx, st, GIQL_PH_FILL, pred ? 3 : 2);
    hipLaunchKernelGGL(k_merge_heads, dim3(cdiv((u64)s->n, 256)), dim3(256), 0, st, cb.flags, cb.excl,
                       (u32)s->n, cb.head_pos);
    if (pred && h_total) {
      HIP_TRY(hipMemsetAsync(cb.seg_end, 0, (size_t)h_total * sizeof(u32), st));
      hipLaunchKernelGGL(k_merge_segmax, dim3(cdiv((u64)s->n, 256)), dim3(256), 0, st, cb.sb.end[0], cb.excl,
                         cb.flags, (u32)s->n, cb.seg_end);
    }
    if (h_total)
      hipLaunchKernelGGL(k_merge_rows, dim3(cdiv(h_total, 256)), dim3(256), 0, st, cb.head_pos,
                         (u32)h_total, (u32)s->n, cb.sb.key[0], cb.pmax, pred ? cb.seg_end : (u32*)nullptr, cb.lb.chrom_first,
                         cb.lb.chrom_base, n_chrom, out_chrom, out_start, out_end, (i64*)out_count);
    GIQL_TRY(post_launch("merge rows"));
  }
  HIP_TRY(hipStreamSynchronize(st));
  collect_spans(ctx);
  *n_out = (int64_t)h_total;
  ctx->stats.n_out = (int64_t)h_total;
  ctx->stats.span = (int64_t)ctx->h_meta->total_span;
  return GIQL_OK;
}

int giql_hip_merge_dev(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom, int64_t distance,
                       int32_t* out_chrom, int32_t* out_start, int32_t* out_end, int64_t* out_count,
                       int64_t capacity, int64_t* n_out, void* stream) {
  return with_order_fallback(ctx, [&] { return giql_hip_merge_dev_impl(ctx, s, n_chrom, distance, out_chrom, out_start, out_end, out_count, capacity, n_out, stream); });
}

int giql_hip_merge_pred_dev(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom, int64_t distance,
                            const giql_pred* preds, int32_t n_preds, int32_t* out_chrom, int32_t* out_start,
                            int32_t* out_end, int64_t* out_count, int64_t capacity, int64_t* n_out, void* stream) {
  if (n_preds < 0 || n_preds > SEL_MAX_PREDS || (n_preds && !preds))
    return set_err(GIQL_ERR_INVALID, "bad predicates (at most %d)", SEL_MAX_PREDS);
  DevPreds ps;
  GIQL_TRY(convert_preds(preds, n_preds, ps, nullptr));
  return with_order_fallback(ctx, [&] { return giql_hip_merge_dev_impl(ctx, s, n_chrom, distance, out_chrom, out_start, out_end, out_count, capacity, n_out, stream, &ps); });
}

// ------------------------------------------- distinct intervals + segment sums
// (the aggregate half of count_overlaps: GROUP BY the left interval, SUM of the per-row
// counts -- src/giql/expanders/intersects_duckdb.py:806-854)
static int giql_hip_group_rows_dev_impl(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom,
                                        int32_t* group_of_row, int32_t* rep_row, int64_t* n_groups,
                                        void* stream) {
  if (!ctx || !n_groups) return set_err(GIQL_ERR_INVALID, "ctx/n_groups is NULL");
  GIQL_TRY(check_side(s, "s"));
  if (n_chrom < 0) return set_err(GIQL_ERR_INVALID, "n_chrom < 0");
  if ((size_t)s->n > OS_MAX_ROWS) return set_err(GIQL_ERR_INVALID, "side larger than 2^30 rows");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  ctx->planned = false;
  reset_stats(ctx);
  ctx->stats.n_a = s->n;
  *n_groups = 0;
  if (s->n == 0) return GIQL_OK;
  if (!group_of_row || !rep_row) return set_err(GIQL_ERR_INVALID, "output buffer is NULL");
  if (n_chrom == 0) return set_err(GIQL_ERR_CHROM, "rows but n_chrom = 0");
  const size_t n = (size_t)s->n;
  LinBufs lb;
  SortBufs sb;
  OsScratch os;
  u32 *flags = nullptr, *excl = nullptr, *dummy_irr = nullptr;
  u64 *bsums = nullptr, *total = nullptr;
  auto carve = [&](char* base) {
    Carver c{base};
    common_sizes(c, n_chrom, lb);
    sort_sizes(c, n, sb, true);
    os_scratch_sizes(c, n, os);
    flags = c.take<u32>(n);
    excl = c.take<u32>(n + 1);
    bsums = c.take<u64>(cdiv((u64)n, SCAN_TILE) + 1);
    total = c.take<u64>(1);
    dummy_irr = c.take<u32>(16);
    return c.off;
  };
  GIQL_TRY(ensure_arena(ctx, carve(nullptr), st));
  carve(ctx->arena);
  giql_side raw = *s;  // identical RAW coordinates are what GROUP BY compares
  raw.start_off = raw.end_off = 0;
  giql_side none;
  memset(&none, 0, sizeof(none));
  const bool two_sorts = ctx->nearest_two_sorts;
  GIQL_TRY(run_spans(ctx, st, raw, none, n_chrom, lb));
  GIQL_TRY(run_linearize(ctx, st, raw, n_chrom, lb, sb.key[0], sb.end[0], dummy_irr, 0, 1, os.hist,
                         os.gbase, two_sorts ? os.hist_e : nullptr, two_sorts ? os.gbase_e : nullptr));
  if (two_sorts) {  // (start, end) order = stable sort by end, then stable sort by start
    SortBufs by_end = sb;
    for (int k = 0; k < 2; k++) {
      by_end.key[k] = sb.end[k];
      by_end.end[k] = sb.key[k];
    }
    GIQL_TRY(run_sort_onesweep(ctx, st, by_end, (u32)n, os.gbase_e, os.status));
    adopt_by_end(sb, by_end);
    GIQL_TRY(run_sort_onesweep(ctx, st, sb, (u32)n, os.gbase, os.status, /*keep_rids=*/true));
  } else {
    GIQL_TRY(run_sort_onesweep(ctx, st, sb, (u32)n, os.gbase, os.status));
    Phase ph(ctx, st, GIQL_PH_AUX);
    hipLaunchKernelGGL(k_fix_start_ties, dim3(cdiv(n, 256)), dim3(256), 0, st, sb.key[0], sb.end[0],
                       sb.rid[0], (u32)n, ctx->d_meta);
  }
  {
    Phase ph(ctx, st, GIQL_PH_COUNT);
    hipLaunchKernelGGL(k_group_flags, dim3(cdiv(n, 256)), dim3(256), 0, st, sb.key[0], sb.end[0], (u32)n,
                       flags);
    GIQL_TRY(post_launch("group flags"));
  }
  GIQL_TRY(run_scan<u32>(ctx, st, GIQL_PH_SCAN, flags, (u64)n, excl, bsums, total));
  {
    Phase ph(ctx, st, GIQL_PH_FILL);
    hipLaunchKernelGGL(k_group_ids, dim3(cdiv(n, 256)), dim3(256), 0, st, sb.rid[0], flags, excl, (u32)n,
                       group_of_row, rep_row);
    GIQL_TRY(post_launch("group ids"));
  }
  u64 h_total = 0;
  HIP_TRY(hipMemcpyAsync(&h_total, total, sizeof(u64), hipMemcpyDeviceToHost, st));
  GIQL_TRY(read_meta(ctx, st));
  if (!two_sorts && ctx->h_meta->aux0 != 0) {  // a long run of equal starts: two-sort plan
    ctx->nearest_two_sorts = true;
    return giql_hip_group_rows_dev_impl(ctx, s, n_chrom, group_of_row, rep_row, n_groups, stream);
  }
  collect_spans(ctx);
  *n_groups = (int64_t)h_total;
  ctx->stats.n_out = (int64_t)h_total;
  return GIQL_OK;
}

int giql_hip_group_rows_dev(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom, int32_t* group_of_row,
                            int32_t* rep_row, int64_t* n_groups, void* stream) {
  return with_order_fallback(ctx, [&] {
    return giql_hip_group_rows_dev_impl(ctx, s, n_chrom, group_of_row, rep_row, n_groups, stream);
  });
}

int giql_hip_segment_sum_dev(giql_hip_ctx* ctx, const int64_t* values, const int32_t* group_of_row,
                             int64_t n, int64_t* sums, int64_t n_groups, void* stream) {
  if (!ctx || n < 0 || n_groups < 0 || n_groups > 0x7FFFFFFFll) return set_err(GIQL_ERR_INVALID, "bad arguments");
  if ((n > 0 && (!values || !group_of_row)) || (n_groups > 0 && !sums))
    return set_err(GIQL_ERR_INVALID, "NULL buffer");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  reset_stats(ctx);
  if (n_groups > 0) HIP_TRY(hipMemsetAsync(sums, 0, (size_t)n_groups * sizeof(int64_t), st));
  if (n == 0) return GIQL_OK;
  HIP_TRY(hipMemsetAsync(&ctx->d_meta->status, 0, sizeof(int), st));
  u32 grid = cdiv((u64)n, 256 * 8);
  if (grid > GIQL_STREAM_GRID) grid = GIQL_STREAM_GRID;
  {
    Phase ph(ctx, st, GIQL_PH_AUX);
    hipLaunchKernelGGL(k_segment_sum, dim3(grid), dim3(256), 0, st, (const i64*)values, group_of_row, (u64)n,
                       (u32)n_groups, (unsigned long long*)sums, ctx->d_meta);
    GIQL_TRY(post_launch("segment sum"));
  }
  GIQL_TRY(read_meta(ctx, st));
  collect_spans(ctx);
  return GIQL_OK;
}

// -------------------------------------------------------------------- spans
int giql_hip_chrom_spans_dev(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b,
                             int32_t n_chrom, int64_t* spans_out, void* stream) {
  if (!ctx || !spans_out) return set_err(GIQL_ERR_INVALID, "ctx/spans_out is NULL");
  GIQL_TRY(check_side(a, "a"));
  GIQL_TRY(check_side(b, "b"));
  if (n_chrom <= 0) return set_err(GIQL_ERR_INVALID, "n_chrom <= 0");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  ctx->planned = false;
  LinBufs lb;
  auto carve = [&](char* base) {
    Carver c{base};
    common_sizes(c, n_chrom, lb);
    return c.off;
  };
  GIQL_TRY(ensure_arena(ctx, carve(nullptr), st));
  carve(ctx->arena);
  GIQL_TRY(run_spans(ctx, st, *a, *b, n_chrom, lb));
  std::vector<int> mn((size_t)n_chrom), mx((size_t)n_chrom);
  HIP_TRY(hipMemcpyAsync(mn.data(), lb.gmin, (size_t)n_chrom * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(mx.data(), lb.gmax, (size_t)n_chrom * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ctx->h_meta, ctx->d_meta, sizeof(DevMeta), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (ctx->h_meta->status == GIQL_ERR_CHROM)
    return set_err(GIQL_ERR_CHROM, "a chrom id is outside [0, n_chrom)");
  int omin = a->start_off, omax = a->start_off;
  const int offs[3] = {a->end_off, b->start_off, b->end_off};
  for (int k = 0; k < 3; k++) {
    if (offs[k] < omin) omin = offs[k];
    if (offs[k] > omax) omax = offs[k];
  }
  for (int c = 0; c < n_chrom; c++)
    spans_out[c] = mn[c] <= mx[c] ? ((int64_t)mx[c] + omax) - ((int64_t)mn[c] + omin) + 1 : 0;
  return GIQL_OK;
}

// ---------------------------------------------------------------- checksum
int giql_hip_pairs_checksum_dev(giql_hip_ctx* ctx, const int32_t* row_a, const int32_t* row_b,
                                int64_t n, void* stream, uint64_t* out) {
  if (!ctx || !out || n < 0) return set_err(GIQL_ERR_INVALID, "bad arguments");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(ctx->d_scratch64, 0, sizeof(u64), st));
  if (n > 0) {
    u32 grid = cdiv((u64)n, 256 * 8);
    if (grid > GIQL_STREAM_GRID) grid = GIQL_STREAM_GRID;
    hipLaunchKernelGGL(k_pairs_checksum, dim3(grid), dim3(256), 0, st, row_a, row_b, (u64)n,
                       ctx->d_scratch64);
    GIQL_TRY(post_launch("checksum"));
  }
  u64 h = 0;
  HIP_TRY(hipMemcpyAsync(&h, ctx->d_scratch64, sizeof(u64), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *out = h;
  return GIQL_OK;
}


// ------------------------------------------- compact plan (multi-GPU exchange)
// The uniform-length plan IS a compact description of the result: per query row its id, the
// first matching position in the other side's sorted order and the number of matches, plus
// that side's row ids in sorted order -- 12 B per query row + 4 B per row instead of 8 B per
// pair (65 MB instead of 400 MB per rank at BASELINE config 4 on 8 GPUs).  Ranks exchange THAT
// and every receiver expands it with the same partition + fill kernels the local join uses.
// (the counts come from the offsets: the fused count leaves upper bounds, not counts, in the plan's cnt array)
__global__ __launch_bounds__(256) void k_plan_export(const u32* __restrict__ q_rid, const u32* __restrict__ lo,
                                                      const u64* __restrict__ off, u32 n_q,
                                                      const u32* __restrict__ s_rid, u32 n_s, u32 q_add, u32 s_add,
                                                      int32_t* __restrict__ q_rid_out, u32* __restrict__ lo_out,
                                                      u32* __restrict__ cnt_out, int32_t* __restrict__ s_rid_out) {
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < (u64)n_q + n_s; i += stride) {
    if (i < n_q) {
      q_rid_out[i] = (int32_t)(q_rid[i] + q_add);
      lo_out[i] = lo[i];
      cnt_out[i] = (u32)(off[i + 1] - off[i]);
    } else {
      const u64 j = i - n_q;
      s_rid_out[j] = (int32_t)(s_rid[j] + s_add);
    }
  }
}

int giql_hip_inner_plan_export_dev(giql_hip_ctx* ctx, int32_t* q_rid_out, uint32_t* lo_out, uint32_t* cnt_out,
                                   int32_t* s_rid_out, int64_t q_capacity, int64_t s_capacity,
                                   int32_t rid_add_a, int32_t rid_add_b, int32_t* query_is_a, int64_t* n_q,
                                   int64_t* n_s, void* stream) {
  if (!ctx || !query_is_a || !n_q || !n_s) return set_err(GIQL_ERR_INVALID, "NULL argument");
  if (!ctx->planned) return set_err(GIQL_ERR_STATE, "plan export without a successful inner_plan");
  if (ctx->plan_is_join && ctx->n_reg + ctx->n_irr != 0)
    return set_err(GIQL_ERR_STATE, "the last giql_hip_inner_join_dev wrote its pairs itself and kept no plan: "
                                   "call giql_hip_inner_plan_dev first");
  InnerState& S = ctx->inner;
  const bool empty = ctx->n_reg + ctx->n_irr == 0;
  if (!empty && (S.uniform == 0 || ctx->n_irr != 0 || ctx->n_c1 != 0))
    return set_err(GIQL_ERR_STATE, "the last plan is not in the compact single-range form "
                                   "(general two-class join or irregular rows): exchange the pairs instead");
  const bool q_is_a = S.uniform != 2;  // in the plan's labels
  *query_is_a = (q_is_a != ctx->swapped) ? 1 : 0;
  if (ctx->swapped) {
    const int32_t t = rid_add_a;
    rid_add_a = rid_add_b;
    rid_add_b = t;
  }
  *n_q = empty ? 0 : (q_is_a ? ctx->n_a : ctx->n_b);
  *n_s = empty ? 0 : (q_is_a ? ctx->n_b : ctx->n_a);
  if (empty) return GIQL_OK;
  if (*n_q > q_capacity || *n_s > s_capacity)
    return set_err(GIQL_ERR_CAPACITY, "plan export needs %lld query rows and %lld sorted rows", (long long)*n_q,
                   (long long)*n_s);
  if (!q_rid_out || !lo_out || !cnt_out || !s_rid_out) return set_err(GIQL_ERR_INVALID, "NULL output");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  const u32* qrid = q_is_a ? S.sa.rid[0] : S.sb.rid[0];
  const u32* srid = q_is_a ? S.sb.rid[0] : S.sa.rid[0];
  u32 grid = cdiv((u64)(*n_q + *n_s), 256 * 8);
  if (grid > GIQL_STREAM_GRID) grid = GIQL_STREAM_GRID;
  hipLaunchKernelGGL(k_plan_export, dim3(grid), dim3(256), 0, st, qrid, S.lo2, S.off2, (u32)*n_q, srid, (u32)*n_s,
                     (u32)(q_is_a ? rid_add_a : rid_add_b), (u32)(q_is_a ? rid_add_b : rid_add_a), q_rid_out,
                     lo_out, cnt_out, s_rid_out);
  return post_launch("plan export");
}

int giql_hip_fill_from_plan_dev(giql_hip_ctx* ctx, const int32_t* q_rid, const uint32_t* lo, const uint32_t* cnt,
                                int64_t n_q, const int32_t* s_rid, int64_t n_s, int32_t* row_q, int32_t* row_s,
                                int64_t capacity, int64_t n_pairs_expected, void* stream, int64_t* n_pairs) {
  if (!ctx || !n_pairs) return set_err(GIQL_ERR_INVALID, "ctx/n_pairs is NULL");
  if (n_q < 0 || n_q > 0x7FFFFFF0ll || n_s < 0 || n_s > 0x7FFFFFF0ll || capacity < 0)
    return set_err(GIQL_ERR_INVALID, "bad sizes");
  *n_pairs = 0;
  if (n_q == 0 || n_s == 0) return GIQL_OK;
  if (!q_rid || !lo || !cnt || !s_rid) return set_err(GIQL_ERR_INVALID, "NULL plan array");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  constexpr u32 T2 = FILL_NT * FILL_ITEMS_C2;
  // scratch: exclusive u64 offsets [n_q + 1], scan partials, partition array for `capacity` pairs
  const u64 nt_cap = ((u64)capacity + T2 - 1) / T2;
  if (nt_cap > 0x7FFFFFF0ull) return set_err(GIQL_ERR_INVALID, "output too large");
  u64* off = nullptr;
  u64* bsums = nullptr;
  u32* part = nullptr;
  auto carve = [&](char* base) {
    Carver c{base};
    off = c.take<u64>((size_t)n_q + 1);
    bsums = c.take<u64>(cdiv((u64)n_q, SCAN_TILE) + 2);
    part = c.take<u32>((size_t)nt_cap + 2);
    return c.off;
  };
  const size_t need = carve(nullptr);
  if (need > ctx->xplan_cap) {
    HIP_TRY(hipStreamSynchronize(st));
    if (ctx->xplan) HIP_TRY(hipFree(ctx->xplan));
    ctx->xplan = nullptr;
    ctx->xplan_cap = 0;
    const size_t want = align_up(need + need / 8, (size_t)1 << 20);
    hipError_t e = hipMalloc((void**)&ctx->xplan, want);
    if (e != hipSuccess) return set_err(GIQL_ERR_NOMEM, "hipMalloc(%zu bytes) for the plan scratch failed", want);
    ctx->xplan_cap = want;
  }
  carve(ctx->xplan);
  GIQL_TRY(run_scan<u64>(ctx, st, GIQL_PH_SCAN, cnt, (u64)n_q, off, bsums, off + n_q));
  u64 total = 0;
  if (n_pairs_expected >= 0) {
    total = (u64)n_pairs_expected;  // the caller knows it (all-gathered counts): no read-back, no stream sync
  } else {
    HIP_TRY(hipMemcpyAsync(&total, off + n_q, sizeof(u64), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  *n_pairs = (int64_t)total;
  if (total == 0) return GIQL_OK;
  if (total > (u64)capacity)
    return set_err(GIQL_ERR_CAPACITY, "capacity %lld < %llu pairs", (long long)capacity, (unsigned long long)total);
  if (!row_q || !row_s) return set_err(GIQL_ERR_INVALID, "row_q/row_s is NULL");
  const u32 nt = (u32)((total + T2 - 1) / T2);
  {
    Phase ph(ctx, st, GIQL_PH_PARTITION);
    hipLaunchKernelGGL(k_partition, dim3(cdiv((u64)nt + 1, 256)), dim3(256), 0, st, off, (u32)n_q, (u64)0, T2, nt,
                       part, (const u64*)(off + n_q), (u64)capacity);
  }
  {
    // the pair count is read on the device (off[n_q]): a wrong n_pairs_expected cannot overrun
    Phase ph(ctx, st, GIQL_PH_FILL);
    hipLaunchKernelGGL((k_fill<FILL_ITEMS_C2>), dim3(nt), dim3(FILL_NT), 0, st, off, lo,
                       reinterpret_cast<const u32*>(q_rid), (u32)n_q, reinterpret_cast<const u32*>(s_rid), part,
                       (u64)0, (u64)0, row_q, row_s, (const u64*)(off + n_q), (u64)capacity);
  }
  return post_launch("fill from plan");
}

// ------------------------------------------------------------ copy probe
// The box's own streaming-copy rate with THIS library's access pattern (16 B per lane, grid-
// stride): the yardstick bench.py reports next to the 8 TB/s peak.
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, u64 n16) {
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

int giql_hip_copy_probe_dev(giql_hip_ctx* ctx, const void* src, void* dst, int64_t bytes, int32_t reps,
                            void* stream, double* gbytes_per_s) {
  if (!ctx || !src || !dst || !gbytes_per_s || bytes < 16 || reps < 1) return set_err(GIQL_ERR_INVALID, "bad arguments");
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return set_err(GIQL_ERR_INVALID, "buffers must be 16-byte aligned");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  const u64 n16 = (u64)bytes / 16;
  const u32 grid = (u32)ctx->n_cu * 8;  // one resident wave of 256-thread blocks
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, n16);  // warm-up
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, n16);
  (void)hipEventRecord(e1, st);
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess) return set_err(GIQL_ERR_HIP, "copy probe failed: %s", hipGetErrorString(e));
  GIQL_TRY(post_launch("copy probe"));
  *gbytes_per_s = ms > 0.f ? 2.0 * (double)(n16 * 16) * reps / ((double)ms * 1e6) : 0.0;
  return GIQL_OK;
}

// The same question asked properly (round 4): what does this device read / write / copy per second, by access
// shape?  mode 0 read only, 1 write only, 2 copy, 3 hipMemcpyDtoDAsync (an outside reference); in_flight = 16-byte
// accesses a thread keeps in flight (1, 2, 4 or 8); nontemporal = nt loads and stores; blocks_per_cu sizes the grid
// (256-thread blocks).  *gbytes_per_s counts every byte moved: read for mode 0, written for 1, both for 2 / 3.
int giql_hip_stream_probe_dev(giql_hip_ctx* ctx, const void* src, void* dst, int64_t bytes, int32_t mode,
                              int32_t in_flight, int32_t nontemporal, int32_t blocks_per_cu, int32_t reps, void* stream,
                              double* gbytes_per_s) {
  if (!ctx || !src || !dst || !gbytes_per_s || bytes < (1 << 20) || reps < 1 || mode < 0 || mode > 3 ||
      blocks_per_cu < 1 || blocks_per_cu > 64 || (in_flight != 1 && in_flight != 2 && in_flight != 4 && in_flight != 8))
    return set_err(GIQL_ERR_INVALID, "bad arguments");
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return set_err(GIQL_ERR_INVALID, "buffers must be 16-byte aligned");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  const u64 tile = (u64)256 * (u64)in_flight;
  const u64 n16 = (u64)bytes / 16 / tile * tile;  // whole tiles only
  const u32 grid = (u32)ctx->n_cu * (u32)blocks_per_cu;
  auto once = [&]() {
    const probe_u4* s = (const probe_u4*)src;
    probe_u4* d = (probe_u4*)dst;
    u32* sk = ctx->bucket_qwin;  // the read-only form's sink (256 words, rewritten by every join that reads them)
    if (mode == 3) {
      (void)hipMemcpyDtoDAsync((hipDeviceptr_t)dst, (hipDeviceptr_t)src, n16 * 16, st);
    } else if (nontemporal) {
      if (mode == 0) launch_stream_probe<0, true>(in_flight, grid, st, s, d, n16, sk);
      else if (mode == 1) launch_stream_probe<1, true>(in_flight, grid, st, s, d, n16, sk);
      else launch_stream_probe<2, true>(in_flight, grid, st, s, d, n16, sk);
    } else {
      if (mode == 0) launch_stream_probe<0, false>(in_flight, grid, st, s, d, n16, sk);
      else if (mode == 1) launch_stream_probe<1, false>(in_flight, grid, st, s, d, n16, sk);
      else launch_stream_probe<2, false>(in_flight, grid, st, s, d, n16, sk);
    }
  };
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  once();  // warm-up
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < reps; r++) once();
  (void)hipEventRecord(e1, st);
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess) return set_err(GIQL_ERR_HIP, "stream probe failed: %s", hipGetErrorString(e));
  GIQL_TRY(post_launch("stream probe"));
  const double moved = (mode >= 2 ? 2.0 : 1.0) * (double)(n16 * 16) * reps;
  *gbytes_per_s = ms > 0.f ? moved / ((double)ms * 1e6) : 0.0;
  return GIQL_OK;
}

// ------------------------------------------------ projection (Arrow take)
int giql_hip_take_dev(giql_hip_ctx* ctx, const void* const* cols, const int32_t* elem_bytes,
                      int32_t n_cols, int64_t n_rows, const int32_t* idx, int64_t n,
                      void* const* outs, void* stream) {
  if (!ctx || n_cols < 0 || n < 0 || n_rows < 0 || n_rows > 0x7FFFFFFFll || (n_cols && (!cols || !elem_bytes || !outs)))
    return set_err(GIQL_ERR_INVALID, "bad arguments");
  if (n > 0 && !idx) return set_err(GIQL_ERR_INVALID, "idx is NULL");
  for (int c = 0; c < n_cols; c++) {
    const int e = elem_bytes[c];
    if (e != 1 && e != 2 && e != 4 && e != 8 && e != 16)
      return set_err(GIQL_ERR_INVALID, "column %d: elem_bytes=%d not in {1,2,4,8,16}", c, e);
    if (n > 0 && (!outs[c] || (n_rows > 0 && !cols[c])))
      return set_err(GIQL_ERR_INVALID, "column %d: NULL buffer", c);
    if (((uintptr_t)cols[c] | (uintptr_t)outs[c]) % (uintptr_t)e)
      return set_err(GIQL_ERR_INVALID, "column %d: buffer not aligned to its element size", c);
  }
  if (n == 0 || n_cols == 0) return GIQL_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  reset_stats(ctx);
  HIP_TRY(hipMemsetAsync(&ctx->d_meta->status, 0, sizeof(int), st));
  u32 grid = cdiv((u64)n, (u64)TK_NT * 4 * 4);
  if (grid > GIQL_STREAM_GRID) grid = GIQL_STREAM_GRID;
  for (int c0 = 0; c0 < n_cols; c0 += TK_MAX_COLS) {
    TakeArgs a;
    memset(&a, 0, sizeof(a));
    a.n_cols = n_cols - c0 < TK_MAX_COLS ? n_cols - c0 : TK_MAX_COLS;
    a.vec = ((uintptr_t)idx % 16) == 0;
    for (int c = 0; c < a.n_cols; c++) {
      a.col[c] = cols[c0 + c];
      a.out[c] = outs[c0 + c];
      a.elem[c] = elem_bytes[c0 + c];
      if ((uintptr_t)a.out[c] % 16) a.vec = 0;
    }
    Phase ph(ctx, st, GIQL_PH_AUX);
    hipLaunchKernelGGL(k_take, dim3(grid), dim3(TK_NT), 0, st, a, idx, (u64)n, (u32)n_rows,
                       ctx->d_meta);
    GIQL_TRY(post_launch("take"));
  }
  GIQL_TRY(read_meta(ctx, st));
  collect_spans(ctx);
  ctx->stats.n_out = n;
  return GIQL_OK;
}

int giql_hip_take_utf8_plan_dev(giql_hip_ctx* ctx, const int32_t* offsets, int64_t n_rows,
                                const int32_t* idx, int64_t n, int32_t* out_offsets,
                                int64_t* n_bytes, void* stream) {
  if (!ctx || !out_offsets || !n_bytes || n < 0 || n > 0x7FFFFFF0ll || n_rows < 0 || n_rows > 0x7FFFFFFFll)
    return set_err(GIQL_ERR_INVALID, "bad arguments");
  if ((n > 0 && !idx) || (n_rows > 0 && !offsets)) return set_err(GIQL_ERR_INVALID, "NULL buffer");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  reset_stats(ctx);
  *n_bytes = 0;
  if (n == 0) {
    HIP_TRY(hipMemsetAsync(out_offsets, 0, sizeof(int32_t), st));
    HIP_TRY(hipStreamSynchronize(st));
    return GIQL_OK;
  }
  Carver c{nullptr};
  u64* bsums = c.take<u64>(cdiv((u64)n, SCAN_TILE) + 1);
  u64* total = c.take<u64>(1);
  GIQL_TRY(ensure_arena(ctx, c.off, st));
  c = Carver{ctx->arena};
  bsums = c.take<u64>(cdiv((u64)n, SCAN_TILE) + 1);
  total = c.take<u64>(1);
  HIP_TRY(hipMemsetAsync(&ctx->d_meta->status, 0, sizeof(int), st));
  u32 grid = cdiv((u64)n, (u64)TK_NT * 4);
  if (grid > GIQL_STREAM_GRID) grid = GIQL_STREAM_GRID;
  {
    Phase ph(ctx, st, GIQL_PH_AUX);
    hipLaunchKernelGGL(k_take_utf8_len, dim3(grid), dim3(TK_NT), 0, st, offsets, (u32)n_rows, idx,
                       (u64)n, reinterpret_cast<u32*>(out_offsets), ctx->d_meta);
    GIQL_TRY(post_launch("take_utf8_len"));
  }
  GIQL_TRY(run_scan<u32>(ctx, st, GIQL_PH_SCAN, reinterpret_cast<u32*>(out_offsets), (u64)n,
                         reinterpret_cast<u32*>(out_offsets), bsums, total));
  // out_offsets[n] = total (low word; the range check below rejects anything wider)
  HIP_TRY(hipMemcpyAsync(out_offsets + n, total, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  u64 h_total = 0;
  HIP_TRY(hipMemcpyAsync(&h_total, total, sizeof(u64), hipMemcpyDeviceToHost, st));
  GIQL_TRY(read_meta(ctx, st));
  collect_spans(ctx);
  if (h_total > 0x7FFFFFFFull)
    return set_err(GIQL_ERR_CAPACITY, "taken utf8 data is %llu bytes: exceeds int32 offsets",
                   (unsigned long long)h_total);
  *n_bytes = (int64_t)h_total;
  return GIQL_OK;
}

int giql_hip_take_utf8_fill_dev(giql_hip_ctx* ctx, const int32_t* offsets, const uint8_t* data,
                                int64_t n_rows, const int32_t* idx, int64_t n,
                                const int32_t* out_offsets, uint8_t* out_data, void* stream) {
  if (!ctx || n < 0 || n_rows < 0 || n_rows > 0x7FFFFFFFll) return set_err(GIQL_ERR_INVALID, "bad arguments");
  if (n == 0) return GIQL_OK;
  if (!idx || !out_offsets || (n_rows > 0 && !offsets)) return set_err(GIQL_ERR_INVALID, "NULL buffer");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  reset_stats(ctx);
  u32 grid = cdiv((u64)n, (u64)TK_NT * 2);
  if (grid > 2u * GIQL_STREAM_GRID) grid = 2u * GIQL_STREAM_GRID;
  {
    Phase ph(ctx, st, GIQL_PH_AUX);
    hipLaunchKernelGGL(k_take_utf8_copy, dim3(grid), dim3(TK_NT), 0, st, offsets, data, (u32)n_rows,
                       idx, (u64)n, out_offsets, out_data);
    GIQL_TRY(post_launch("take_utf8_copy"));
  }
  HIP_TRY(hipStreamSynchronize(st));
  collect_spans(ctx);
  ctx->stats.n_out = n;
  return GIQL_OK;
}

// ------------------------------------------------ residual predicates (select)
static int check_operand(const giql_operand& o, int k, const char* which) {
  if (o.side == GIQL_SIDE_LIT) return GIQL_OK;
  if (o.side != GIQL_SIDE_A && o.side != GIQL_SIDE_B)
    return set_err(GIQL_ERR_INVALID, "predicate %d %s: side %d", k, which, o.side);
  if (o.type < GIQL_T_I32 || o.type > GIQL_T_U8)
    return set_err(GIQL_ERR_INVALID, "predicate %d %s: type %d", k, which, o.type);
  if (!o.data) return set_err(GIQL_ERR_INVALID, "predicate %d %s: NULL column", k, which);
  return GIQL_OK;
}

int giql_hip_select_dev(giql_hip_ctx* ctx, const giql_pred* preds, int32_t n_preds,
                        const int32_t* idx_a, int64_t n_rows_a, const int32_t* idx_b,
                        int64_t n_rows_b, int64_t n, int32_t* out_a, int32_t* out_b,
                        int64_t* n_kept, void* stream) {
  return giql_hip_select_expr_dev(ctx, preds, n_preds, nullptr, 0, idx_a, n_rows_a, idx_b, n_rows_b, n, out_a, out_b,
                                  n_kept, stream);
}

int giql_hip_select_expr_dev(giql_hip_ctx* ctx, const giql_pred* preds, int32_t n_preds,
                             const giql_operand* nodes, int32_t n_nodes,
                             const int32_t* idx_a, int64_t n_rows_a, const int32_t* idx_b,
                             int64_t n_rows_b, int64_t n, int32_t* out_a, int32_t* out_b,
                             int64_t* n_kept, void* stream) {
  if (!ctx || !n_kept || n < 0 || n > 0x7FFFFFF0ll || n_preds < 0 || n_preds > SEL_MAX_PREDS ||
      (n_preds && !preds) || n_rows_a < 0 || n_rows_b < 0 || n_rows_a > 0x7FFFFFFFll ||
      n_rows_b > 0x7FFFFFFFll || n_nodes < 0 || n_nodes > SEL_MAX_NODES || (n_nodes && !nodes))
    return set_err(GIQL_ERR_INVALID, "bad arguments (at most %d predicates and %d expression nodes, n < 2^31)",
                   SEL_MAX_PREDS, SEL_MAX_NODES);
  static_assert(GIQL_SIDE_EXPR == SEL_SIDE_EXPR && GIQL_X_ADD == SEL_X_ADD && GIQL_X_GREATEST == SEL_X_GREATEST &&
                GIQL_X_NEG == SEL_X_NEG && GIQL_X_ABS == SEL_X_ABS && GIQL_X_DIV == SEL_X_DIV, "expression node kinds");
  DevPreds ps;
  bool uses[2] = {false, false};
  bool any_expr = false;
  GIQL_TRY(convert_preds(preds, n_preds, ps, uses, nodes, n_nodes, &any_expr));
  // a side given neither ids nor a row count is addressed by the candidate index
  if (!idx_a && n_rows_a == 0 && !uses[0]) n_rows_a = n;
  if (!idx_b && n_rows_b == 0 && !uses[1]) n_rows_b = n;
  *n_kept = 0;
  if (n == 0) return GIQL_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  reset_stats(ctx);
  const u32 nb = cdiv((u64)n, SEL_TILE);
  Carver c{nullptr};
  DevOperand* prog = nullptr;
  auto carve = [&](Carver& cv, u64*& mask, u32*& cnt, u64*& bsums, u64*& total) {
    mask = cv.take<u64>(((size_t)n + 63) / 64);
    cnt = cv.take<u32>(nb);
    bsums = cv.take<u64>(cdiv((u64)nb, SCAN_TILE) + 1);
    total = cv.take<u64>(1);
    prog = cv.take<DevOperand>(SEL_MAX_NODES);
  };
  u64 *mask, *bsums, *total;
  u32* cnt;
  carve(c, mask, cnt, bsums, total);
  GIQL_TRY(ensure_arena(ctx, c.off, st));
  c = Carver{ctx->arena};
  carve(c, mask, cnt, bsums, total);
  HIP_TRY(hipMemsetAsync(&ctx->d_meta->status, 0, sizeof(int), st));
  if (any_expr)  // (pageable source: staged by the runtime before the call returns)
    HIP_TRY(hipMemcpyAsync(prog, nodes, (size_t)n_nodes * sizeof(DevOperand), hipMemcpyHostToDevice, st));
  {
    Phase ph(ctx, st, GIQL_PH_COUNT);
    if (any_expr)
      hipLaunchKernelGGL((k_select_count<true>), dim3(nb), dim3(SEL_NT), 0, st, ps, (const DevOperand*)prog, idx_a,
                         (u32)n_rows_a, idx_b, (u32)n_rows_b, (u64)n, mask, cnt, ctx->d_meta);
    else
      hipLaunchKernelGGL((k_select_count<false>), dim3(nb), dim3(SEL_NT), 0, st, ps, (const DevOperand*)nullptr, idx_a,
                         (u32)n_rows_a, idx_b, (u32)n_rows_b, (u64)n, mask, cnt, ctx->d_meta);
    GIQL_TRY(post_launch("select_count"));
  }
  GIQL_TRY(run_scan<u32>(ctx, st, GIQL_PH_SCAN, cnt, (u64)nb, cnt, bsums, total));
  if (out_a || out_b) {
    Phase ph(ctx, st, GIQL_PH_FILL);
    hipLaunchKernelGGL(k_select_scatter, dim3(nb), dim3(SEL_NT), 0, st, idx_a, idx_b, (u64)n, mask, cnt,
                       out_a, out_b);
    GIQL_TRY(post_launch("select_scatter"));
  }
  u64 h_total = 0;
  HIP_TRY(hipMemcpyAsync(&h_total, total, sizeof(u64), hipMemcpyDeviceToHost, st));
  GIQL_TRY(read_meta(ctx, st));
  collect_spans(ctx);
  *n_kept = (int64_t)h_total;
  ctx->stats.n_out = (int64_t)h_total;
  return GIQL_OK;
}

int giql_hip_mark_dev(giql_hip_ctx* ctx, const int32_t* idx, int64_t n, uint8_t* flags,
                      int64_t n_rows, void* stream) {
  if (!ctx || n < 0 || n_rows < 0 || n_rows > 0x7FFFFFFFll || (n > 0 && (!idx || !flags)))
    return set_err(GIQL_ERR_INVALID, "bad arguments");
  if (n == 0) return GIQL_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  reset_stats(ctx);
  HIP_TRY(hipMemsetAsync(&ctx->d_meta->status, 0, sizeof(int), st));
  u32 grid = cdiv((u64)n, 256 * 8);
  if (grid > GIQL_STREAM_GRID) grid = GIQL_STREAM_GRID;
  {
    Phase ph(ctx, st, GIQL_PH_AUX);
    hipLaunchKernelGGL(k_mark, dim3(grid), dim3(256), 0, st, idx, (u64)n, (u32)n_rows, flags, ctx->d_meta);
    GIQL_TRY(post_launch("mark"));
  }
  GIQL_TRY(read_meta(ctx, st));
  collect_spans(ctx);
  return GIQL_OK;
}

// ----------------------------------------------------- host-buffer variants
struct DevSide {
  giql_side s;
  int32_t* buf = nullptr;
  ~DevSide() {
    if (buf) (void)hipFree(buf);
  }
};

static int upload_side(const giql_side* h, DevSide& d) {
  d.s = *h;
  if (h->n == 0) {
    d.s.chrom = d.s.start = d.s.end = nullptr;
    return GIQL_OK;
  }
  const size_t n = (size_t)h->n;
  HIP_TRY(hipMalloc((void**)&d.buf, 3 * n * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(d.buf, h->chrom, n * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d.buf + n, h->start, n * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d.buf + 2 * n, h->end, n * sizeof(int32_t), hipMemcpyHostToDevice));
  d.s.chrom = d.buf;
  d.s.start = d.buf + n;
  d.s.end = d.buf + 2 * n;
  return GIQL_OK;
}

// Pinned host memory for library-owned outputs (released by giql_hip_free_host).  Page-locking is the
// expensive part of the PCIe-inclusive calls -- 3.2 GB of pairs: 203 ms to pin, 57 ms to copy
// (GIQL_HIP_DEBUG_E2E=1) -- so released buffers are kept for the next call: a small process-wide pool,
// best fit, at most GIQL_HIP_HOST_POOL_MB (default 8192; 0 = none) of idle memory.
struct HostPool {
  struct Buf {
    void* p;
    size_t bytes;
    bool idle;
    bool pinned;  // page-locked (hipHostMalloc) or plain memory (the compact-plan path's outputs: written by host threads)
  };
  static void release(void* p, bool pinned) {
    if (pinned)
      (void)hipHostFree(p);
    else
      free(p);
  }
  std::mutex mu;
  std::vector<Buf> bufs;
  size_t idle_limit;
  HostPool() {
    const char* e = getenv("GIQL_HIP_HOST_POOL_MB");
    idle_limit = (size_t)(e ? strtoull(e, nullptr, 10) : 8192ull) << 20;
  }
  void* get(size_t bytes, bool pinned = true) {
    {
      std::lock_guard<std::mutex> g(mu);
      Buf* best = nullptr;
      for (auto& b : bufs)  // (a page-locked buffer serves a plain request as well)
        if (b.idle && b.bytes >= bytes && (b.pinned || !pinned) && (!best || b.bytes < best->bytes)) best = &b;
      if (best && best->bytes <= 2 * bytes + (1u << 20)) {  // not a 3 GB buffer for a 4-byte result
        best->idle = false;
        return best->p;
      }
    }
    void* p = nullptr;
    size_t cap = bytes + bytes / 16;  // a little head-room: the next result of a similar call fits too
    if (pinned) {
      if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    } else {
      // plain memory on 2 MiB boundaries, huge pages asked for: the expansion threads touch it for the first time
      // (4 KB pages: ~800,000 faults for 3.2 GB of pairs)
      cap = align_up(cap, (size_t)2 << 20);
      p = aligned_alloc((size_t)2 << 20, cap);
      if (!p) return nullptr;
#if defined(MADV_HUGEPAGE)
      (void)madvise(p, cap, MADV_HUGEPAGE);
#endif
    }
    std::lock_guard<std::mutex> g(mu);
    bufs.push_back({p, cap, false, pinned});
    return p;
  }
  void put(void* p) {
    std::vector<std::pair<void*, bool>> drop;
    {
      std::lock_guard<std::mutex> g(mu);
      bool known = false;
      for (auto& b : bufs)
        if (b.p == p) {
          b.idle = true;
          known = true;
        }
      if (!known) drop.push_back({p, true});
      // over the limit: release the smallest idle buffers first (the big ones are the ones worth keeping)
      size_t idle = 0;
      for (auto& b : bufs)
        if (b.idle) idle += b.bytes;
      while (idle > idle_limit) {
        size_t k = bufs.size();
        for (size_t i = 0; i < bufs.size(); i++)
          if (bufs[i].idle && (k == bufs.size() || bufs[i].bytes < bufs[k].bytes)) k = i;
        if (k == bufs.size()) break;
        idle -= bufs[k].bytes;
        drop.push_back({bufs[k].p, bufs[k].pinned});
        bufs.erase(bufs.begin() + (long)k);
      }
    }
    for (auto& d : drop) release(d.first, d.second);
  }
};
static HostPool& host_pool() {
  static HostPool* pool = new HostPool();  // never destroyed: no HIP calls at process exit
  return *pool;
}
static void* host_alloc(size_t bytes) { return host_pool().get(bytes ? bytes : 1); }
static void* host_alloc_plain(size_t bytes) { return host_pool().get(bytes ? bytes : 1, false); }

int giql_hip_host_pool_trim(int64_t keep_bytes, int64_t* released) {
  if (keep_bytes < 0) return set_err(GIQL_ERR_INVALID, "keep_bytes < 0");
  HostPool& hp = host_pool();
  std::vector<std::pair<void*, bool>> drop;
  size_t freed = 0;
  {
    std::lock_guard<std::mutex> g(hp.mu);
    size_t idle = 0;
    for (auto& b : hp.bufs)
      if (b.idle) idle += b.bytes;
    while (idle > (size_t)keep_bytes) {  // the largest first: they are what a trim is called for
      size_t k = hp.bufs.size();
      for (size_t i = 0; i < hp.bufs.size(); i++)
        if (hp.bufs[i].idle && (k == hp.bufs.size() || hp.bufs[i].bytes > hp.bufs[k].bytes)) k = i;
      if (k == hp.bufs.size()) break;
      idle -= hp.bufs[k].bytes;
      freed += hp.bufs[k].bytes;
      drop.push_back({hp.bufs[k].p, hp.bufs[k].pinned});
      hp.bufs.erase(hp.bufs.begin() + (long)k);
    }
  }
  for (auto& d : drop) HostPool::release(d.first, d.second);
  if (released) *released = (int64_t)freed;
  return GIQL_OK;
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};

// ---- the host-buffer INNER join, pipelined over row blocks of the larger table (round 3) ----
// One shot, the call is a chain on the PCIe link: 1.3 GB of columns in (23 ms), the join (2.4 ms), 3.2 GB of
// pairs out (57 ms).  The link is full duplex and an INNER join is a union over row blocks of one side --
// A x B = U_j A x B_j -- so the larger table goes up block by block: while block j's pairs travel to the host on a
// copy stream, block j + 1's columns travel to the device (the host thread sits in that copy) and its join runs;
// only the first upload and the last download stand alone.  The smaller table is uploaded once and sorted again
// per block (a fraction of a millisecond each).  Row ids of the blocked table get the block's first row added on
// the device.  GIQL_HIP_E2E_BLOCK_ROWS: rows per block (default 4M; 0 = one shot).  Measured at 10M x 100M on a
// settled context: 83 ms one shot, 88 / 76 / 71 / 73 ms with blocks of 16M / 8M / 4M / 2M rows -- an upload next to a
// download runs at 25-40 GB/s instead of 56, so the two directions overlap only in part.
__global__ __launch_bounds__(256) void k_add_i32(int32_t* __restrict__ v, u64 n, int32_t add) {
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) v[i] += add;
}

struct E2eResources {  // released on every path out of inner_host_pipelined
  hipStream_t cs = nullptr;
  hipEvent_t filled[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
  int32_t* in[2] = {nullptr, nullptr};
  int32_t* out[2] = {nullptr, nullptr};
  size_t out_cap[2] = {0, 0};  // pairs per array
  int32_t *ha = nullptr, *hb = nullptr;
  size_t h_cap = 0;
  ~E2eResources() {
    if (cs) (void)hipStreamSynchronize(cs);
    for (int k = 0; k < 2; k++) {
      if (in[k]) (void)hipFree(in[k]);
      if (out[k]) (void)hipFree(out[k]);
      if (filled[k]) (void)hipEventDestroy(filled[k]);
      if (copied[k]) (void)hipEventDestroy(copied[k]);
    }
    if (cs) (void)hipStreamDestroy(cs);
    giql_hip_free_host(ha);
    giql_hip_free_host(hb);
  }
};

static int inner_host_pipelined(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                                size_t block_rows, int64_t* n_pairs, int32_t** row_a, int32_t** row_b) {
  const bool b_big = b->n >= a->n;
  const giql_side* big = b_big ? b : a;
  const giql_side* small = b_big ? a : b;
  DevSide dsmall;
  GIQL_TRY(upload_side(small, dsmall));
  E2eResources R;
  const bool dbg = getenv("GIQL_HIP_DEBUG_E2E") != nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&R.cs, hipStreamNonBlocking));
  for (int k = 0; k < 2; k++) {
    HIP_TRY(hipEventCreateWithFlags(&R.filled[k], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&R.copied[k], hipEventDisableTiming));
    HIP_TRY(hipMalloc((void**)&R.in[k], 3 * block_rows * sizeof(int32_t)));
  }
  const size_t n_big = (size_t)big->n;
  size_t total = 0;
  int slot = 0;
  for (size_t j0 = 0; j0 < n_big; j0 += block_rows, slot ^= 1) {
    const size_t nblk = n_big - j0 < block_rows ? n_big - j0 : block_rows;
    // this slot's columns were last read by the join of two blocks ago, its pairs last copied out then too
    HIP_TRY(hipEventSynchronize(R.filled[slot]));
    int32_t* in = R.in[slot];
    const auto t_up = std::chrono::steady_clock::now();
    // (plain synchronous copies: issued on a stream of their own they took twice as long next to the downloads)
    HIP_TRY(hipMemcpy(in, big->chrom + j0, nblk * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(in + block_rows, big->start + j0, nblk * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(in + 2 * block_rows, big->end + j0, nblk * sizeof(int32_t), hipMemcpyHostToDevice));
    if (dbg)
      fprintf(stderr, "[giql_hip_inner] block at row %zu: upload %.2f ms\n", j0,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up).count());
    giql_side blk = *big;
    blk.chrom = in;
    blk.start = in + block_rows;
    blk.end = in + 2 * block_rows;
    blk.n = (int64_t)nblk;
    int64_t nj = 0;
    GIQL_TRY(giql_hip_inner_plan_dev(ctx, b_big ? &dsmall.s : &blk, b_big ? &blk : &dsmall.s, n_chrom, nullptr, &nj));
    HIP_TRY(hipEventSynchronize(R.copied[slot]));  // the slot's previous pairs have left the device
    if (nj > 0) {
      const size_t stride = align_up((size_t)nj, (size_t)1 << 19);  // each row on a 2 MiB boundary (the fill's stores)
      if (stride > R.out_cap[slot]) {
        if (R.out[slot]) HIP_TRY(hipFree(R.out[slot]));
        R.out[slot] = nullptr;
        R.out_cap[slot] = 0;
        const size_t want = stride + stride / 8;
        if (hipMalloc((void**)&R.out[slot], 2 * want * sizeof(int32_t)) != hipSuccess)
          return set_err(GIQL_ERR_NOMEM, "hipMalloc for %lld pairs failed", (long long)nj);
        R.out_cap[slot] = want;
      }
      int32_t* oa = R.out[slot];
      int32_t* ob = oa + R.out_cap[slot];
      GIQL_TRY(giql_hip_inner_fill_dev(ctx, oa, ob, nj, nullptr));
      if (j0 > 0) {
        u32 grid = cdiv((u64)nj, 256 * 8);
        if (grid > GIQL_STREAM_GRID) grid = GIQL_STREAM_GRID;
        hipLaunchKernelGGL(k_add_i32, dim3(grid), dim3(256), 0, (hipStream_t) nullptr, b_big ? ob : oa, (u64)nj, (int32_t)j0);
        GIQL_TRY(post_launch("row id offset"));
      }
      // host arrays: sized from the first block's yield (the pool usually hands back the previous call's arrays)
      if (total + (size_t)nj > R.h_cap) {
        HIP_TRY(hipStreamSynchronize(R.cs));  // nothing in flight into the arrays that are replaced
        const double per_row = (double)(total + (size_t)nj) / (double)(j0 + nblk);
        size_t want = (size_t)(per_row * (double)n_big * 1.08) + (1u << 20);
        if (want < total + (size_t)nj) want = total + (size_t)nj;
        int32_t* na_ = (int32_t*)host_alloc(want * sizeof(int32_t));
        int32_t* nb_ = (int32_t*)host_alloc(want * sizeof(int32_t));
        if (!na_ || !nb_) {
          giql_hip_free_host(na_);
          giql_hip_free_host(nb_);
          return set_err(GIQL_ERR_NOMEM, "out of host memory for %zu pairs", want);
        }
        if (total) {
          memcpy(na_, R.ha, total * sizeof(int32_t));
          memcpy(nb_, R.hb, total * sizeof(int32_t));
        }
        giql_hip_free_host(R.ha);
        giql_hip_free_host(R.hb);
        R.ha = na_;
        R.hb = nb_;
        R.h_cap = want;
      }
      HIP_TRY(hipEventRecord(R.filled[slot], nullptr));
      HIP_TRY(hipStreamWaitEvent(R.cs, R.filled[slot], 0));
      HIP_TRY(hipMemcpyAsync(R.ha + total, oa, (size_t)nj * sizeof(int32_t), hipMemcpyDeviceToHost, R.cs));
      HIP_TRY(hipMemcpyAsync(R.hb + total, ob, (size_t)nj * sizeof(int32_t), hipMemcpyDeviceToHost, R.cs));
      HIP_TRY(hipEventRecord(R.copied[slot], R.cs));
      total += (size_t)nj;
    }
  }
  HIP_TRY(hipStreamSynchronize(R.cs));
  if (!R.ha) {  // no pair at all: the caller still gets arrays to free
    R.ha = (int32_t*)host_alloc(sizeof(int32_t));
    R.hb = (int32_t*)host_alloc(sizeof(int32_t));
    if (!R.ha || !R.hb) return set_err(GIQL_ERR_NOMEM, "out of host memory");
  }
  *n_pairs = (int64_t)total;
  *row_a = R.ha;
  *row_b = R.hb;
  R.ha = R.hb = nullptr;  // handed over
  return GIQL_OK;
}

// ---- the host-buffer INNER join with a COMPACT-PLAN download (round 4; GIQL_HIP_E2E_COMPACT=1) ----
// 3.2 GB of pairs take 57 ms on the PCIe link; the plan they are made of -- per query row {row id, first match,
// count} + the other side's row ids in sorted order, giql_hip_inner_plan_export_dev -- is 0.52 GB at the headline
// sizes (9 ms), and host threads expand it at the host's memory rate (tools/probes/host_expand_probe.cpp: 150 GB/s
// of pairs with 16 threads on the GPU box = 21 ms for 3.2 GB).  So: columns up, plan, export, the three per-query
// arrays down, then the sorted ids in chunks while the threads already expand the queries whose ranges have arrived
// (blocks of 65,536 queries handed out in order; a block waits for the chunk that holds its last id).  The pairs
// come out in the plan's query order.  Only the single-range form (fixed-length side, no irregular rows) has a compact
// plan: any other plan is filled and downloaded as before.  GIQL_HIP_E2E_THREADS: expansion threads (default 32: tools/e2e_threads.py -- 16: 45.8 ms, 32: 39.9, 64: 40.0 at the headline sizes).
struct CompactHost {  // page-locked staging of the plan, back to the pool on every path out
  int32_t* q_rid = nullptr;
  u32 *lo = nullptr, *cnt = nullptr;
  int32_t* s_rid = nullptr;
  ~CompactHost() {
    giql_hip_free_host(q_rid);
    giql_hip_free_host(lo);
    giql_hip_free_host(cnt);
    giql_hip_free_host(s_rid);
  }
};

// returns GIQL_OK with *done = false when the plan has no compact form (the caller fills and downloads)
static int inner_host_compact_tail(giql_hip_ctx* ctx, int64_t n, int32_t* ha, int32_t* hb, bool* done) {
  *done = false;
  InnerState& S = ctx->inner;
  if (n <= 0 || S.uniform == 0 || ctx->n_irr != 0 || ctx->n_c1 != 0 || ctx->plan_is_join) return GIQL_OK;
  const bool q_is_a_plan = S.uniform != 2;
  const size_t nq = (size_t)(q_is_a_plan ? ctx->n_a : ctx->n_b), ns = (size_t)(q_is_a_plan ? ctx->n_b : ctx->n_a);
  if (nq == 0 || ns == 0) return GIQL_OK;
  // device staging: the context's output staging buffer (3 nq + ns words)
  const size_t need = (3 * nq + ns + 64) * sizeof(u32);
  if (need > ctx->stage_out_cap) {
    if (ctx->stage_out) (void)hipFree(ctx->stage_out);
    ctx->stage_out = nullptr;
    ctx->stage_out_cap = 0;
    if (hipMalloc(&ctx->stage_out, need + need / 16) != hipSuccess)
      return set_err(GIQL_ERR_NOMEM, "hipMalloc for the compact plan (%zu bytes) failed", need);
    ctx->stage_out_cap = need + need / 16;
  }
  int32_t* d_qrid = (int32_t*)ctx->stage_out;
  u32* d_lo = (u32*)d_qrid + nq;
  u32* d_cnt = d_lo + nq;
  int32_t* d_srid = (int32_t*)(d_cnt + nq);
  int32_t query_is_a = 0;
  int64_t xq = 0, xs = 0;
  const int rc_x = giql_hip_inner_plan_export_dev(ctx, d_qrid, d_lo, d_cnt, d_srid, (int64_t)nq, (int64_t)ns, 0, 0,
                                                  &query_is_a, &xq, &xs, nullptr);
  if (rc_x == GIQL_ERR_STATE) return GIQL_OK;
  GIQL_TRY(rc_x);
  if ((size_t)xq != nq || (size_t)xs != ns) return set_err(GIQL_ERR_STATE, "plan export sizes changed under the call");
  CompactHost H;
  H.q_rid = (int32_t*)host_alloc(nq * 4);
  H.lo = (u32*)host_alloc(nq * 4);
  H.cnt = (u32*)host_alloc(nq * 4);
  H.s_rid = (int32_t*)host_alloc(ns * 4);
  if (!H.q_rid || !H.lo || !H.cnt || !H.s_rid) return set_err(GIQL_ERR_NOMEM, "out of host memory for the compact plan");
  constexpr size_t CHUNK = (size_t)16 << 20;   // sorted ids per download chunk (64 MB: ~1.1 ms on the link)
  constexpr size_t QBLK = (size_t)1 << 16;     // queries per expansion block
  const size_t n_chunks = (ns + CHUNK - 1) / CHUNK, n_blk = (nq + QBLK - 1) / QBLK;
  std::vector<hipEvent_t> ev(n_chunks + 1, nullptr);
  struct EvGuard {
    std::vector<hipEvent_t>& e;
    ~EvGuard() {
      for (hipEvent_t x : e)
        if (x) (void)hipEventDestroy(x);
    }
  } ev_guard{ev};
  for (auto& e : ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipStream_t st = nullptr;  // the export kernel ran on the null stream: the copies follow it there
  HIP_TRY(hipMemcpyAsync(H.cnt, d_cnt, nq * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(H.lo, d_lo, nq * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(H.q_rid, d_qrid, nq * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipEventRecord(ev[0], st));
  for (size_t c = 0; c < n_chunks; c++) {
    const size_t c0 = c * CHUNK, cn = ns - c0 < CHUNK ? ns - c0 : CHUNK;
    HIP_TRY(hipMemcpyAsync(H.s_rid + c0, d_srid + c0, cn * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(ev[c + 1], st));
  }
  HIP_TRY(hipEventSynchronize(ev[0]));  // the per-query arrays are here
  int n_thr = 32;
  if (const char* e = getenv("GIQL_HIP_E2E_THREADS")) n_thr = atoi(e);
  const int hw = (int)std::thread::hardware_concurrency();
  if (hw > 0 && n_thr > hw) n_thr = hw;
  if (n_thr < 1) n_thr = 1;
  if ((size_t)n_thr > n_blk) n_thr = (int)n_blk;
  bool stream_stores = true;  // GIQL_HIP_E2E_NT=0: plain stores
  if (const char* e = getenv("GIQL_HIP_E2E_NT")) stream_stores = atoi(e) != 0;
  (void)stream_stores;
  // block offsets: per-block sums in parallel, a serial scan over the (few hundred) blocks
  std::vector<u64> blk_off(n_blk + 1, 0);
  std::vector<u32> blk_need(n_blk, 0);  // one past the last sorted id a block reads
  std::atomic<size_t> next{0};
  std::atomic<size_t> chunks_here{0};
  std::atomic<int> bad{0};
  const u32* const lo = H.lo;
  const u32* const cnt = H.cnt;
  auto sums = [&] {
    for (;;) {
      const size_t bk = next.fetch_add(1);
      if (bk >= n_blk) return;
      const size_t q0 = bk * QBLK, q1 = q0 + QBLK < nq ? q0 + QBLK : nq;
      u64 t = 0;
      u32 need_s = 0;
      for (size_t q = q0; q < q1; q++) {
        t += cnt[q];
        const u64 e = (u64)lo[q] + cnt[q];
        if (cnt[q] && e > need_s) need_s = e > (u64)ns ? 0xFFFFFFFFu : (u32)e;
      }
      blk_off[bk + 1] = t;
      blk_need[bk] = need_s;
    }
  };
  {
    std::vector<std::thread> th;
    try {
      for (int t = 1; t < n_thr; t++) th.emplace_back(sums);
    } catch (...) {  // no more threads to be had: the ones that started (and this one) do the work
    }
    sums();
    for (auto& t : th) t.join();
  }
  for (size_t bk = 0; bk < n_blk; bk++) {
    if (blk_need[bk] == 0xFFFFFFFFu) return set_err(GIQL_ERR_STATE, "compact plan: a range past the sorted ids");
    blk_off[bk + 1] += blk_off[bk];
  }
  if (blk_off[n_blk] != (u64)n) return set_err(GIQL_ERR_STATE, "compact plan: %llu pairs, the plan counted %lld",
                                               (unsigned long long)blk_off[n_blk], (long long)n);
  int32_t* const out_q = query_is_a ? ha : hb;
  int32_t* const out_s = query_is_a ? hb : ha;
  const int32_t* const q_rid = H.q_rid;
  const int32_t* const s_rid = H.s_rid;
  next.store(0);
  auto expand = [&] {
    for (;;) {
      const size_t bk = next.fetch_add(1);
      if (bk >= n_blk) return;
      const size_t want = ((size_t)blk_need[bk] + CHUNK - 1) / CHUNK;  // chunks that must have arrived
      while (chunks_here.load(std::memory_order_acquire) < want) {
        if (bad.load()) return;
        std::this_thread::yield();
      }
      const size_t q0 = bk * QBLK, q1 = q0 + QBLK < nq ? q0 + QBLK : nq;
      u64 o = blk_off[bk];
#if defined(__SSE2__)
      if (stream_stores) {
        // The pairs are staged in two 4 KB buffers (L1) and leave in aligned 16-byte NON-TEMPORAL stores: a plain store
        // makes the core read every output line before writing it, i.e. 6.4 GB of memory traffic for 3.2 GB of pairs.
        // Both arrays come page-aligned from the pool and share the offset, so one alignment serves both.
        constexpr u32 BUF = 1024;
        alignas(64) int32_t bq[BUF], bs[BUF];
        const u64 o_end = blk_off[bk + 1];
        size_t q = q0;
        u32 k = 0;  // next pair of query q
        auto next_pair = [&](int32_t& vq, int32_t& vs) {  // (the block holds o_end - o more pairs: never runs past q1)
          while (k >= cnt[q]) {
            q++;
            k = 0;
          }
          vq = q_rid[q];
          vs = s_rid[lo[q] + k];
          k++;
        };
        while (o < o_end && (o & 15u)) {  // up to the first 64-byte boundary: plain stores
          next_pair(out_q[o], out_s[o]);
          o++;
        }
        while (o_end - o >= BUF) {
          u32 f = 0;
          while (f < BUF) {  // whole runs of one query at a time
            while (k >= cnt[q]) {
              q++;
              k = 0;
            }
            const u32 c = cnt[q] - k < BUF - f ? cnt[q] - k : BUF - f;
            const int32_t id = q_rid[q];
            const int32_t* src = s_rid + lo[q] + k;
            for (u32 j = 0; j < c; j++) {
              bq[f + j] = id;
              bs[f + j] = src[j];
            }
            f += c;
            k += c;
          }
          for (u32 j = 0; j < BUF; j += 4) {
            _mm_stream_si128(reinterpret_cast<__m128i*>(out_q + o + j), _mm_load_si128(reinterpret_cast<const __m128i*>(bq + j)));
            _mm_stream_si128(reinterpret_cast<__m128i*>(out_s + o + j), _mm_load_si128(reinterpret_cast<const __m128i*>(bs + j)));
          }
          o += BUF;
        }
        while (o < o_end) {
          next_pair(out_q[o], out_s[o]);
          o++;
        }
        _mm_sfence();
        continue;
      }
#endif
      for (size_t q = q0; q < q1; q++) {
        const u32 c = cnt[q];
        const int32_t id = q_rid[q];
        const int32_t* src = s_rid + lo[q];
        for (u32 k = 0; k < c; k++) {
          out_q[o + k] = id;
          out_s[o + k] = src[k];
        }
        o += c;
      }
    }
  };
  std::vector<std::thread> th;
  try {
    for (int t = 0; t < n_thr; t++) th.emplace_back(expand);
  } catch (...) {  // (as above; this thread expands too once every chunk has arrived)
  }
  int rc = GIQL_OK;
  for (size_t c = 0; c < n_chunks; c++) {  // this thread follows the link and publishes what has arrived
    if (hipEventSynchronize(ev[c + 1]) != hipSuccess) {
      rc = set_err(GIQL_ERR_HIP, "D2H copy of the compact plan failed");
      bad.store(1);
      break;
    }
    chunks_here.store(c + 1, std::memory_order_release);
  }
  if (rc == GIQL_OK) expand();  // whatever blocks are left (all of them when no helper thread could be started)
  for (auto& t : th) t.join();
  GIQL_TRY(rc);
  *done = true;
  return GIQL_OK;
}

int giql_hip_inner(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                   int64_t* n_pairs, int32_t** row_a, int32_t** row_b) {
  if (!ctx || !n_pairs || !row_a || !row_b) return set_err(GIQL_ERR_INVALID, "NULL argument");
  GIQL_TRY(check_side(a, "a"));
  GIQL_TRY(check_side(b, "b"));
  HIP_TRY(hipSetDevice(ctx->device));
  *row_a = *row_b = nullptr;
  // GIQL_HIP_E2E_COMPACT: 1 = compact-plan download whenever the plan has that form, 0 = never (round 3: pairs
  // downloaded, the larger table uploaded block by block); unset = compact unless this context's last plan says the
  // tables do not take the single-range form (then the block pipeline, which hides most of the upload, is the better bet)
  const char* e_compact = getenv("GIQL_HIP_E2E_COMPACT");
  const int compact_mode = e_compact ? (atoi(e_compact) != 0 ? 1 : 0) : -1;
  const bool try_compact = compact_mode == 1 ||
                           (compact_mode < 0 && !(ctx->spec_valid && (ctx->spec_form == 0 || !ctx->last_no_irr)));
  if (!try_compact) {
    const char* e = getenv("GIQL_HIP_E2E_BLOCK_ROWS");
    const size_t block_rows = e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)4 << 20);
    const size_t n_big = (size_t)(a->n > b->n ? a->n : b->n);
    if (block_rows > 0 && n_big >= 2 * block_rows && a->n > 0 && b->n > 0)
      return inner_host_pipelined(ctx, a, b, n_chrom, block_rows, n_pairs, row_a, row_b);
  }
  // GIQL_HIP_DEBUG_E2E=1: where the PCIe-inclusive call spends its wall time (stderr)
  const bool dbg = getenv("GIQL_HIP_DEBUG_E2E") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(now() - t).count();
  };
  auto t0 = now();
  DevSide da, db;
  GIQL_TRY(upload_side(a, da));
  GIQL_TRY(upload_side(b, db));
  const double ms_h2d = ms_since(t0);
  t0 = now();
  int64_t n = 0;
  GIQL_TRY(giql_hip_inner_plan_dev(ctx, &da.s, &db.s, n_chrom, nullptr, &n));
  const double ms_plan = ms_since(t0);
  t0 = now();
  *n_pairs = n;
  // The compact plan pays when it is smaller than the pairs (and the result is worth a team of threads)
  bool compact = false;
  if (try_compact && n > 0 && ctx->inner.uniform != 0 && ctx->n_irr == 0 && ctx->n_c1 == 0 && !ctx->plan_is_join) {
    const bool q_is_a_plan = ctx->inner.uniform != 2;
    const int64_t nq = q_is_a_plan ? ctx->n_a : ctx->n_b, ns = q_is_a_plan ? ctx->n_b : ctx->n_a;
    // ... on a host with the cores to expand it: 16 threads write ~150 GB/s of pairs, 4 would lose against the link
    const unsigned hw = std::thread::hardware_concurrency();
    compact = compact_mode == 1 || (n >= (4ll << 20) && 8 * n >= 12 * nq + 4 * ns && hw >= 16);
  }
  // library-owned host outputs are PINNED (the D2H copy of the pairs runs at link speed, not through a pageable
  // bounce buffer) -- unless host threads write them (compact plan): plain memory then, nothing to page-lock (3.2 GB:
  // ~200 ms on a first call); giql_hip_free_host releases either kind
  const size_t out_bytes = (size_t)(n > 0 ? n : 1) * sizeof(int32_t);
  int32_t* ha = (int32_t*)(compact ? host_alloc_plain(out_bytes) : host_alloc(out_bytes));
  int32_t* hb = (int32_t*)(compact ? host_alloc_plain(out_bytes) : host_alloc(out_bytes));
  if (!ha || !hb) {
    giql_hip_free_host(ha);
    giql_hip_free_host(hb);
    return set_err(GIQL_ERR_NOMEM, "out of host memory for %lld pairs", (long long)n);
  }
  const double ms_pin = ms_since(t0);
  t0 = now();
  double ms_fill = 0, ms_d2h = 0;
  bool compact_done = false;
  if (compact && n > 0) {
    const int rc = inner_host_compact_tail(ctx, n, ha, hb, &compact_done);
    if (rc != GIQL_OK) {
      giql_hip_free_host(ha);
      giql_hip_free_host(hb);
      return rc;
    }
  }
  if (n > 0 && !compact_done) {
    // row_b starts on a 2 MiB boundary of its own: a row that begins in the middle of a cache
    // line makes every 256-byte wave store of the fill touch three lines instead of two
    const size_t stride = align_up((size_t)n, (size_t)1 << 19);
    // the device staging of the pairs is kept by the context (a fresh 3.2 GB hipMalloc costs ~160 ms)
    const size_t need = 2 * stride * sizeof(int32_t);
    int rc = GIQL_OK;
    if (need > ctx->stage_out_cap) {
      if (ctx->stage_out) (void)hipFree(ctx->stage_out);
      ctx->stage_out = nullptr;
      ctx->stage_out_cap = 0;
      const size_t want = need + need / 16;
      if (hipMalloc(&ctx->stage_out, want) != hipSuccess)
        rc = set_err(GIQL_ERR_NOMEM, "hipMalloc for %lld pairs failed", (long long)n);
      else
        ctx->stage_out_cap = want;
    }
    int32_t* d_a = (int32_t*)ctx->stage_out;
    int32_t* d_b = d_a + stride;
    if (rc == GIQL_OK) rc = giql_hip_inner_fill_dev(ctx, d_a, d_b, n, nullptr);
    if (dbg) {
      (void)hipDeviceSynchronize();
      ms_fill = ms_since(t0);
      t0 = now();
    }
    if (rc == GIQL_OK && hipMemcpy(ha, d_a, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)
      rc = set_err(GIQL_ERR_HIP, "D2H copy of row_a failed");
    if (rc == GIQL_OK && hipMemcpy(hb, d_b, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)
      rc = set_err(GIQL_ERR_HIP, "D2H copy of row_b failed");
    if (rc != GIQL_OK) {
      giql_hip_free_host(ha);
      giql_hip_free_host(hb);
      return rc;
    }
  }
  ms_d2h = ms_since(t0);
  if (dbg)
    fprintf(stderr, "[giql_hip_inner] H2D %.1f ms, plan %.1f, pinned alloc %.1f, output alloc + fill %.1f, %s %.1f (%lld pairs)\n",
            ms_h2d, ms_plan, ms_pin, ms_fill, compact_done ? "compact plan D2H + host expansion" : "D2H", ms_d2h, (long long)n);
  *row_a = ha;
  *row_b = hb;
  return GIQL_OK;
}

int giql_hip_semi_anti(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                       int anti, int64_t* n_out, int32_t** rows_a) {
  if (!ctx || !n_out || !rows_a) return set_err(GIQL_ERR_INVALID, "NULL argument");
  GIQL_TRY(check_side(a, "a"));
  GIQL_TRY(check_side(b, "b"));
  HIP_TRY(hipSetDevice(ctx->device));
  *rows_a = nullptr;
  DevSide da, db;
  GIQL_TRY(upload_side(a, da));
  GIQL_TRY(upload_side(b, db));
  DevBuf out;
  const size_t na = (size_t)a->n;
  HIP_TRY(hipMalloc(&out.p, (na ? na : 1) * sizeof(int32_t)));
  int64_t n = 0;
  GIQL_TRY(giql_hip_semi_anti_dev(ctx, &da.s, &db.s, n_chrom, anti, (int32_t*)out.p, &n, nullptr));
  int32_t* h = (int32_t*)host_alloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  if (!h) return set_err(GIQL_ERR_NOMEM, "out of host memory");
  if (n > 0 && hipMemcpy(h, out.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
    giql_hip_free_host(h);
    return set_err(GIQL_ERR_HIP, "D2H copy failed");
  }
  *n_out = n;
  *rows_a = h;
  return GIQL_OK;
}

int giql_hip_count(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                   int64_t* counts_out) {
  if (!ctx) return set_err(GIQL_ERR_INVALID, "NULL argument");
  GIQL_TRY(check_side(a, "a"));
  GIQL_TRY(check_side(b, "b"));
  if (a->n == 0) return GIQL_OK;
  if (!counts_out) return set_err(GIQL_ERR_INVALID, "counts_out is NULL");
  HIP_TRY(hipSetDevice(ctx->device));
  DevSide da, db;
  GIQL_TRY(upload_side(a, da));
  GIQL_TRY(upload_side(b, db));
  DevBuf out;
  HIP_TRY(hipMalloc(&out.p, (size_t)a->n * sizeof(int64_t)));
  GIQL_TRY(giql_hip_count_dev(ctx, &da.s, &db.s, n_chrom, (int64_t*)out.p, nullptr));
  HIP_TRY(hipMemcpy(counts_out, out.p, (size_t)a->n * sizeof(int64_t), hipMemcpyDeviceToHost));
  return GIQL_OK;
}

int giql_hip_nearest(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b, int32_t n_chrom,
                     int is_signed, int64_t max_distance, int32_t* idx_b_out, int64_t* dist_out) {
  if (!ctx) return set_err(GIQL_ERR_INVALID, "NULL argument");
  GIQL_TRY(check_side(a, "a"));
  GIQL_TRY(check_side(b, "b"));
  if (a->n == 0) return GIQL_OK;
  if (!idx_b_out || !dist_out) return set_err(GIQL_ERR_INVALID, "output is NULL");
  HIP_TRY(hipSetDevice(ctx->device));
  DevSide da, db;
  GIQL_TRY(upload_side(a, da));
  GIQL_TRY(upload_side(b, db));
  DevBuf oi, od;
  HIP_TRY(hipMalloc(&oi.p, (size_t)a->n * sizeof(int32_t)));
  HIP_TRY(hipMalloc(&od.p, (size_t)a->n * sizeof(int64_t)));
  GIQL_TRY(giql_hip_nearest_dev(ctx, &da.s, &db.s, n_chrom, is_signed, max_distance,
                                (int32_t*)oi.p, (int64_t*)od.p, nullptr));
  HIP_TRY(hipMemcpy(idx_b_out, oi.p, (size_t)a->n * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(dist_out, od.p, (size_t)a->n * sizeof(int64_t), hipMemcpyDeviceToHost));
  return GIQL_OK;
}

void giql_hip_free_host(void* p) {
  if (p) host_pool().put(p);
}

#if defined(GIQL_OS_TIMELINE)
// diagnostic builds only (tools/os_timeline.py): the phase stamps of the last sort pass
int giql_hip_debug_timeline(unsigned long long* out, int64_t n_words) {
  const size_t cap = (size_t)OS_TL_TILES * 16;
  const size_t n = (size_t)n_words < cap ? (size_t)n_words : cap;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_os_tl), n * sizeof(unsigned long long)));
  return GIQL_OK;
}
#endif

}  // extern "C"
