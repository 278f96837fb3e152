/*
 * include/giql_hip.h -- C ABI of libgiql_hip.so, the MI355X (gfx950) execution
 * backend for GIQL's column-to-column INTERSECTS range join (+ SEMI / ANTI,
 * per-row COUNT and NEAREST k=1).
 *
 * This is the drop-in boundary.  In the reference the path ends with a SQL string
 * handed to an engine; what the engine (DuckDB IE_JOIN) computes is what these
 * entry points compute, on the same inputs:
 *
 *   reference interface replaced (path:line under /root/reference/)
 *   ----------------------------------------------------------------
 *   giql_hip_inner_*     per-chromosome INNER IEJoin plan + UNION ALL
 *                        src/giql/expanders/intersects_duckdb.py:1283-1299,
 *                        1317-1330, 1336-1400; _per_chrom.py:46-74; predicate
 *                        src/giql/expanders/intersects.py:149-154
 *   giql_hip_semi_anti_* SEMI / ANTI (WHERE [NOT] EXISTS) plan
 *                        src/giql/expanders/intersects_duckdb.py:1254-1282,
 *                        1321-1324
 *   giql_hip_count_*     count_overlaps (COUNT per left row, zero-filled)
 *                        src/giql/expanders/intersects_duckdb.py:432-548, 806-854
 *   giql_hip_nearest_*   NEAREST k=1 (LATERAL top-k subquery + distance CASE)
 *                        src/giql/expanders/nearest.py:255-397;
 *                        src/giql/expanders/_distance.py:67-87
 *   giql_side.start_off / end_off
 *                        canonical_start / canonical_end
 *                        src/giql/canonical.py:16-52
 *
 * Conventions
 *   - C linkage, plain pointers and sizes, no C++ / torch types.
 *   - Every entry returns an int status: 0 = ok, < 0 = error; the message is
 *     available from giql_hip_last_error() (thread-local).  A context belongs to one thread at
 *     a time; different contexts may run on different threads and streams at once (the
 *     pinned-output pool is the only shared state, and it is locked).
 *   - Column buffers are BORROWED Arrow int32 data buffers (validity must be
 *     all-valid, offset already applied); the library never writes or frees them.
 *   - chrom is a dictionary id in [0, n_chrom) from a dictionary SHARED by both
 *     sides (the reference compares VARCHAR values).
 *   - "_dev" entry points take DEVICE pointers and a hipStream_t (as void*; NULL
 *     = the default stream); results stay on the device.  The host-buffer entry
 *     points stage H2D/D2H themselves and return PINNED host arrays
 *     (hipHostMalloc: the D2H copy runs at link speed) that the caller releases
 *     with giql_hip_free_host() -- not with free().  Page-locking costs more than the
 *     copy (3.2 GB: ~200 ms against ~60 ms), so released arrays are kept in a small
 *     process-wide pool for the next call (GIQL_HIP_HOST_POOL_MB of idle memory at most,
 *     default 8192; 0 = release at once).
 *   - caller-owned DEVICE outputs (row_a / row_b of the fill) should start on a
 *     128-byte boundary each: the fill stores 256 bytes per wave instruction, and a
 *     row that begins inside a cache line makes every store touch three lines
 *     instead of two (measured 0.73-0.80 -> 0.70 ms on the 404M-pair fill).
 *   - Output order is unspecified (as upstream); pairs are a
 *     multiset (bag semantics).
 */
#ifndef GIQL_HIP_H
#define GIQL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIQL_HIP_ABI_VERSION 2  /* 2: giql_hip_stats.phase_bytes, pinned host outputs, plan export */

enum {
  GIQL_OK = 0,
  GIQL_ERR_INVALID = -1,   /* bad argument */
  GIQL_ERR_HIP = -2,       /* HIP runtime failure */
  GIQL_ERR_NOMEM = -3,     /* allocation failure */
  GIQL_ERR_CHROM = -4,     /* chrom id outside [0, n_chrom) */
  GIQL_ERR_SPAN = -5,      /* linearised coordinate space exceeds 32 bits */
  GIQL_ERR_CAPACITY = -6,  /* caller-provided output too small */
  GIQL_ERR_STATE = -7      /* fill without a successful plan */
};

typedef struct giql_hip_ctx giql_hip_ctx;

/* One join side (reference: the (chrom,start,end) columns a ResolvedColumn
 * names, src/giql/resolver.py:256-299, with its table's encoding). */
typedef struct giql_side {
  const int32_t* chrom;
  const int32_t* start;
  const int32_t* end;
  int64_t n;
  int32_t start_off; /* canonical start = start + start_off (0 or -1)       */
  int32_t end_off;   /* canonical end   = end   + end_off   (+1, 0 or -1)   */
} giql_side;

/* Phase indices of giql_hip_stats.phase_ms (hipEvent-timed on the call's
 * stream when profiling is enabled). */
enum {
  GIQL_PH_SPAN = 0,      /* per-chromosome min/max + offsets                */
  GIQL_PH_LINEARIZE = 1, /* (chrom,start,end) -> 32-bit linear keys         */
  GIQL_PH_SORT_HIST = 2, /* radix: per-tile digit histograms                */
  GIQL_PH_SORT_SCAN = 3, /* radix: scan of tile histograms                  */
  GIQL_PH_SORT_SCATTER = 4, /* radix: ranked scatter of (key,end,rid)       */
  GIQL_PH_COUNT = 5,     /* per-row range bounds + counts                   */
  GIQL_PH_SCAN = 6,      /* exclusive scan of counts -> output offsets      */
  GIQL_PH_PARTITION = 7, /* output-tile -> row partition                    */
  GIQL_PH_FILL = 8,      /* pair materialisation                            */
  GIQL_PH_IRREGULAR = 9, /* literal-predicate path for end<=start rows      */
  GIQL_PH_AUX = 10,      /* prefix-max / compaction / nearest kernels       */
  GIQL_PH_SORT_LOCAL = 11, /* three-stage sort: bucket bounds + in-LDS bucket sort */
  GIQL_PH_N = 16
};

typedef struct giql_hip_stats {
  int64_t n_a, n_b;
  int64_t n_out;            /* pairs / rows produced by the last call       */
  int64_t n_irregular_a;    /* rows with canonical end <= start             */
  int64_t n_irregular_b;
  int64_t workspace_bytes;  /* device arena size                            */
  int64_t span;             /* linearised coordinate span                   */
  float phase_ms[GIQL_PH_N];
  int32_t phase_launches[GIQL_PH_N];
  int64_t phase_bytes[GIQL_PH_N]; /* ALGORITHMIC bytes of the phase's launches (read every input
                               of a kernel once + write every output once), accounted by the
                               host code that issues them for the sort phases (SORT_SCATTER,
                               SORT_LOCAL); 0 = not accounted */
  float total_ms;           /* sum of phase_ms                              */
  int32_t profiled;         /* 1 if phase_ms are valid                      */
  int32_t reserved;         /* bits 0-3: INNER join form (0 = general two-class join,
                               1 / 2 = uniform-length form with B / A as the fixed-
                               length side); bit 4: that side was sorted straight from
                               its raw columns (digit histogram in the span pass) -- after SEMI / ANTI /
                               COUNT: the fixed-length B was sorted without its lowest digit (three
                               passes), equal upper 24 bits looked at row by row; bit 5: a side
                               was sorted in three stages (two global passes + the in-LDS
                               bucket sort); bit 6: the context fell back to the four-pass
                               sort (a bucket too large for LDS); bit 7: the INNER plan ran with
                               the sides exchanged (the larger table planned as B; pairs, stats and
                               the exported plan are in the caller's labels all the same);
                               bits 8-14: sort tile order in force (2 =
                               blockIdx order, 0 = ticket order); bit 15: the range count of the
                               fixed-length form ran inside the bucket sort (no count kernel, the
                               sorted keys never stored); bits 16-26: calls
                               repeated in ticket order after a look-back timeout; bits 27-28:
                               16 - the key bits of a bucket of the last three-stage sort (0: buckets
                               of 65,536 keys; 1-3: 2^15 / 2^14 / 2^13 keys after THREE global
                               passes -- tables past ~2,800 rows per 65,536 positions); bit 29: the
                               bucket stage wrote the pairs itself (one-call form: no sorted id,
                               bound or offset array was stored, no scan and no fill kernel ran);
                               bit 30: the last plan launched its own fill
                               (giql_hip_inner_join_dev); bit 31: a side arrived in (chrom id, start)
                               order and skipped its sort */
} giql_hip_stats;

/* ---- library / context ------------------------------------------------- */
int giql_hip_abi_version(void);
const char* giql_hip_last_error(void);
int giql_hip_device_count(int* n_devices);
int giql_hip_create(int device, giql_hip_ctx** out);
int giql_hip_destroy(giql_hip_ctx* ctx);
/* Pre-size the device arena (bytes); optional, the arena grows on demand. */
int giql_hip_reserve(giql_hip_ctx* ctx, int64_t bytes);
/* enabled: 0 = off, 1 = hipEvent pairs around every phase, 2 = only around the sort passes,
 * 16 + p = only around phase p (GIQL_PH_*) -- an event pair costs the stream a few
 * microseconds of idle time per phase. */
int giql_hip_set_profiling(giql_hip_ctx* ctx, int enabled);
int giql_hip_get_stats(giql_hip_ctx* ctx, giql_hip_stats* out);

/* ---- device-resident entry points -------------------------------------- */
/* INNER join in two calls so the caller owns the output:
 *   plan: sort + count + scan; returns the exact number of pairs;
 *   fill: writes row_a[i], row_b[i] for i < n_pairs (capacity >= n_pairs). */
int giql_hip_inner_plan_dev(giql_hip_ctx* ctx, const giql_side* a,
                            const giql_side* b, int32_t n_chrom, void* stream,
                            int64_t* n_pairs);
int giql_hip_inner_fill_dev(giql_hip_ctx* ctx, int32_t* row_a, int32_t* row_b,
                            int64_t capacity, void* stream);

/* Plan + fill in one call into caller-owned buffers (capacity pairs each).  When the
 * context's guesses hold -- same join form as its previous plan, no irregular
 * rows, the pairs fit -- the fill is launched inside the plan with no stream sync
 * in between.  GIQL_ERR_CAPACITY leaves the plan valid: *n_pairs is the size to
 * offer to giql_hip_inner_fill_dev.
 * In the fixed-length form with a table of 32M rows and more the pairs may be written
 * by the sort's last stage itself (stats.reserved bit 29): nothing but the pairs leaves
 * that call, so giql_hip_inner_fill_dev / giql_hip_inner_plan_export_dev after it
 * return GIQL_ERR_STATE -- they follow a giql_hip_inner_plan_dev.  The ORDER of the
 * pairs is unspecified in every form (and not reproducible from call to call in this
 * one); the multiset of pairs is exact. */
int giql_hip_inner_join_dev(giql_hip_ctx* ctx, const giql_side* a,
                            const giql_side* b, int32_t n_chrom, int32_t* row_a,
                            int32_t* row_b, int64_t capacity, void* stream,
                            int64_t* n_pairs);

/* SEMI (anti = 0) / ANTI (anti = 1): A row ids with / without an overlapping
 * B row.  rows_out has capacity a->n; *n_out receives the count. */
int giql_hip_semi_anti_dev(giql_hip_ctx* ctx, const giql_side* a,
                           const giql_side* b, int32_t n_chrom, int anti,
                           int32_t* rows_out, int64_t* n_out, void* stream);

/* Overlap count per A row (counts_out[a->n], original row order). */
int giql_hip_count_dev(giql_hip_ctx* ctx, const giql_side* a,
                       const giql_side* b, int32_t n_chrom,
                       int64_t* counts_out, void* stream);

/* NEAREST k=1.  idx_b_out[i] = nearest B row of A row i (-1: none on that
 * chromosome / within max_distance), dist_out[i] = distance (signed when
 * is_signed).  max_distance < 0 = unlimited.  Requires start <= end rows. */
int giql_hip_nearest_dev(giql_hip_ctx* ctx, const giql_side* a,
                         const giql_side* b, int32_t n_chrom, int is_signed,
                         int64_t max_distance, int32_t* idx_b_out,
                         int64_t* dist_out, void* stream);

/* The same with the output SURVEY.md section 8 (a9) sizes: one {idx_b, distance}
 * int32 record per A row, idx_dist_out[2 * i] = nearest B row (-1: none),
 * idx_dist_out[2 * i + 1] = distance -- 8 bytes per row, written straight to
 * their place (no second pass over the result).  GIQL_ERR_INVALID when a distance
 * does not fit int32 (coordinates near the ends of the int32 range): the int64
 * entry point above is the one for such data.  8-byte aligned.  Replaces the
 * same reference lines (src/giql/expanders/nearest.py:336-397). */
int giql_hip_nearest32_dev(giql_hip_ctx* ctx, const giql_side* a,
                           const giql_side* b, int32_t n_chrom, int is_signed,
                           int64_t max_distance, int32_t* idx_dist_out,
                           void* stream);

/* Per-chromosome coordinate span (max - min + 1 over both sides' canonical
 * coordinates, 0 for an absent chromosome) into the HOST array spans_out[n_chrom].
 * The join entry points place all chromosomes on one 32-bit axis and return
 * GIQL_ERR_SPAN when the spans sum past 2^32 - 1; callers then split the
 * chromosomes into groups that fit (chromosomes are independent units of the
 * join, src/giql/expanders/_per_chrom.py:3-9) and join group by group. */
int giql_hip_chrom_spans_dev(giql_hip_ctx* ctx, const giql_side* a,
                             const giql_side* b, int32_t n_chrom,
                             int64_t* spans_out /* host */, void* stream);

/* ---- CLUSTER / MERGE (the sort + scan operators next to the join) ---------
 * CLUSTER: src/giql/expanders/cluster.py:210-300 -- per partition (s->chrom; the
 * caller folds the strand into the id when stranded), rows ordered by start,
 *   is_new = NOT (running MAX(end) of the preceding rows + distance >= start),
 *   cluster_id = SUM(is_new) OVER (PARTITION BY chrom ORDER BY start),
 * written per input row (1-based within the partition) to cluster_id_out[s->n].
 * The window SQL reads the RAW columns: start_off / end_off must be 0.  Rows
 * need start <= end (GIQL_ERR_INVALID otherwise). */
int giql_hip_cluster_dev(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom,
                         int64_t distance, int64_t* cluster_id_out, void* stream);
/* CLUSTER(interval[, d], predicate := <conjunction of comparisons over columns and PREV(col)>),
 * src/giql/expanders/cluster.py:281-296, 587-640: a row stays in the running cluster only when
 * it is adjacent AND every predicate holds between it and its immediate predecessor in the
 * partition's start order (the LAG the reference emits).  giql_pred (declared below, with the
 * residual predicates) reads the current row through operand side A and the predecessor through
 * side B -- both DEVICE columns of s->n rows, addressed by row id; a NULL operand starts a new
 * cluster (the CASE's ELSE arm).  Rows of equal start keep their input order. */
struct giql_pred;
int giql_hip_cluster_pred_dev(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom,
                              int64_t distance, const struct giql_pred* preds, int32_t n_preds,
                              int64_t* cluster_id_out, void* stream);
/* MERGE: src/giql/expanders/merge.py:186-330 -- GROUP BY chrom, cluster id ->
 * chrom, MIN(start), MAX(end) [, COUNT(*) when out_count != NULL], ordered by
 * (chrom, start).  Outputs have `capacity` entries (s->n always suffices);
 * *n_out receives the number of merged regions. */
int giql_hip_merge_dev(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom,
                       int64_t distance, int32_t* out_chrom, int32_t* out_start,
                       int32_t* out_end, int64_t* out_count, int64_t capacity,
                       int64_t* n_out, void* stream);
/* MERGE(..., predicate := ...): merge.py:201-210 hands the predicate to the CLUSTER it is built on, so a
 * merged region is a cluster of giql_hip_cluster_pred_dev; its MAX(end) is taken over the region's own
 * rows (a region may end while an earlier one still reaches further). */
int giql_hip_merge_pred_dev(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom,
                            int64_t distance, const struct giql_pred* preds, int32_t n_preds,
                            int32_t* out_chrom, int32_t* out_start, int32_t* out_end,
                            int64_t* out_count, int64_t capacity, int64_t* n_out, void* stream);

/* ---- the aggregate half of count_overlaps --------------------------------
 * GROUP BY the left interval + SUM of the per-row counts
 * (src/giql/expanders/intersects_duckdb.py:806-854; a key held by k duplicate left
 * rows counts k times its overlaps, tests/test_duckdb_iejoin.py:66-81).
 * group_rows: rows with identical (chrom, raw start, raw end) share a group;
 * group_of_row[i] in [0, *n_groups) per input row, rep_row[g] = one row id of
 * group g (both device, s->n entries).  segment_sum: sums[g] = sum of values[i]
 * over the rows of group g (sums: device, n_groups entries, zeroed here). */
int giql_hip_group_rows_dev(giql_hip_ctx* ctx, const giql_side* s, int32_t n_chrom,
                            int32_t* group_of_row, int32_t* rep_row,
                            int64_t* n_groups, void* stream);
int giql_hip_segment_sum_dev(giql_hip_ctx* ctx, const int64_t* values,
                             const int32_t* group_of_row, int64_t n, int64_t* sums,
                             int64_t n_groups, void* stream);

/* ---- projection materialisation (Arrow `take` by the join's row ids) ------
 * Replaces the outer SELECT that rebuilds the projected columns of both sides
 * around the per-chromosome join, src/giql/expanders/intersects_duckdb.py:
 * 1402-1644.  All pointers are DEVICE pointers except
 * the small argument arrays (cols / elem_bytes / outs), which are host arrays.
 *
 * Fixed-width columns: outs[c][i] = cols[c][idx[i]] for i < n, elem_bytes[c] in
 * {1,2,4,8,16}; every column has n_rows rows.  idx[i] < 0 (NEAREST's "none")
 * yields zero bytes; idx[i] >= n_rows is GIQL_ERR_INVALID. */
int giql_hip_take_dev(giql_hip_ctx* ctx, const void* const* cols,
                      const int32_t* elem_bytes, int32_t n_cols, int64_t n_rows,
                      const int32_t* idx, int64_t n, void* const* outs,
                      void* stream);
/* utf8 / binary columns (Arrow int32 offsets[n_rows + 1] + data bytes), in two
 * calls so the caller owns the output: plan writes out_offsets[n + 1] and
 * returns the byte count; fill copies the bytes into out_data[n_bytes]. */
int giql_hip_take_utf8_plan_dev(giql_hip_ctx* ctx, const int32_t* offsets,
                                int64_t n_rows, const int32_t* idx, int64_t n,
                                int32_t* out_offsets, int64_t* n_bytes,
                                void* stream);
int giql_hip_take_utf8_fill_dev(giql_hip_ctx* ctx, const int32_t* offsets,
                                const uint8_t* data, int64_t n_rows,
                                const int32_t* idx, int64_t n,
                                const int32_t* out_offsets, uint8_t* out_data,
                                void* stream);

/* ---- residual predicates (extra ON / WHERE conjuncts beside the INTERSECTS) --
 * The reference inlines them into the per-chromosome join's ON clause,
 * src/giql/expanders/intersects_duckdb.py:1164-1177, 1239-1243 (SURVEY.md
 * section 8f-3).  Here a predicate is `lhs op rhs`; an operand is a payload
 * column of side A / side B (DEVICE pointer, addressed through the candidate's
 * row id) or a literal.  Integers compare as int64, anything involving a float
 * as double; a NULL operand (valid[row] == 0) makes the predicate not true. */
enum { GIQL_OP_EQ = 0, GIQL_OP_NE = 1, GIQL_OP_LT = 2, GIQL_OP_LE = 3, GIQL_OP_GT = 4, GIQL_OP_GE = 5,
       GIQL_OP_IS_NULL = 6, GIQL_OP_NOT_NULL = 7 /* unary: lhs only, rhs ignored */,
       GIQL_OP_IS_TRUE = 8 /* unary: lhs is a boolean program (GIQL_X_EQ .. GIQL_X_NOT nodes), kept when TRUE */ };
enum { GIQL_T_I32 = 0, GIQL_T_I64 = 1, GIQL_T_F32 = 2, GIQL_T_F64 = 3, GIQL_T_U8 = 4 };
enum { GIQL_SIDE_A = 0, GIQL_SIDE_B = 1, GIQL_SIDE_LIT = 2,
       GIQL_SIDE_EXPR = 3 /* an arithmetic expression: giql_hip_select_expr_dev */ };
/* node kinds of an expression program (giql_operand.side of a node): 0 / 1 / 2 push a column value / a literal */
enum { GIQL_X_ADD = 16, GIQL_X_SUB = 17, GIQL_X_MUL = 18, GIQL_X_DIV = 19, GIQL_X_NEG = 20, GIQL_X_ABS = 21,
       GIQL_X_LEAST = 22, GIQL_X_GREATEST = 23,
       /* boolean nodes over three-valued results (round 4): a whole condition as one program, the way the
        * reference inlines it as text (src/giql/expanders/intersects_duckdb.py:889-957) -- no normal form */
       GIQL_X_EQ = 24, GIQL_X_NE = 25, GIQL_X_LT = 26, GIQL_X_LE = 27, GIQL_X_GT = 28, GIQL_X_GE = 29,
       GIQL_X_ISNULL = 30, GIQL_X_NOTNULL = 31, GIQL_X_AND = 32, GIQL_X_OR = 33, GIQL_X_NOT = 34 };

typedef struct giql_operand {
  int32_t side;          /* GIQL_SIDE_*                                          */
  int32_t type;          /* GIQL_T_* of the column (unused for a literal)        */
  const void* data;      /* device column                                        */
  const uint8_t* valid;  /* device byte-per-row validity, NULL = all valid       */
  int64_t lit_i;         /* literal: integer value ...                           */
  double lit_f;          /* ... or floating value when lit_is_float              */
  int32_t lit_is_float;
  int32_t reserved;
} giql_operand;

typedef struct giql_pred {
  giql_operand lhs, rhs;
  int32_t op;            /* GIQL_OP_*                                            */
  int32_t group;         /* 0: a conjunct of its own; g != 0: OR-ed with the     */
                         /* neighbouring predicates that carry the same g        */
} giql_pred;

/* The reference inlines ANY residual expression as SQL text (_classify_extras,
 * intersects_duckdb.py:889-912: only a nested INTERSECTS, sub-queries, aggregates
 * and window functions fall back).  Here the caller hands over the expression in
 * conjunctive normal form: predicates are AND-ed, except that a run of adjacent
 * predicates sharing one non-zero `group` forms ONE clause whose members are OR-ed.
 * NOT is pushed into the comparisons by the caller (NOT (x < y) = x >= y, De Morgan
 * for AND / OR; exact under SQL's three-valued logic because a filter keeps TRUE
 * only and AND / OR are monotone), BETWEEN and IN (list) are spelt as comparisons.
 *
 * Stable filter of n candidates by n_preds (<= 16) predicates.
 * Candidate i addresses side A by idx_a[i] (i itself when idx_a is NULL) and side
 * B by idx_b[i] likewise; the kept candidates' ids are written, in input order,
 * to out_a / out_b (capacity n each; either may be NULL) and *n_kept receives
 * their number.  With idx_a / idx_b = the join's pairs this is the post-join
 * residual filter; with both NULL it filters the rows of one table. */
int giql_hip_select_dev(giql_hip_ctx* ctx, const giql_pred* preds, int32_t n_preds,
                        const int32_t* idx_a, int64_t n_rows_a,
                        const int32_t* idx_b, int64_t n_rows_b, int64_t n,
                        int32_t* out_a, int32_t* out_b, int64_t* n_kept,
                        void* stream);
/* The same with ARITHMETIC operands -- the overlap-fraction recipes of docs/recipes/intersect.rst:144-190
 * ("(LEAST(a.end, b.end) - GREATEST(a.start, b.start)) >= 0.5 * (a.end - a.start)"), which the reference
 * inlines into its join's ON clause as text (intersects_duckdb.py:889-912, 1239-1243).  An operand of side
 * GIQL_SIDE_EXPR is a postfix program over `nodes` (a HOST array of n_nodes <= 256 giql_operand-shaped
 * nodes): lit_i = its first node, type = its node count.  A node of side A / B / LIT pushes that value, a node
 * of side GIQL_X_* pops its one (NEG, ABS, ISNULL, NOTNULL, NOT) or two arguments and pushes the result; at
 * most 12 values are live.  Semantics are those of the reference's execution target (DuckDB): integer + - *
 * stay 64-bit integers, `/` is a floating division and NULL on a zero divisor, NULL propagates through
 * arithmetic, LEAST / GREATEST skip NULL arguments; comparisons and AND / OR / NOT are three-valued (a boolean
 * program is the lhs of a predicate with op GIQL_OP_IS_TRUE: the candidate is kept when it is TRUE). */
int giql_hip_select_expr_dev(giql_hip_ctx* ctx, const giql_pred* preds, int32_t n_preds,
                             const giql_operand* nodes, int32_t n_nodes,
                             const int32_t* idx_a, int64_t n_rows_a,
                             const int32_t* idx_b, int64_t n_rows_b, int64_t n,
                             int32_t* out_a, int32_t* out_b, int64_t* n_kept,
                             void* stream);
/* flags[idx[i]] = 1 for i < n (flags: device, n_rows bytes, caller-initialised):
 * the left rows that keep at least one pair, for SEMI / ANTI with two-sided
 * residuals (src/giql/expanders/intersects_duckdb.py:1254-1282). */
int giql_hip_mark_dev(giql_hip_ctx* ctx, const int32_t* idx, int64_t n,
                      uint8_t* flags, int64_t n_rows, void* stream);

/* ---- host-buffer entry points (Arrow buffers in host memory) ------------
 * giql_hip_inner: columns of both tables in host memory in, the pairs out in host arrays the library
 * owns (giql_hip_free_host; the order of the pairs is unspecified, as an INNER join's always is).
 * Since round 4 a large result whose plan has the compact form (one table of fixed length, no
 * irregular row) comes down as that PLAN -- per query row {row id, first match, count} + the other
 * side's row ids in sorted order, 0.52 GB instead of 3.2 GB at 10M x 100M -- and is expanded into
 * plain (not page-locked) memory by host threads while the ids still arrive: 71 -> 40-47 ms, first
 * call 230-345 -> 78 ms (taken on hosts with 16 hardware threads and more).  GIQL_HIP_E2E_COMPACT=0 / 1:
 * never / whenever the form allows;
 * GIQL_HIP_E2E_THREADS (default 32).  Otherwise the pairs themselves are downloaded into page-locked
 * arrays, and a table of 8M rows and more is uploaded in blocks of 4M rows
 * (GIQL_HIP_E2E_BLOCK_ROWS; 0 = one shot): an INNER join is the union of the joins of its row
 * blocks, so block j's pairs travel to the host while block j + 1 travels to the device.
 * Reference counterpart: conn.execute(sql).arrow() around the per-chromosome plan of
 * src/giql/expanders/intersects_duckdb.py:1283-1330 (host tables in, host result out). */
int giql_hip_inner(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b,
                   int32_t n_chrom, int64_t* n_pairs, int32_t** row_a,
                   int32_t** row_b);
int giql_hip_semi_anti(giql_hip_ctx* ctx, const giql_side* a,
                       const giql_side* b, int32_t n_chrom, int anti,
                       int64_t* n_out, int32_t** rows_a);
int giql_hip_count(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b,
                   int32_t n_chrom, int64_t* counts_out /* host, a->n */);
int giql_hip_nearest(giql_hip_ctx* ctx, const giql_side* a, const giql_side* b,
                     int32_t n_chrom, int is_signed, int64_t max_distance,
                     int32_t* idx_b_out /* host */, int64_t* dist_out /* host */);
void giql_hip_free_host(void* p);

/* Order-independent 64-bit checksum of a device-resident pair multiset (same
 * function as the oracle's, for full-size parity checks). */
int giql_hip_pairs_checksum_dev(giql_hip_ctx* ctx, const int32_t* row_a,
                                const int32_t* row_b, int64_t n, void* stream,
                                uint64_t* out);

/* NEAREST k >= 1 (src/giql/expanders/nearest.py:240-252, 336-397): for A row i the k nearest B
 * rows on its chromosome in the reference's order ABS(distance), start, end land in
 * idx_b_out[i*k .. i*k+k) / dist_out[i*k ..) (row-major; unused slots -1 / 0), within
 * max_distance when >= 0.  1 <= k <= 2^20 and n_a * k < 2^31.  Rows tied on (distance, start, end) are
 * order-ambiguous upstream too (nearest.py:366-372).  Requires start <= end rows. */
int giql_hip_nearest_k_dev(giql_hip_ctx* ctx, const giql_side* a,
                           const giql_side* b, int32_t n_chrom, int32_t k,
                           int is_signed, int64_t max_distance,
                           int32_t* idx_b_out, int64_t* dist_out, void* stream);

/* ---- compact plan: the multi-GPU exchange ------------
 * Replaces the reference's UNION ALL over per-chromosome branches
 * (src/giql/expanders/_per_chrom.py:46-74) across devices.  After a successful
 * giql_hip_inner_plan_dev in the single-range (uniform-length) form with no
 * irregular rows, the plan is a compact description of the pairs: per query
 * row {row id, first matching position, match count} + the other side's row ids
 * in sorted order.  Export copies it into caller buffers, adding rid_add_a /
 * rid_add_b to the A / B row ids on the way (shard-local -> global ids);
 * *query_is_a tells which side the query rows are.  GIQL_ERR_STATE when the
 * last plan has another form (the caller then exchanges the pairs themselves),
 * GIQL_ERR_CAPACITY (with *n_q / *n_s set) when a buffer is short. */
int giql_hip_inner_plan_export_dev(giql_hip_ctx* ctx, int32_t* q_rid_out,
                                   uint32_t* lo_out, uint32_t* cnt_out,
                                   int32_t* s_rid_out, int64_t q_capacity,
                                   int64_t s_capacity, int32_t rid_add_a,
                                   int32_t rid_add_b, int32_t* query_is_a,
                                   int64_t* n_q, int64_t* n_s, void* stream);
/* Expand a compact plan (possibly another device's) into index pairs: pair k of
 * query row i is (q_rid[i], s_rid[lo[i] + k]), k < cnt[i]; row_q receives the
 * query side's ids.  n_pairs_expected >= 0: the caller knows the pair count
 * (e.g. from an all-gather of counts) and no read-back / stream sync happens;
 * -1: the count is read back.  The kernels take the count from the device
 * either way, so a wrong expectation cannot overrun `capacity`. */
int giql_hip_fill_from_plan_dev(giql_hip_ctx* ctx, const int32_t* q_rid,
                                const uint32_t* lo, const uint32_t* cnt,
                                int64_t n_q, const int32_t* s_rid, int64_t n_s,
                                int32_t* row_q, int32_t* row_s, int64_t capacity,
                                int64_t n_pairs_expected, void* stream,
                                int64_t* n_pairs);

/* Streaming-copy rate of this device with the library's access pattern (16 B
 * per lane), bytes read + written per second / 1e9: the measured yardstick
 * bench.py reports next to the 8 TB/s HBM peak. */
int giql_hip_copy_probe_dev(giql_hip_ctx* ctx, const void* src, void* dst,
                            int64_t bytes, int32_t reps, void* stream,
                            double* gbytes_per_s);

/* ---- table index (round 4) -------------------------------------------------
 * The reference tells its users to CREATE INDEX ... (chrom, start, "end") on both
 * join sides (docs/transpilation/performance.rst:111-130): what the engine keeps
 * between queries.  Here the properties of a TABLE that every join over it would
 * recompute -- chromosome bases on the linear axis, the fixed length, the
 * (key, row id) rows grouped by 65,536-key bucket and sorted -- live in HBM as an
 * explicit object (8 bytes per row, 12 for tables of variable length).  The
 * table's columns are read once, at creation; the index keeps no pointer to them.
 * GIQL_ERR_STATE when the table does not take the indexed form (more than 32
 * chromosomes, a negative coordinate, irregular rows, more than ~2800 rows per
 * 8,192 positions -- denser tables than 2800 per 65,536 are indexed with buckets
 * of 2^15 / 2^14 / 2^13 keys): the ordinary join serves such tables. */
typedef struct giql_hip_index giql_hip_index;
int giql_hip_index_create_dev(giql_hip_ctx* ctx, const giql_side* side,
                              int32_t n_chrom, void* stream, giql_hip_index** out);
int giql_hip_index_destroy(giql_hip_index* idx);
int giql_hip_index_info(const giql_hip_index* idx, int64_t* n_rows, int64_t* bytes,
                        int32_t* general, int64_t* span);
/* INNER join of `a` against an indexed table (per-chromosome INNER plan,
 * src/giql/expanders/intersects_duckdb.py:1283-1330, with one side indexed as
 * performance.rst advises): pair k = (row_a[k] of a, row_idx[k] of the indexed
 * table), same bag semantics and unspecified order as giql_hip_inner_join_dev.
 * `a`'s chrom ids speak the INDEXED table's dictionary (ids >= its n_chrom
 * match nothing).  Per call: a's keys on the index's axis, its sort, and the
 * bucket stage over the index -- the indexed table's span pass and global sort
 * passes are not repeated.  GIQL_ERR_STATE: `a` holds irregular rows or rows
 * longer than 32768 positions (use the ordinary join); GIQL_ERR_CAPACITY with
 * *n_pairs set when the buffers are short (nothing useful was written). */
int giql_hip_inner_join_indexed_dev(giql_hip_ctx* ctx, const giql_hip_index* idx,
                                    const giql_side* a, int32_t* row_a,
                                    int32_t* row_idx, int64_t capacity,
                                    void* stream, int64_t* n_pairs);

/* What this device reads / writes / copies per second, by access shape (round 4:
 * the measured ceiling the kernels are held against; SURVEY.md section 8d "verify
 * on the box with a copy kernel").  mode 0 = read only, 1 = write only, 2 = copy,
 * 3 = hipMemcpyDtoDAsync (an outside reference); in_flight = 16-byte accesses a
 * thread keeps in flight (1 / 2 / 4 / 8); nontemporal = nt loads and stores;
 * blocks_per_cu sizes the grid of 256-thread blocks.  *gbytes_per_s counts every
 * byte moved (read for mode 0, written for mode 1, both for 2 / 3), / 1e9. */
int giql_hip_stream_probe_dev(giql_hip_ctx* ctx, const void* src, void* dst,
                              int64_t bytes, int32_t mode, int32_t in_flight,
                              int32_t nontemporal, int32_t blocks_per_cu,
                              int32_t reps, void* stream, double* gbytes_per_s);

/* The host-buffer entry points (giql_hip_inner & co.) keep released output
 * buffers (page-locked ones, and the plain ones of the compact-plan path) for the next call (up to GIQL_HIP_HOST_POOL_MB, default 8192).
 * This returns every idle one beyond keep_bytes to the OS; *released (optional)
 * = bytes freed.  Buffers a caller still owns are untouched.  There is no
 * reference counterpart: DuckDB owns its buffers (conn.execute(sql)). */
int giql_hip_host_pool_trim(int64_t keep_bytes, int64_t* released);

#ifdef __cplusplus
}
#endif
#endif /* GIQL_HIP_H */
