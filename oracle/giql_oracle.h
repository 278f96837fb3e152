/*
 * oracle/giql_oracle.h -- CPU restatement of the INTERSECTS / NEAREST hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under giql_amd/ (the product) may include,
 * link, import or execute anything in oracle/.  Allowed users: tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg -- and there only as
 * the checker / the timed CPU baseline, never as the thing shipped.
 *
 * Parity status: PINNED by the reference's own known-answer tests (restated as
 * data in the tests/golden JSON fixtures, checked by tests/test_oracle_golden.py) and by
 * sqlite3 executing the reference's emitted predicate / distance-CASE text
 * (tests/golden/make_golden.py).  The reference's real engine (DuckDB IE_JOIN,
 * duckdb>=1.4.0, pyproject.toml:38) is a third-party dependency that is absent
 * from /root/reference and from this image, so there is no oracle/_ref build.
 *
 * All citations are path:line under /root/reference/.
 */
#ifndef GIQL_ORACLE_H
#define GIQL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One join side: borrowed int32 column buffers (Arrow data buffers, no nulls).
 * start_off / end_off are the canonicalisation offsets of
 * src/giql/canonical.py:16-52 (start: 0 or -1; end: +1, 0 or -1). */
typedef struct ora_side {
  const int32_t* chrom;
  const int32_t* start;
  const int32_t* end;
  int64_t n;
  int32_t start_off;
  int32_t end_off;
} ora_side;

/* INNER join, bag semantics (src/giql/expanders/intersects.py:149-154,
 * intersects_duckdb.py:1235-1243, 1283-1299).  Literal nested loop. */
int ora_inner_brute(const ora_side* a, const ora_side* b, int64_t* n_pairs,
                    int32_t** row_a, int32_t** row_b);

/* Same result set, per-chromosome sort + prefix-max sweep, OpenMP over
 * chromosomes (the structure of intersects_duckdb.py:1317-1330 /
 * _per_chrom.py:46-69: partition by chrom, join each, UNION ALL).  Exact for
 * the literal predicate on any input (no start<end assumption). */
int ora_inner_sweep(const ora_side* a, const ora_side* b, int n_threads,
                    int64_t* n_pairs, int32_t** row_a, int32_t** row_b);

/* Per-A-row overlap count (intersects_duckdb.py:806-854 before GROUP BY;
 * tests/test_duckdb_iejoin.py:66-81). counts has a->n entries. */
int ora_count_brute(const ora_side* a, const ora_side* b, int64_t* counts);
int ora_count_sweep(const ora_side* a, const ora_side* b, int n_threads,
                    int64_t* counts);

/* SEMI (anti=0) / ANTI (anti=1): A row ids, ascending, one per qualifying left
 * row (intersects_duckdb.py:1254-1282, 1321-1324; tests :49-63). */
int ora_semi_anti(const ora_side* a, const ora_side* b, int anti, int n_threads,
                  int64_t* n_out, int32_t** rows_a);

/* NEAREST k=1 (src/giql/expanders/nearest.py:313-333, 387-396;
 * _distance.py:67-87).  Per A row: idx_b = chosen B row (-1 when the A row's
 * chromosome has no B row, or none within max_distance), dist = distance
 * (signed when is_signed).  max_distance < 0 means "no limit".
 * Ties: |distance|, then b.start, then b.end, then lowest row id. */
int ora_nearest_k1_brute(const ora_side* a, const ora_side* b, int is_signed,
                         int64_t max_distance, int32_t* idx_b, int64_t* dist);
/* NEAREST k >= 1 by brute force (nearest.py:336-397): [n_a * k] row-major outputs. */
int ora_nearest_k_brute(const ora_side* a, const ora_side* b, int32_t k, int is_signed,
                        int64_t max_distance, int threads, int32_t* idx_b, int64_t* dist);
int ora_nearest_k1_sweep(const ora_side* a, const ora_side* b, int is_signed,
                         int64_t max_distance, int n_threads, int32_t* idx_b,
                         int64_t* dist);

/* CLUSTER (src/giql/expanders/cluster.py:210-300): per partition (chrom ids here;
 * the caller folds strand into the id when stranded), rows ordered by RAW start
 * (no canonicalisation: the emitted window SQL reads the raw columns), stable;
 *   is_new = NOT (MAX(end) over the preceding rows + distance >= start)   (first row: new)
 *   cluster_id = SUM(is_new) OVER (PARTITION BY chrom ORDER BY start)      -- RANGE frame:
 *   rows with equal start are peers and share the sum.
 * ids[i] (1-based within the partition) is written for every input row. */
int ora_cluster(const ora_side* s, int64_t distance, int64_t* ids);

/* MERGE (src/giql/expanders/merge.py:186-330): GROUP BY chrom, cluster id ->
 * chrom, MIN(start), MAX(end), COUNT(*); ORDER BY chrom, start.  Outputs are
 * malloc'ed arrays of *n_out entries (ora_free). */
int ora_merge(const ora_side* s, int64_t distance, int64_t* n_out, int32_t** chrom,
              int32_t** start, int32_t** end, int64_t** count);

/* Order-independent 64-bit checksum of a pair multiset. */
uint64_t ora_pairs_checksum(const int32_t* row_a, const int32_t* row_b,
                            int64_t n);

void ora_free(void* p);
int ora_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
