/*
 * oracle/giql_oracle.c -- CPU restatement of the INTERSECTS / NEAREST hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see giql_oracle.h).  Plain C + OpenMP.
 * Citations are path:line under /root/reference/.
 *
 * Semantics restated:
 *   predicate   a.chrom = b.chrom AND a.start < b.end AND a.end > b.start on
 *               canonical 0-based half-open coordinates
 *               (src/giql/expanders/intersects.py:149-154;
 *                src/giql/expanders/intersects_duckdb.py:1235-1243)
 *   canonical   start' = start + start_off, end' = end + end_off
 *               (src/giql/canonical.py:16-52)
 *   INNER       bag semantics, per-chromosome partition then UNION ALL
 *               (intersects_duckdb.py:1283-1299, 1317-1330)
 *   SEMI/ANTI   one output row per qualifying left row; ANTI keeps rows on
 *               chromosomes absent from the right side
 *               (intersects_duckdb.py:1254-1282, 1321-1324)
 *   COUNT       overlapping right rows per left row
 *               (tests/test_duckdb_iejoin.py:66-81)
 *   NEAREST     distance CASE of src/giql/expanders/_distance.py:67-87, order
 *               ABS(distance), start, end, LIMIT 1
 *               (src/giql/expanders/nearest.py:387-396)
 */
#include "giql_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORA_OK 0
#define ORA_ENOMEM -1
#define ORA_EINVAL -2

static inline int64_t cs_of(const ora_side* s, int64_t i) {
  return (int64_t)s->start[i] + s->start_off;
}
static inline int64_t ce_of(const ora_side* s, int64_t i) {
  return (int64_t)s->end[i] + s->end_off;
}

void ora_free(void* p) { free(p); }

int ora_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---------------------------------------------------------------- vectors */
typedef struct {
  int32_t* a;
  int32_t* b;
  int64_t n, cap;
} pairvec;

static int pv_push(pairvec* v, int32_t ra, int32_t rb) {
  if (v->n == v->cap) {
    int64_t nc = v->cap ? v->cap * 2 : 1024;
    int32_t* na = (int32_t*)realloc(v->a, (size_t)nc * sizeof(int32_t));
    if (!na) return ORA_ENOMEM;
    v->a = na;
    int32_t* nb = (int32_t*)realloc(v->b, (size_t)nc * sizeof(int32_t));
    if (!nb) return ORA_ENOMEM;
    v->b = nb;
    v->cap = nc;
  }
  v->a[v->n] = ra;
  v->b[v->n] = rb;
  v->n++;
  return ORA_OK;
}

/* ------------------------------------------------------------ brute force */
int ora_inner_brute(const ora_side* a, const ora_side* b, int64_t* n_pairs,
                    int32_t** row_a, int32_t** row_b) {
  pairvec v = {0, 0, 0, 0};
  for (int64_t i = 0; i < a->n; i++) {
    const int64_t as = cs_of(a, i), ae = ce_of(a, i);
    const int32_t ac = a->chrom[i];
    for (int64_t j = 0; j < b->n; j++) {
      if (ac == b->chrom[j] && as < ce_of(b, j) && ae > cs_of(b, j)) {
        if (pv_push(&v, (int32_t)i, (int32_t)j)) {
          free(v.a);
          free(v.b);
          return ORA_ENOMEM;
        }
      }
    }
  }
  *n_pairs = v.n;
  *row_a = v.a;
  *row_b = v.b;
  return ORA_OK;
}

int ora_count_brute(const ora_side* a, const ora_side* b, int64_t* counts) {
  for (int64_t i = 0; i < a->n; i++) {
    const int64_t as = cs_of(a, i), ae = ce_of(a, i);
    const int32_t ac = a->chrom[i];
    int64_t c = 0;
    for (int64_t j = 0; j < b->n; j++)
      c += (ac == b->chrom[j] && as < ce_of(b, j) && ae > cs_of(b, j));
    counts[i] = c;
  }
  return ORA_OK;
}

/* distance CASE, src/giql/expanders/_distance.py:67-87 (unstranded). The
 * chrom test is done by the caller (WHERE ref.chrom = target.chrom,
 * nearest.py:327). */
static inline int64_t distance_case(int64_t as, int64_t ae, int64_t bs,
                                    int64_t be, int is_signed) {
  if (as < be && ae > bs) return 0;
  if (ae <= bs) return bs - ae + 1;
  return is_signed ? -(as - be + 1) : (as - be + 1);
}

static inline int64_t iabs64(int64_t x) { return x < 0 ? -x : x; }

int ora_nearest_k1_brute(const ora_side* a, const ora_side* b, int is_signed,
                         int64_t max_distance, int32_t* idx_b, int64_t* dist) {
  for (int64_t i = 0; i < a->n; i++) {
    const int64_t as = cs_of(a, i), ae = ce_of(a, i);
    const int32_t ac = a->chrom[i];
    int64_t best = -1, best_d = 0, best_s = 0, best_e = 0;
    for (int64_t j = 0; j < b->n; j++) {
      if (b->chrom[j] != ac) continue;
      const int64_t bs = cs_of(b, j), be = ce_of(b, j);
      const int64_t d = distance_case(as, ae, bs, be, is_signed);
      const int64_t ad = iabs64(d);
      if (max_distance >= 0 && ad > max_distance) continue;
      /* ORDER BY ABS(distance), start, end  (nearest.py:392-395) */
      if (best < 0 || ad < iabs64(best_d) ||
          (ad == iabs64(best_d) &&
           (bs < best_s || (bs == best_s && be < best_e)))) {
        best = j;
        best_d = d;
        best_s = bs;
        best_e = be;
      }
    }
    idx_b[i] = (int32_t)best;
    dist[i] = best < 0 ? 0 : best_d;
  }
  return ORA_OK;
}


/* NEAREST k >= 1, brute force: for every A row the k best B rows of its chromosome under
 * ORDER BY ABS(distance), start, end LIMIT k (src/giql/expanders/nearest.py:336-397,
 * _distance.py:67-87), kept by insertion into a k-slot list.  idx_b / dist are [n_a * k]
 * row-major; unused slots -1 / 0.  Rows tied on (|d|, start, end) keep the lower row id. */
int ora_nearest_k_brute(const ora_side* a, const ora_side* b, int32_t k, int is_signed,
                        int64_t max_distance, int threads, int32_t* idx_b, int64_t* dist) {
  if (k < 1) return ORA_EINVAL;
  (void)threads;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)
  for (int64_t i = 0; i < a->n; i++) {
    const int64_t as = cs_of(a, i), ae = ce_of(a, i);
    const int32_t ac = a->chrom[i];
    int32_t* bi = idx_b + i * k;
    int64_t* bd = dist + i * k;
    int n = 0;
    for (int64_t j = 0; j < b->n; j++) {
      if (b->chrom[j] != ac) continue;
      const int64_t bs = cs_of(b, j), be = ce_of(b, j);
      const int64_t d = distance_case(as, ae, bs, be, is_signed);
      const int64_t ad = iabs64(d);
      if (max_distance >= 0 && ad > max_distance) continue;
      /* position of (ad, bs, be) among the kept ones; equal keys go after (stable by row id) */
      int pos = n;
      while (pos > 0) {
        const int64_t j2 = bi[pos - 1];
        const int64_t ad2 = iabs64(bd[pos - 1]);
        const int64_t bs2 = cs_of(b, j2), be2 = ce_of(b, j2);
        if (ad < ad2 || (ad == ad2 && (bs < bs2 || (bs == bs2 && be < be2))))
          pos--;
        else
          break;
      }
      if (pos >= k) continue;
      const int last = n < k ? n : k - 1;
      for (int t = last; t > pos; t--) {
        bi[t] = bi[t - 1];
        bd[t] = bd[t - 1];
      }
      bi[pos] = (int32_t)j;
      bd[pos] = d;
      if (n < k) n++;
    }
    for (int t = n; t < k; t++) {
      bi[t] = -1;
      bd[t] = 0;
    }
  }
  return ORA_OK;
}

/* --------------------------------------------------- per-chromosome index */
typedef struct {
  int32_t n_chrom;  /* max chrom id + 1 over both sides */
  int64_t* a_off;   /* [n_chrom+1] CSR offsets into a_idx */
  int32_t* a_idx;   /* A row ids grouped by chrom (ascending inside a chrom) */
  int64_t* b_off;   /* [n_chrom+1] */
  int32_t* b_idx;   /* B row ids per chrom, sorted by (cstart[, cend]) */
  int64_t* b_cs;    /* canonical start, same order as b_idx */
  int64_t* b_ce;    /* canonical end */
  int64_t* b_pmax;  /* running max of b_ce inside the chrom */
} chrom_index;

static void ci_free(chrom_index* ci) {
  free(ci->a_off);
  free(ci->a_idx);
  free(ci->b_off);
  free(ci->b_idx);
  free(ci->b_cs);
  free(ci->b_ce);
  free(ci->b_pmax);
  memset(ci, 0, sizeof(*ci));
}

static int group_by_chrom(const ora_side* s, int32_t n_chrom, int64_t** off_out,
                          int32_t** idx_out) {
  int64_t* off = (int64_t*)calloc((size_t)n_chrom + 1, sizeof(int64_t));
  int32_t* idx = (int32_t*)malloc((size_t)(s->n > 0 ? s->n : 1) * sizeof(int32_t));
  if (!off || !idx) {
    free(off);
    free(idx);
    return ORA_ENOMEM;
  }
  for (int64_t i = 0; i < s->n; i++) off[s->chrom[i] + 1]++;
  for (int32_t c = 0; c < n_chrom; c++) off[c + 1] += off[c];
  int64_t* cur = (int64_t*)malloc((size_t)(n_chrom + 1) * sizeof(int64_t));
  if (!cur) {
    free(off);
    free(idx);
    return ORA_ENOMEM;
  }
  memcpy(cur, off, (size_t)(n_chrom + 1) * sizeof(int64_t));
  for (int64_t i = 0; i < s->n; i++) idx[cur[s->chrom[i]]++] = (int32_t)i;
  free(cur);
  *off_out = off;
  *idx_out = idx;
  return ORA_OK;
}

/* Stable LSD radix sort of idx[0..n) by the signed 32-bit key key_of[idx]. */
static int radix_sort_idx(int32_t* idx, int64_t n, const int32_t* col) {
  if (n < 2) return ORA_OK;
  int32_t* tmp = (int32_t*)malloc((size_t)n * sizeof(int32_t));
  if (!tmp) return ORA_ENOMEM;
  int32_t* src = idx;
  int32_t* dst = tmp;
  for (int pass = 0; pass < 4; pass++) {
    int64_t hist[257];
    memset(hist, 0, sizeof(hist));
    const int sh = pass * 8;
    for (int64_t i = 0; i < n; i++) {
      uint32_t k = (uint32_t)col[src[i]] ^ 0x80000000u;
      hist[((k >> sh) & 255u) + 1]++;
    }
    for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
    for (int64_t i = 0; i < n; i++) {
      uint32_t k = (uint32_t)col[src[i]] ^ 0x80000000u;
      dst[hist[(k >> sh) & 255u]++] = src[i];
    }
    int32_t* t = src;
    src = dst;
    dst = t;
  }
  /* 4 passes: result is back in idx */
  free(tmp);
  return ORA_OK;
}

static int ci_build(const ora_side* a, const ora_side* b, int sort_by_end_too,
                    int n_threads, chrom_index* ci) {
  memset(ci, 0, sizeof(*ci));
  int32_t mx = -1;
  for (int64_t i = 0; i < a->n; i++) {
    if (a->chrom[i] < 0) return ORA_EINVAL;
    if (a->chrom[i] > mx) mx = a->chrom[i];
  }
  for (int64_t i = 0; i < b->n; i++) {
    if (b->chrom[i] < 0) return ORA_EINVAL;
    if (b->chrom[i] > mx) mx = b->chrom[i];
  }
  ci->n_chrom = mx + 1;
  int rc = group_by_chrom(a, ci->n_chrom, &ci->a_off, &ci->a_idx);
  if (rc) return rc;
  rc = group_by_chrom(b, ci->n_chrom, &ci->b_off, &ci->b_idx);
  if (rc) {
    ci_free(ci);
    return rc;
  }
  const size_t nb = (size_t)(b->n > 0 ? b->n : 1);
  ci->b_cs = (int64_t*)malloc(nb * sizeof(int64_t));
  ci->b_ce = (int64_t*)malloc(nb * sizeof(int64_t));
  ci->b_pmax = (int64_t*)malloc(nb * sizeof(int64_t));
  if (!ci->b_cs || !ci->b_ce || !ci->b_pmax) {
    ci_free(ci);
    return ORA_ENOMEM;
  }
  int err = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int32_t c = 0; c < ci->n_chrom; c++) {
    const int64_t lo = ci->b_off[c], hi = ci->b_off[c + 1];
    int32_t* idx = ci->b_idx + lo;
    /* (start, end) lexicographic = stable sort by end, then by start */
    if (sort_by_end_too && radix_sort_idx(idx, hi - lo, b->end)) err = 1;
    if (radix_sort_idx(idx, hi - lo, b->start)) err = 1;
    int64_t run = INT64_MIN;
    for (int64_t k = lo; k < hi; k++) {
      const int32_t r = ci->b_idx[k];
      ci->b_cs[k] = cs_of(b, r);
      ci->b_ce[k] = ce_of(b, r);
      if (ci->b_ce[k] > run) run = ci->b_ce[k];
      ci->b_pmax[k] = run;
    }
  }
  if (err) {
    ci_free(ci);
    return ORA_ENOMEM;
  }
  return ORA_OK;
}

/* number of entries of v[lo..hi) that are < x */
static inline int64_t lower_bound64(const int64_t* v, int64_t lo, int64_t hi,
                                    int64_t x) {
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if (v[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

/* first index in [lo,hi) with v[idx] > x (v non-decreasing) */
static inline int64_t first_greater64(const int64_t* v, int64_t lo, int64_t hi,
                                      int64_t x) {
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if (v[mid] > x)
      hi = mid;
    else
      lo = mid + 1;
  }
  return lo;
}

/* ----------------------------------------------------------------- sweeps */
#define ORA_CHUNK 32768

typedef struct {
  int32_t chrom;
  int64_t lo, hi; /* range inside a_idx */
} task_t;

static int make_tasks(const chrom_index* ci, task_t** tasks_out,
                      int64_t* n_tasks_out) {
  int64_t nt = 0;
  for (int32_t c = 0; c < ci->n_chrom; c++) {
    int64_t n = ci->a_off[c + 1] - ci->a_off[c];
    nt += (n + ORA_CHUNK - 1) / ORA_CHUNK;
  }
  task_t* t = (task_t*)malloc((size_t)(nt > 0 ? nt : 1) * sizeof(task_t));
  if (!t) return ORA_ENOMEM;
  int64_t k = 0;
  for (int32_t c = 0; c < ci->n_chrom; c++) {
    for (int64_t lo = ci->a_off[c]; lo < ci->a_off[c + 1]; lo += ORA_CHUNK) {
      int64_t hi = lo + ORA_CHUNK;
      if (hi > ci->a_off[c + 1]) hi = ci->a_off[c + 1];
      t[k].chrom = c;
      t[k].lo = lo;
      t[k].hi = hi;
      k++;
    }
  }
  *tasks_out = t;
  *n_tasks_out = nt;
  return ORA_OK;
}

int ora_inner_sweep(const ora_side* a, const ora_side* b, int n_threads,
                    int64_t* n_pairs, int32_t** row_a, int32_t** row_b) {
  if (n_threads < 1) n_threads = 1;
  chrom_index ci;
  int rc = ci_build(a, b, 0, n_threads, &ci);
  if (rc) return rc;
  task_t* tasks;
  int64_t nt;
  rc = make_tasks(&ci, &tasks, &nt);
  if (rc) {
    ci_free(&ci);
    return rc;
  }
  pairvec* out = (pairvec*)calloc((size_t)(nt > 0 ? nt : 1), sizeof(pairvec));
  if (!out) {
    free(tasks);
    ci_free(&ci);
    return ORA_ENOMEM;
  }
  int err = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int64_t t = 0; t < nt; t++) {
    const int32_t c = tasks[t].chrom;
    const int64_t blo = ci.b_off[c], bhi = ci.b_off[c + 1];
    pairvec local = {0, 0, 0, 0}; /* thread-private: no false sharing on out[] */
    pairvec* v = &local;
    for (int64_t k = tasks[t].lo; k < tasks[t].hi; k++) {
      const int32_t ra = ci.a_idx[k];
      const int64_t as = cs_of(a, ra), ae = ce_of(a, ra);
      /* candidates: b.start < a.end  (prefix of the start-sorted run) */
      int64_t j = lower_bound64(ci.b_cs, blo, bhi, ae);
      /* walk down while some earlier row can still have b.end > a.start */
      for (j = j - 1; j >= blo && ci.b_pmax[j] > as; j--) {
        if (ci.b_ce[j] > as) {
          if (pv_push(v, ra, ci.b_idx[j])) err = 1;
        }
      }
    }
    out[t] = local;
  }
  int64_t total = 0;
  for (int64_t t = 0; t < nt; t++) total += out[t].n;
  int32_t* ra = (int32_t*)malloc((size_t)(total > 0 ? total : 1) * sizeof(int32_t));
  int32_t* rb = (int32_t*)malloc((size_t)(total > 0 ? total : 1) * sizeof(int32_t));
  if (!ra || !rb) err = 1;
  if (!err) {
    int64_t* base = (int64_t*)malloc((size_t)(nt + 1) * sizeof(int64_t));
    if (!base) {
      err = 1;
    } else {
      base[0] = 0;
      for (int64_t t = 0; t < nt; t++) base[t + 1] = base[t] + out[t].n;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
      for (int64_t t = 0; t < nt; t++) {
        if (out[t].n) {
          memcpy(ra + base[t], out[t].a, (size_t)out[t].n * sizeof(int32_t));
          memcpy(rb + base[t], out[t].b, (size_t)out[t].n * sizeof(int32_t));
        }
      }
      free(base);
    }
  }
  for (int64_t t = 0; t < nt; t++) {
    free(out[t].a);
    free(out[t].b);
  }
  free(out);
  free(tasks);
  ci_free(&ci);
  if (err) {
    free(ra);
    free(rb);
    return ORA_ENOMEM;
  }
  *n_pairs = total;
  *row_a = ra;
  *row_b = rb;
  return ORA_OK;
}

int ora_count_sweep(const ora_side* a, const ora_side* b, int n_threads,
                    int64_t* counts) {
  if (n_threads < 1) n_threads = 1;
  chrom_index ci;
  int rc = ci_build(a, b, 0, n_threads, &ci);
  if (rc) return rc;
  task_t* tasks;
  int64_t nt;
  rc = make_tasks(&ci, &tasks, &nt);
  if (rc) {
    ci_free(&ci);
    return rc;
  }
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int64_t t = 0; t < nt; t++) {
    const int32_t c = tasks[t].chrom;
    const int64_t blo = ci.b_off[c], bhi = ci.b_off[c + 1];
    for (int64_t k = tasks[t].lo; k < tasks[t].hi; k++) {
      const int32_t ra = ci.a_idx[k];
      const int64_t as = cs_of(a, ra), ae = ce_of(a, ra);
      int64_t j = lower_bound64(ci.b_cs, blo, bhi, ae);
      int64_t cnt = 0;
      for (j = j - 1; j >= blo && ci.b_pmax[j] > as; j--) cnt += (ci.b_ce[j] > as);
      counts[ra] = cnt;
    }
  }
  free(tasks);
  ci_free(&ci);
  return ORA_OK;
}

int ora_semi_anti(const ora_side* a, const ora_side* b, int anti, int n_threads,
                  int64_t* n_out, int32_t** rows_a) {
  if (n_threads < 1) n_threads = 1;
  chrom_index ci;
  int rc = ci_build(a, b, 0, n_threads, &ci);
  if (rc) return rc;
  uint8_t* hit = (uint8_t*)calloc((size_t)(a->n > 0 ? a->n : 1), 1);
  if (!hit) {
    ci_free(&ci);
    return ORA_ENOMEM;
  }
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int32_t c = 0; c < ci.n_chrom; c++) {
    const int64_t blo = ci.b_off[c], bhi = ci.b_off[c + 1];
    for (int64_t k = ci.a_off[c]; k < ci.a_off[c + 1]; k++) {
      const int32_t ra = ci.a_idx[k];
      const int64_t as = cs_of(a, ra), ae = ce_of(a, ra);
      /* EXISTS b: b.start < a.end AND b.end > a.start
       *   <=> max{b.end : b.start < a.end} > a.start */
      const int64_t j = lower_bound64(ci.b_cs, blo, bhi, ae);
      hit[ra] = (uint8_t)(j > blo && ci.b_pmax[j - 1] > as);
    }
  }
  int64_t n = 0;
  for (int64_t i = 0; i < a->n; i++) n += ((hit[i] != 0) != (anti != 0));
  int32_t* rows = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  if (!rows) {
    free(hit);
    ci_free(&ci);
    return ORA_ENOMEM;
  }
  int64_t k = 0;
  for (int64_t i = 0; i < a->n; i++)
    if ((hit[i] != 0) != (anti != 0)) rows[k++] = (int32_t)i;
  free(hit);
  ci_free(&ci);
  *n_out = n;
  *rows_a = rows;
  return ORA_OK;
}

/* Requires start' <= end' on both sides (well-formed rows); the brute form
 * above is the literal CASE for anything else. */
int ora_nearest_k1_sweep(const ora_side* a, const ora_side* b, int is_signed,
                         int64_t max_distance, int n_threads, int32_t* idx_b,
                         int64_t* dist) {
  if (n_threads < 1) n_threads = 1;
  chrom_index ci;
  int rc = ci_build(a, b, 1, n_threads, &ci);
  if (rc) return rc;
  task_t* tasks;
  int64_t nt;
  rc = make_tasks(&ci, &tasks, &nt);
  if (rc) {
    ci_free(&ci);
    return rc;
  }
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int64_t t = 0; t < nt; t++) {
    const int32_t c = tasks[t].chrom;
    const int64_t blo = ci.b_off[c], bhi = ci.b_off[c + 1];
    for (int64_t k = tasks[t].lo; k < tasks[t].hi; k++) {
      const int32_t ra = ci.a_idx[k];
      const int64_t as = cs_of(a, ra), ae = ce_of(a, ra);
      int64_t best = -1, best_d = 0;
      if (bhi > blo) {
        const int64_t hi = lower_bound64(ci.b_cs, blo, bhi, ae);
        if (hi > blo && ci.b_pmax[hi - 1] > as) {
          /* overlap: first (start,end)-ordered row whose end exceeds a.start */
          best = first_greater64(ci.b_pmax, blo, hi, as);
          best_d = 0;
        } else {
          int64_t up = -1, up_d = 0, dn = -1, dn_d = 0;
          if (hi > blo) {
            const int64_t m = ci.b_pmax[hi - 1];
            up = first_greater64(ci.b_pmax, blo, hi, m - 1);
            up_d = as - m + 1;
          }
          if (hi < bhi) {
            dn = hi;
            dn_d = ci.b_cs[hi] - ae + 1;
          }
          if (up >= 0 && (dn < 0 || up_d <= dn_d)) {
            best = up;
            best_d = is_signed ? -up_d : up_d;
          } else if (dn >= 0) {
            best = dn;
            best_d = dn_d;
          }
        }
        if (best >= 0 && max_distance >= 0 && iabs64(best_d) > max_distance)
          best = -1;
      }
      idx_b[ra] = best < 0 ? -1 : ci.b_idx[best];
      dist[ra] = best < 0 ? 0 : best_d;
    }
  }
  free(tasks);
  ci_free(&ci);
  return ORA_OK;
}

/* ------------------------------------------------------------- checksum */
uint64_t ora_pairs_checksum(const int32_t* row_a, const int32_t* row_b,
                            int64_t n) {
  uint64_t acc = 0;
#pragma omp parallel for reduction(+ : acc)
  for (int64_t i = 0; i < n; i++) {
    uint64_t x = ((uint64_t)(uint32_t)row_a[i] << 32) | (uint32_t)row_b[i];
    x *= 0x9E3779B97F4A7C15ull;
    x ^= (x >> 32);
    x *= 0xD6E8FEB86659FD93ull;
    acc += x;
  }
  return acc;
}


/* ------------------------------------------------------------------ CLUSTER / MERGE
 * Restates the window SQL of src/giql/expanders/cluster.py:210-300 and the GROUP BY of
 * src/giql/expanders/merge.py:186-330 on raw coordinates. */
typedef struct {
  int32_t chrom, start, end;
  int64_t row;
} ora_crow;

static int ora_crow_cmp(const void* x, const void* y) {
  const ora_crow* a = (const ora_crow*)x;
  const ora_crow* b = (const ora_crow*)y;
  if (a->chrom != b->chrom) return a->chrom < b->chrom ? -1 : 1;
  if (a->start != b->start) return a->start < b->start ? -1 : 1;
  return a->row < b->row ? -1 : (a->row > b->row ? 1 : 0); /* stable */
}

/* sorted rows + per-row is_new flag; returns the sorted array (caller frees) */
static ora_crow* ora_cluster_flags(const ora_side* s, int64_t distance, uint8_t** flags_out) {
  const int64_t n = s->n;
  ora_crow* r = (ora_crow*)malloc((size_t)(n > 0 ? n : 1) * sizeof(ora_crow));
  uint8_t* f = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
  if (!r || !f) {
    free(r);
    free(f);
    return NULL;
  }
  for (int64_t i = 0; i < n; i++) {
    r[i].chrom = s->chrom[i];
    r[i].start = s->start[i];
    r[i].end = s->end[i];
    r[i].row = i;
  }
  qsort(r, (size_t)n, sizeof(ora_crow), ora_crow_cmp);
  int64_t run_max = 0;
  for (int64_t i = 0; i < n; i++) {
    if (i == 0 || r[i].chrom != r[i - 1].chrom) {
      f[i] = 1; /* MAX over an empty frame is NULL -> CASE default 1 */
    } else {
      f[i] = (run_max + distance >= (int64_t)r[i].start) ? 0 : 1;
    }
    if (i == 0 || r[i].chrom != r[i - 1].chrom || (int64_t)r[i].end > run_max) run_max = r[i].end;
  }
  *flags_out = f;
  return r;
}

int ora_cluster(const ora_side* s, int64_t distance, int64_t* ids) {
  uint8_t* f = NULL;
  ora_crow* r = ora_cluster_flags(s, distance, &f);
  if (!r) return -1;
  const int64_t n = s->n;
  int64_t i = 0;
  int64_t sum = 0;
  while (i < n) {
    if (i == 0 || r[i].chrom != r[i - 1].chrom) sum = 0;
    /* peers: rows of this partition with the same start share SUM(...) (RANGE frame) */
    int64_t j = i;
    while (j < n && r[j].chrom == r[i].chrom && r[j].start == r[i].start) sum += f[j++];
    for (int64_t k = i; k < j; k++) ids[r[k].row] = sum;
    i = j;
  }
  free(r);
  free(f);
  return 0;
}

int ora_merge(const ora_side* s, int64_t distance, int64_t* n_out, int32_t** chrom,
              int32_t** start, int32_t** end, int64_t** count) {
  const int64_t n = s->n;
  int64_t* ids = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
  if (!ids || ora_cluster(s, distance, ids) != 0) {
    free(ids);
    return -1;
  }
  /* GROUP BY chrom, cluster id over the (chrom, start)-sorted rows: the ids are
   * non-decreasing along that order, so groups are runs */
  uint8_t* f = NULL;
  ora_crow* r = ora_cluster_flags(s, distance, &f);
  if (!r) {
    free(ids);
    return -1;
  }
  int64_t m = 0;
  for (int64_t i = 0; i < n; i++)
    if (i == 0 || r[i].chrom != r[i - 1].chrom || ids[r[i].row] != ids[r[i - 1].row]) m++;
  *chrom = (int32_t*)malloc((size_t)(m > 0 ? m : 1) * sizeof(int32_t));
  *start = (int32_t*)malloc((size_t)(m > 0 ? m : 1) * sizeof(int32_t));
  *end = (int32_t*)malloc((size_t)(m > 0 ? m : 1) * sizeof(int32_t));
  *count = (int64_t*)malloc((size_t)(m > 0 ? m : 1) * sizeof(int64_t));
  int64_t g = -1;
  for (int64_t i = 0; i < n; i++) {
    if (i == 0 || r[i].chrom != r[i - 1].chrom || ids[r[i].row] != ids[r[i - 1].row]) {
      g++;
      (*chrom)[g] = r[i].chrom;
      (*start)[g] = r[i].start;
      (*end)[g] = r[i].end;
      (*count)[g] = 0;
    }
    if (r[i].start < (*start)[g]) (*start)[g] = r[i].start;
    if (r[i].end > (*end)[g]) (*end)[g] = r[i].end;
    (*count)[g]++;
  }
  *n_out = m;
  free(r);
  free(f);
  free(ids);
  return 0;
}
