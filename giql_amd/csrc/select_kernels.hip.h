// select_kernels.hip.h -- residual predicates: the extra ON / WHERE conjuncts that
// sit beside the INTERSECTS.
//
// The reference inlines them into the per-chromosome join's ON clause
// (src/giql/expanders/intersects_duckdb.py:1164-1177, 1239-1243: "a.start < b.end
// AND a.end > b.start AND <residual> ...").  Here a residual is a comparison between
// two operands, each a payload column of one side (addressed through the pair's
// row id) or a literal; a conjunction of them is evaluated per pair -- or per input
// row when a predicate names one side only -- and the survivors are compacted
// stably: evaluate + ballot mask + per-block count, scan, scatter.
//
// HBM-bound: the pairs are read twice (8 B each), every referenced column is
// gathered once per pair, the kept pairs are written once.
#pragma once
#include "dev_common.hip.h"

namespace giql {

constexpr int SEL_NT = 256;
constexpr int SEL_ITEMS = 8;
constexpr int SEL_TILE = SEL_NT * SEL_ITEMS;
constexpr int SEL_MAX_PREDS = 16;

// mirrors giql_operand / giql_pred of include/giql_hip.h
struct DevOperand {
  int side;  // 0 = A (row_a), 1 = B (row_b), 2 = literal
  int type;  // 0 i32, 1 i64, 2 f32, 3 f64, 4 u8
  const void* data;
  const uint8_t* valid;
  i64 lit_i;
  double lit_f;
  int lit_is_float;
  int pad;
};
struct DevPred {
  DevOperand lhs, rhs;
  int op;     // 0 ==, 1 !=, 2 <, 3 <=, 4 >, 5 >=, 6 IS NULL, 7 IS NOT NULL, 8 IS TRUE (lhs only)
  int group;  // != 0 and equal to the predecessor's: OR-ed with it; otherwise a new AND-ed clause
};
struct DevPreds {
  DevPred p[SEL_MAX_PREDS];
  int n;
};

struct SelValue {
  i64 i;
  double f;
  bool is_float;
  bool null;
};

__device__ __forceinline__ SelValue sel_load(const DevOperand& o, int ia, int ib) {
  SelValue v;
  v.null = false;
  if (o.side == 2) {
    v.is_float = o.lit_is_float != 0;
    v.i = o.lit_i;
    v.f = o.lit_f;
    return v;
  }
  const int r = o.side == 0 ? ia : ib;
  if (o.valid && !o.valid[r]) v.null = true;
  v.is_float = o.type == 2 || o.type == 3;
  v.i = 0;
  v.f = 0.0;
  switch (o.type) {
    case 0: v.i = reinterpret_cast<const int*>(o.data)[r]; break;
    case 1: v.i = reinterpret_cast<const i64*>(o.data)[r]; break;
    case 2: v.f = (double)reinterpret_cast<const float*>(o.data)[r]; break;
    case 3: v.f = reinterpret_cast<const double*>(o.data)[r]; break;
    default: v.i = reinterpret_cast<const uint8_t*>(o.data)[r]; break;
  }
  return v;
}

template <typename T>
__device__ __forceinline__ bool sel_cmp(T a, T b, int op) {
  switch (op) {
    case 0: return a == b;
    case 1: return a != b;
    case 2: return a < b;
    case 3: return a <= b;
    case 4: return a > b;
    default: return a >= b;
  }
}

// ---- arithmetic operands (round 3) ---------------------------------------------------------
// An operand of side SEL_SIDE_EXPR is a postfix program over the node array `prog` (device
// memory; DevOperand-shaped nodes): nodes of side 0 / 1 / 2 push a column value / a literal,
// nodes of side >= SEL_X_ADD pop their arguments and push the result.  lit_i = first node,
// type = node count.  The reference inlines such text into its join's ON clause verbatim
// (intersects_duckdb.py:889-912, 1239-1243); the semantics are DuckDB's, its execution target:
// integer + - * stay 64-bit integers, `/` is a floating division (NULL on a zero divisor), NULL
// propagates through arithmetic, LEAST / GREATEST skip NULL arguments.
constexpr int SEL_SIDE_EXPR = 3;
constexpr int SEL_X_ADD = 16, SEL_X_SUB = 17, SEL_X_MUL = 18, SEL_X_DIV = 19, SEL_X_NEG = 20, SEL_X_ABS = 21,
              SEL_X_LEAST = 22, SEL_X_GREATEST = 23;
// Boolean nodes (round 4): a whole condition as ONE program -- comparisons, IS [NOT] NULL, AND / OR / NOT over
// three-valued results (a boolean is {i = 0 / 1, null}: Kleene logic, as SQL's) -- so that a condition needs no normal
// form: the reference inlines any boolean combination as text (intersects_duckdb.py:889-957), and a conjunctive
// normal form of an OR of ANDs outgrows every cap.  The predicate that holds such a program has op SEL_OP_IS_TRUE.
constexpr int SEL_X_EQ = 24, SEL_X_NE = 25, SEL_X_LT = 26, SEL_X_LE = 27, SEL_X_GT = 28, SEL_X_GE = 29,
              SEL_X_ISNULL = 30, SEL_X_NOTNULL = 31, SEL_X_AND = 32, SEL_X_OR = 33, SEL_X_NOT = 34;
constexpr int SEL_OP_IS_TRUE = 8;
constexpr int SEL_X_STACK = 12;
constexpr int SEL_MAX_NODES = 256;

__device__ __forceinline__ double sel_as_f(const SelValue& v) { return v.is_float ? v.f : (double)v.i; }

__device__ __noinline__ SelValue sel_eval_prog(const DevOperand* __restrict__ prog, int first, int count, int ia,
                                               int ib) {
  SelValue st[SEL_X_STACK];
  int sp = 0;
  for (int t = first; t < first + count; t++) {
    const DevOperand nd = prog[t];
    if (nd.side <= 2) {
      st[sp++] = sel_load(nd, ia, ib);
      continue;
    }
    if (nd.side == SEL_X_NEG || nd.side == SEL_X_ABS) {
      SelValue& v = st[sp - 1];
      if (nd.side == SEL_X_NEG) {
        v.i = -v.i;
        v.f = -v.f;
      } else {
        v.i = v.i < 0 ? -v.i : v.i;
        v.f = fabs(v.f);
      }
      continue;
    }
    if (nd.side == SEL_X_ISNULL || nd.side == SEL_X_NOTNULL || nd.side == SEL_X_NOT) {
      SelValue& v = st[sp - 1];
      if (nd.side == SEL_X_NOT) {
        v.i = (v.is_float ? v.f != 0.0 : v.i != 0) ? 0 : 1;   // NULL stays NULL
      } else {
        v.i = (nd.side == SEL_X_ISNULL) == v.null ? 1 : 0;
        v.null = false;
      }
      v.is_float = false;
      continue;
    }
    const SelValue b = st[--sp];
    SelValue& a = st[sp - 1];
    const bool fl = a.is_float || b.is_float;
    if (nd.side >= SEL_X_EQ && nd.side <= SEL_X_GE) {
      const int op = nd.side - SEL_X_EQ;
      const bool t = fl ? sel_cmp<double>(sel_as_f(a), sel_as_f(b), op) : sel_cmp<i64>(a.i, b.i, op);
      a.i = t ? 1 : 0;
      a.null = a.null || b.null;
      a.is_float = false;
      continue;
    }
    if (nd.side == SEL_X_AND || nd.side == SEL_X_OR) {
      // Kleene: AND is FALSE as soon as one side is FALSE, NULL when a NULL is left, else TRUE; OR the dual
      const bool ta = a.is_float ? a.f != 0.0 : a.i != 0, tb = b.is_float ? b.f != 0.0 : b.i != 0;
      const bool dom = nd.side == SEL_X_OR;   // the value that decides on its own
      const bool decided = (!a.null && ta == dom) || (!b.null && tb == dom);
      a.null = !decided && (a.null || b.null);
      a.i = decided ? (dom ? 1 : 0) : (dom ? 0 : 1);
      a.is_float = false;
      continue;
    }
    if (nd.side == SEL_X_LEAST || nd.side == SEL_X_GREATEST) {
      if (a.null) {
        a = b;
      } else if (!b.null) {
        const bool least = nd.side == SEL_X_LEAST;
        if (fl) {
          const double x = sel_as_f(a), y = sel_as_f(b);
          a.f = least ? (y < x ? y : x) : (y > x ? y : x);
          a.is_float = true;
        } else {
          a.i = least ? (b.i < a.i ? b.i : a.i) : (b.i > a.i ? b.i : a.i);
        }
      }
      continue;
    }
    a.null = a.null || b.null;
    if (nd.side == SEL_X_DIV) {
      const double y = sel_as_f(b);
      a.f = sel_as_f(a) / (y == 0.0 ? 1.0 : y);
      a.null = a.null || y == 0.0;
      a.is_float = true;
    } else if (fl) {
      const double x = sel_as_f(a), y = sel_as_f(b);
      a.f = nd.side == SEL_X_ADD ? x + y : (nd.side == SEL_X_SUB ? x - y : x * y);
      a.is_float = true;
    } else {
      a.i = nd.side == SEL_X_ADD ? a.i + b.i : (nd.side == SEL_X_SUB ? a.i - b.i : a.i * b.i);
    }
  }
  return st[0];
}

template <bool EXPR>
__device__ __forceinline__ SelValue sel_operand(const DevOperand& o, const DevOperand* __restrict__ prog, int ia,
                                                int ib) {
  if (EXPR && o.side == SEL_SIDE_EXPR) return sel_eval_prog(prog, (int)o.lit_i, o.type, ia, ib);
  return sel_load(o, ia, ib);
}

// SQL three-valued logic collapsed for a filter: a comparison with a NULL operand is
// not true.  The predicates form a conjunction of clauses; a clause is a run of
// predicates sharing one non-zero group id, OR-ed (a NOT has been pushed into the
// comparisons by the caller, so "not true" and "false" need no telling apart:
// AND / OR are monotone).
template <bool EXPR = false>
__device__ __forceinline__ bool sel_eval(const DevPreds& ps, int ia, int ib, const DevOperand* __restrict__ prog = nullptr) {
  bool keep = true, acc = true;
  int prev = 0;
  for (int k = 0; k < ps.n; k++) {
    const int op = ps.p[k].op;
    const SelValue a = sel_operand<EXPR>(ps.p[k].lhs, prog, ia, ib);
    bool t;
    if (op == SEL_OP_IS_TRUE) {
      t = !a.null && (a.is_float ? a.f != 0.0 : a.i != 0);
    } else if (op >= 6) {
      t = (op == 6) == a.null;
    } else {
      const SelValue b = sel_operand<EXPR>(ps.p[k].rhs, prog, ia, ib);
      if (a.is_float || b.is_float)
        t = sel_cmp<double>(a.is_float ? a.f : (double)a.i, b.is_float ? b.f : (double)b.i, op);
      else
        t = sel_cmp<i64>(a.i, b.i, op);
      t = t && !a.null && !b.null;
    }
    const int g = ps.p[k].group;
    if (g != 0 && g == prev) {
      acc = acc || t;
    } else {
      keep = keep && acc;
      acc = t;
    }
    prev = g;
  }
  return keep && acc;
}

// pass 1: ballot masks (one 64-bit word per 64 consecutive candidates) + block counts.
// A row id outside its side's row count raises GIQL_ERR_INVALID and drops the pair.
template <bool EXPR>
__global__ __launch_bounds__(SEL_NT) void k_select_count(DevPreds ps, const DevOperand* __restrict__ prog,
                                                         const int* __restrict__ idx_a,
                                                         u32 n_rows_a,
                                                         const int* __restrict__ idx_b,
                                                         u32 n_rows_b, u64 n,
                                                         u64* __restrict__ mask,
                                                         u32* __restrict__ cnt, DevMeta* meta) {
  __shared__ u32 s_cnt[SEL_NT / WAVE];
  const u64 base = (u64)blockIdx.x * SEL_TILE;
  u32 kept = 0;
  bool bad = false;
#pragma unroll 2
  for (int j = 0; j < SEL_ITEMS; j++) {
    const u64 i = base + (u64)j * SEL_NT + threadIdx.x;
    bool keep = false;
    if (i < n) {
      const int ia = idx_a ? idx_a[i] : (int)i;
      const int ib = idx_b ? idx_b[i] : (int)i;
      if (ia < 0 || (u32)ia >= n_rows_a || ib < 0 || (u32)ib >= n_rows_b)
        bad = true;
      else
        keep = sel_eval<EXPR>(ps, ia, ib, prog);
    }
    const u64 m = __ballot(keep);
    if (lane_id() == 0 && (base + (u64)j * SEL_NT + wave_id() * WAVE) < n) mask[i >> 6] = m;
    kept += (u32)__popcll(m);
  }
  if (lane_id() == 0) s_cnt[wave_id()] = kept;  // every lane of a wave holds the wave total
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 t = 0;
    for (int w = 0; w < SEL_NT / WAVE; w++) t += s_cnt[w];
    cnt[blockIdx.x] = t;
  }
  if (bad) atomicMin(&meta->status, -1);
}

// pass 2: stable scatter of the kept candidates.  Candidate i of the block sits in
// mask word (i - base) / 64; word w of the block covers item j = w / 4, wave w % 4.
__global__ __launch_bounds__(SEL_NT) void k_select_scatter(const int* __restrict__ idx_a,
                                                           const int* __restrict__ idx_b, u64 n,
                                                           const u64* __restrict__ mask,
                                                           const u32* __restrict__ off,
                                                           int* __restrict__ out_a,
                                                           int* __restrict__ out_b) {
  constexpr int WORDS = SEL_TILE / WAVE;  // 32
  __shared__ u32 s_pre[WORDS];
  const u64 base = (u64)blockIdx.x * SEL_TILE;
  const u64 w0 = base >> 6;
  const u64 n_words = (n + 63) >> 6;
  if (threadIdx.x < WAVE) {
    const u32 l = threadIdx.x;
    u32 c = (l < WORDS && w0 + l < n_words) ? (u32)__popcll(mask[w0 + l]) : 0u;
    const u32 incl = wave_incl_scan_add_u32(c);
    if (l < WORDS) s_pre[l] = incl - c;
  }
  __syncthreads();
  const u32 o0 = off[blockIdx.x];
#pragma unroll 2
  for (int j = 0; j < SEL_ITEMS; j++) {
    const u64 i = base + (u64)j * SEL_NT + threadIdx.x;
    if (i >= n) continue;
    const u32 w = (u32)((i - base) >> 6);
    const u64 m = mask[w0 + w];
    if ((m >> lane_id()) & 1ull) {
      const u32 pos = o0 + s_pre[w] + (u32)__popcll(m & lanemask_lt());
      if (out_a) out_a[pos] = idx_a ? idx_a[i] : (int)i;
      if (out_b) out_b[pos] = idx_b ? idx_b[i] : (int)i;
    }
  }
}

// flags[idx[i]] = 1 (SEMI / ANTI with two-sided residuals: the left rows that keep
// at least one pair)
__global__ __launch_bounds__(256) void k_mark(const int* __restrict__ idx, u64 n, u32 n_rows,
                                              uint8_t* __restrict__ flags, DevMeta* meta) {
  const u64 stride = (u64)gridDim.x * 256;
  bool bad = false;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int r = idx[i];
    if (r < 0 || (u32)r >= n_rows) bad = true;
    else flags[r] = 1;
  }
  if (bad) atomicMin(&meta->status, -1);
}

}  // namespace giql
