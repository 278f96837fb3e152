// take_kernels.hip.h -- projection materialisation: Arrow `take` of payload columns
// by the row ids the join returns.
//
// In the reference this is the outer SELECT of the per-chromosome plan: the join
// relation is rebuilt with the projected columns of both sides
// (src/giql/expanders/intersects_duckdb.py:1402-1644).  Here the join returns
// (row_a, row_b) index pairs and these kernels gather the projected columns on the
// device, so a query that wants `a.name, b.score` never ships the 8-byte-per-pair
// index arrays to the host.
//
// HBM-bound: per output row one 4-byte index read (shared by all fixed-width
// columns of a call), one random elem-byte gather and one coalesced elem-byte write
// per column.  Indices < 0 (NEAREST's "no neighbour") produce zero bytes / empty
// strings; validity is the caller's business.
#pragma once
#include "dev_common.hip.h"

namespace giql {

constexpr int TK_NT = 256;
constexpr int TK_MAX_COLS = 8;

struct TakeArgs {
  const void* col[TK_MAX_COLS];
  void* out[TK_MAX_COLS];
  int elem[TK_MAX_COLS];  // bytes per value: 1, 2, 4, 8 or 16
  int n_cols;
  int vec;  // idx and every out pointer are 16-byte aligned: 4-row vector loads/stores
};

template <typename T>
__device__ __forceinline__ T take_one(const void* col, int ix) {
  return ix >= 0 ? reinterpret_cast<const T*>(col)[ix] : T{};
}

// 4 consecutive output rows per lane; the gathers of one column are issued
// back-to-back so four are in flight before the first store.
template <typename T, typename V4>
__device__ __forceinline__ void take_col4(const void* col, void* out, u64 i0, const int (&ix)[4]) {
  T v0 = take_one<T>(col, ix[0]);
  T v1 = take_one<T>(col, ix[1]);
  T v2 = take_one<T>(col, ix[2]);
  T v3 = take_one<T>(col, ix[3]);
  union {
    T t[4];
    V4 v;
  } pk;
  pk.t[0] = v0;
  pk.t[1] = v1;
  pk.t[2] = v2;
  pk.t[3] = v3;
  *reinterpret_cast<V4*>(reinterpret_cast<T*>(out) + i0) = pk.v;
}

struct alignas(16) TkU128 {
  u32 x, y, z, w;
};
struct alignas(16) TkU256 {
  TkU128 a, b;
};
struct alignas(16) TkU512 {
  TkU128 a, b, c, d;
};

__device__ __forceinline__ void take_scalar(const TakeArgs& a, int c, u64 i, int ix) {
  switch (a.elem[c]) {
    case 1: reinterpret_cast<uint8_t*>(a.out[c])[i] = take_one<uint8_t>(a.col[c], ix); break;
    case 2: reinterpret_cast<uint16_t*>(a.out[c])[i] = take_one<uint16_t>(a.col[c], ix); break;
    case 4: reinterpret_cast<u32*>(a.out[c])[i] = take_one<u32>(a.col[c], ix); break;
    case 8: reinterpret_cast<u64*>(a.out[c])[i] = take_one<u64>(a.col[c], ix); break;
    default: reinterpret_cast<TkU128*>(a.out[c])[i] = take_one<TkU128>(a.col[c], ix); break;
  }
}

__global__ __launch_bounds__(TK_NT) void k_take(TakeArgs a, const int* __restrict__ idx, u64 n,
                                                u32 n_rows, DevMeta* meta) {
  const u64 stride = (u64)gridDim.x * TK_NT * 4;
  bool bad = false;
  for (u64 i0 = ((u64)blockIdx.x * TK_NT + threadIdx.x) * 4; i0 < n; i0 += stride) {
    int ix[4];
    const int cnt = (n - i0) >= 4 ? 4 : (int)(n - i0);
    if (cnt == 4 && a.vec) {
      const int4 v = *reinterpret_cast<const int4*>(idx + i0);
      ix[0] = v.x;
      ix[1] = v.y;
      ix[2] = v.z;
      ix[3] = v.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) ix[k] = k < cnt ? idx[i0 + k] : -1;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (ix[k] >= 0 && (u32)ix[k] >= n_rows) {
        bad = true;
        ix[k] = -1;
      }
    }
    if (cnt == 4 && a.vec) {
      for (int c = 0; c < a.n_cols; c++) {
        switch (a.elem[c]) {
          case 1: take_col4<uint8_t, u32>(a.col[c], a.out[c], i0, ix); break;
          case 2: take_col4<uint16_t, uint2>(a.col[c], a.out[c], i0, ix); break;
          case 4: take_col4<u32, TkU128>(a.col[c], a.out[c], i0, ix); break;
          case 8: take_col4<u64, TkU256>(a.col[c], a.out[c], i0, ix); break;
          default: take_col4<TkU128, TkU512>(a.col[c], a.out[c], i0, ix); break;
        }
      }
    } else {
      for (int c = 0; c < a.n_cols; c++)
        for (int k = 0; k < cnt; k++) take_scalar(a, c, i0 + k, ix[k]);
    }
  }
  if (bad) atomicMin(&meta->status, -1 /* GIQL_ERR_INVALID */);
}

// ---- utf8 / binary columns (Arrow int32 offsets + data bytes) ---------------
// pass 1: byte length of every taken row (scanned in place into the output offsets)
__global__ __launch_bounds__(TK_NT) void k_take_utf8_len(const int* __restrict__ offsets,
                                                         u32 n_rows,
                                                         const int* __restrict__ idx, u64 n,
                                                         u32* __restrict__ len_out,
                                                         DevMeta* meta) {
  const u64 stride = (u64)gridDim.x * TK_NT;
  bool bad = false;
  for (u64 i = (u64)blockIdx.x * TK_NT + threadIdx.x; i < n; i += stride) {
    const int ix = idx[i];
    u32 len = 0;
    if (ix >= 0) {
      if ((u32)ix >= n_rows) {
        bad = true;
      } else {
        const int lo = offsets[ix], hi = offsets[ix + 1];
        if (hi < lo) bad = true;
        else len = (u32)(hi - lo);
      }
    }
    len_out[i] = len;
  }
  if (bad) atomicMin(&meta->status, -1);
}

// pass 2: copy the bytes.  One lane per output row for short values (adjacent lanes
// write adjacent output segments); rows longer than TKS_SHORT bytes are copied by the
// whole wave, 64 bytes per step.
constexpr u32 TKS_SHORT = 32;

__device__ __forceinline__ u64 shfl_u64(u64 v, int src_lane) {
  const u32 lo = (u32)__shfl((int)(u32)v, src_lane, 64);
  const u32 hi = (u32)__shfl((int)(u32)(v >> 32), src_lane, 64);
  return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(TK_NT) void k_take_utf8_copy(const int* __restrict__ offsets,
                                                          const uint8_t* __restrict__ data,
                                                          u32 n_rows,
                                                          const int* __restrict__ idx, u64 n,
                                                          const int* __restrict__ out_offsets,
                                                          uint8_t* __restrict__ out) {
  const u64 stride = (u64)gridDim.x * TK_NT;
  const u64 n_pad = (n + 63) & ~(u64)63;  // whole waves stay in the loop together
  for (u64 i = (u64)blockIdx.x * TK_NT + threadIdx.x; i < n_pad; i += stride) {
    u32 len = 0;
    const uint8_t* src = data;
    uint8_t* dst = out;
    if (i < n) {
      const int ix = idx[i];
      if (ix >= 0 && (u32)ix < n_rows) {
        const int lo = offsets[ix];
        const int o0 = out_offsets[i];
        len = (u32)(out_offsets[i + 1] - o0);
        src = data + lo;
        dst = out + o0;
      }
    }
    const bool is_long = len > TKS_SHORT;
    const u32 slen = is_long ? 0u : len;
    // short rows: 16 / 8 / 4-byte pieces at the row's own (arbitrary) alignment -- gfx950
    // serves unaligned global accesses, and hipcc lowers these fixed-size memcpys to single
    // dwordx4 / dwordx2 / dword instructions -- then at most 3 single bytes: a 13-byte value
    // is 3 load/store pairs instead of 13
    {
      u32 k = 0;
      for (; k + 16 <= slen; k += 16) {
        TkU128 w;
        __builtin_memcpy(&w, src + k, 16);
        __builtin_memcpy(dst + k, &w, 16);
      }
      if (k + 8 <= slen) {
        u64 w;
        __builtin_memcpy(&w, src + k, 8);
        __builtin_memcpy(dst + k, &w, 8);
        k += 8;
      }
      if (k + 4 <= slen) {
        u32 w;
        __builtin_memcpy(&w, src + k, 4);
        __builtin_memcpy(dst + k, &w, 4);
        k += 4;
      }
      for (; k < slen; k++) dst[k] = src[k];
    }
    u64 longs = __ballot(is_long);
    while (longs) {
      const int l = __ffsll((long long)longs) - 1;
      longs &= longs - 1;
      const u64 s = shfl_u64((u64)(uintptr_t)src, l);
      const u64 d = shfl_u64((u64)(uintptr_t)dst, l);
      const u32 ln = (u32)__shfl((int)len, l, 64);
      const uint8_t* ws = reinterpret_cast<const uint8_t*>((uintptr_t)s);
      uint8_t* wd = reinterpret_cast<uint8_t*>((uintptr_t)d);
      // 4 bytes per lane per step (unaligned dwords), then the last ln % 4 bytes
      const u32 body = ln & ~3u;
      for (u32 k = lane_id() * 4u; k < body; k += 256u) {
        u32 w;
        __builtin_memcpy(&w, ws + k, 4);
        __builtin_memcpy(wd + k, &w, 4);
      }
      if (lane_id() < (ln & 3u)) wd[body + lane_id()] = ws[body + lane_id()];
    }
  }
}

}  // namespace giql
