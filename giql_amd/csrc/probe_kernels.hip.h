// Streaming probes: what THIS device reads, writes and copies per second with a given access shape.
// The yardstick bench.py prints beside the 8 TB/s HBM3E peak ("verify on the box with a
// copy kernel"); MI355X_MICROARCH.md measures 6.29 TB/s for a float4 copy and 6.0-6.2 TB/s for streamed reads /
// stores, so a probe that reports less says something about the probe's shape (loads in flight, cache policy,
// grid), not about the box.  Not on the product path.
#pragma once
#include <hip/hip_runtime.h>

#include "dev_common.hip.h"

namespace giql {

typedef unsigned int probe_u4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ probe_u4 probe_load(const probe_u4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool NT>
__device__ __forceinline__ void probe_store(probe_u4* p, probe_u4 v) {
  if constexpr (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

// MODE 0: read only (the sum leaves through one store per block so that nothing is optimised away),
// 1: write only, 2: copy.  U = 16-byte accesses in flight per thread; tiles of 256 * U accesses, grid-stride.
template <int MODE, int U, bool NT>
__global__ __launch_bounds__(256) void k_stream_probe(const probe_u4* __restrict__ src, probe_u4* __restrict__ dst,
                                                      u64 n16, u32* __restrict__ sink) {
  const u64 tile = (u64)256 * U;
  const u64 n_tiles = n16 / tile;  // (the tail below one tile is not touched: the host passes whole tiles)
  probe_u4 acc = {0u, 0u, 0u, 0u};
  for (u64 t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const u64 base = t * tile + threadIdx.x;
    probe_u4 v[U];
    if constexpr (MODE != 1) {
#pragma unroll
      for (int k = 0; k < U; k++) v[k] = probe_load<NT>(src + base + (u64)k * 256);
    }
    if constexpr (MODE == 0) {
#pragma unroll
      for (int k = 0; k < U; k++) acc ^= v[k];
    } else if constexpr (MODE == 1) {
      const probe_u4 w = {(u32)t, (u32)threadIdx.x, 0u, 1u};
#pragma unroll
      for (int k = 0; k < U; k++) probe_store<NT>(dst + base + (u64)k * 256, w);
    } else {
#pragma unroll
      for (int k = 0; k < U; k++) probe_store<NT>(dst + base + (u64)k * 256, v[k]);
    }
  }
  if constexpr (MODE == 0) {
    const u32 x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x9E3779B9u) sink[blockIdx.x & 255] = x;  // (practically never: the loads stay live)
  }
}

// host side: pick the instantiation
template <int MODE, bool NT>
static void launch_stream_probe(int in_flight, u32 grid, hipStream_t st, const probe_u4* src, probe_u4* dst, u64 n16,
                                u32* sink) {
  switch (in_flight) {
    case 1: hipLaunchKernelGGL((k_stream_probe<MODE, 1, NT>), dim3(grid), dim3(256), 0, st, src, dst, n16, sink); break;
    case 2: hipLaunchKernelGGL((k_stream_probe<MODE, 2, NT>), dim3(grid), dim3(256), 0, st, src, dst, n16, sink); break;
    case 4: hipLaunchKernelGGL((k_stream_probe<MODE, 4, NT>), dim3(grid), dim3(256), 0, st, src, dst, n16, sink); break;
    default: hipLaunchKernelGGL((k_stream_probe<MODE, 8, NT>), dim3(grid), dim3(256), 0, st, src, dst, n16, sink); break;
  }
}

}  // namespace giql
