// Read-bandwidth probe: three int32 arrays streamed like k_chrom_minmax (grid-stride, 4 rows in
// flight per thread) vs 16-byte loads.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_read(const int* __restrict__ a, const int* __restrict__ b,
                                              const int* __restrict__ c, size_t n, int* out) {
  int acc = 0;
  if (MODE == 0) {  // dword, 4 rows x 3 arrays in flight
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 4 * stride) {
      int v[12];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const size_t i = i0 + u * stride;
        const bool ok = i < n;
        v[3 * u] = ok ? a[i] : 0; v[3 * u + 1] = ok ? b[i] : 0; v[3 * u + 2] = ok ? c[i] : 0;
      }
#pragma unroll
      for (int u = 0; u < 12; u++) acc ^= v[u];
    }
  } else {  // dwordx4 per array per iteration
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const int4 x = ((const int4*)a)[i], y = ((const int4*)b)[i], z = ((const int4*)c)[i];
      acc ^= x.x ^ x.y ^ x.z ^ x.w ^ y.x ^ y.y ^ y.z ^ y.w ^ z.x ^ z.y ^ z.z ^ z.w;
    }
  }
  if (acc == 0x7fffffff) *out = acc;
}

template <int MODE>
void run(const char* name, int* a, int* b, int* c, size_t n, int* out, unsigned grid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_read<MODE>, dim3(grid), dim3(256), 0, 0, a, b, c, n, out);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_read<MODE>, dim3(grid), dim3(256), 0, 0, a, b, c, n, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  printf("%-22s grid %5u  %.3f ms  %.0f GB/s\n", name, grid, ms, 3.0 * n * 4 / ms / 1e6);
}

int main() {
  const size_t n = 100000000;
  int *a, *b, *c, *out; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&out, 4));
  CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 2, n * 4)); CK(hipMemset(c, 3, n * 4));
  for (unsigned grid : {1024u, 2048u, 4096u, 8192u}) {
    run<0>("dword x12 in flight", a, b, c, n, out, grid);
    run<1>("dwordx4 x3 in flight", a, b, c, n, out, grid);
  }
  return 0;
}
