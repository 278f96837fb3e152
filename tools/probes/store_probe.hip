// Store-width probe: how fast can a wave stream 2 output arrays (like k_fill) with
// dword vs dwordx4 stores, temporal vs nontemporal?  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_store(int* __restrict__ a, int* __restrict__ b, size_t n) {
  // tile of 4096 outputs per block, 16 per lane
  const size_t base = (size_t)blockIdx.x * 4096;
  if (base + 4096 > n) return;
  const int tid = threadIdx.x;
  if (MODE == 0 || MODE == 2) {        // dword per lane, wave covers 64 consecutive
    const int w = tid >> 6, l = tid & 63;
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const size_t p = base + w * 1024 + it * 64 + l;
      if (MODE == 0) { a[p] = (int)p; b[p] = (int)(p ^ 5); }
      else { __builtin_nontemporal_store((int)p, a + p); __builtin_nontemporal_store((int)(p ^ 5), b + p); }
    }
  } else {                              // dwordx4 per lane
    const int w = tid >> 6, l = tid & 63;
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const size_t p = base + w * 1024 + it * 256 + l * 4;
      int4 v = make_int4((int)p, (int)p + 1, (int)p + 2, (int)p + 3);
      int4 u = make_int4((int)p ^ 5, (int)p, 7, 9);
      if (MODE == 1) { *(int4*)(a + p) = v; *(int4*)(b + p) = u; }
      else { __builtin_nontemporal_store(v.x, a + p); __builtin_nontemporal_store(v.y, a + p + 1);
             __builtin_nontemporal_store(v.z, a + p + 2); __builtin_nontemporal_store(v.w, a + p + 3);
             __builtin_nontemporal_store(u.x, b + p); __builtin_nontemporal_store(u.y, b + p + 1);
             __builtin_nontemporal_store(u.z, b + p + 2); __builtin_nontemporal_store(u.w, b + p + 3); }
    }
  }
}

template <int MODE>
void run(const char* name, int* a, int* b, size_t n) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)(n / 4096);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_store<MODE>, dim3(grid), dim3(256), 0, 0, a, b, n);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_store<MODE>, dim3(grid), dim3(256), 0, 0, a, b, n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  printf("%-28s %.3f ms  %.0f GB/s\n", name, ms, 2.0 * n * 4 / ms / 1e6);
}

int main() {
  const size_t n = (size_t)400 << 20;  // 400M outputs x 2 arrays = 3.2 GB
  int *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
  run<0>("dword", a, b, n);
  run<1>("dwordx4", a, b, n);
  run<2>("dword nontemporal", a, b, n);
  run<3>("dwordx4 split nontemporal", a, b, n);
  return 0;
}
