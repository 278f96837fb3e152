// Probe: a chain of 40 short kernels (+ 4 memsets, one fork/join) launched directly vs replayed as a hipGraph.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_touch(unsigned* p, unsigned n, unsigned v) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 3u + v;
}
static int chain(hipStream_t st, hipStream_t side, hipEvent_t f, hipEvent_t j, unsigned* a, unsigned* b, unsigned n) {
  for (int k = 0; k < 4; k++) CK(hipMemsetAsync(a, 0, 4096, st));
  CK(hipEventRecord(f, st));
  CK(hipStreamWaitEvent(side, f, 0));
  for (int k = 0; k < 12; k++) hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, side, b, n, (unsigned)k);
  for (int k = 0; k < 12; k++) hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, st, a, n, (unsigned)k);
  CK(hipEventRecord(j, side));
  CK(hipStreamWaitEvent(st, j, 0));
  for (int k = 0; k < 16; k++) hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, st, a, n, (unsigned)k);
  return 0;
}
int main() {
  const unsigned n = 1u << 20;
  unsigned *a, *b, *h;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipHostMalloc(&h, 64));
  hipStream_t cap, side; hipEvent_t f, j;
  CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  CK(hipEventCreateWithFlags(&f, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
  hipStream_t user = 0;  // the legacy default stream, as torch hands it over
  auto now = [] { return std::chrono::steady_clock::now(); };
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = now();
    for (int it = 0; it < 200; it++) {
      if (chain(user, side, f, j, a, b, n)) return 1;
      CK(hipMemcpyAsync(h, a, 64, hipMemcpyDeviceToHost, user));
      CK(hipStreamSynchronize(user));
    }
    double us = std::chrono::duration<double, std::micro>(now() - t0).count() / 200;
    printf("direct: %.1f us per chain\n", us);
  }
  auto t0 = now();
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
  if (chain(cap, side, f, j, a, b, n)) return 1;
  CK(hipStreamEndCapture(cap, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  printf("capture + instantiate: %.1f us\n", std::chrono::duration<double, std::micro>(now() - t0).count());
  for (int rep = 0; rep < 3; rep++) {
    auto t1 = now();
    for (int it = 0; it < 200; it++) {
      CK(hipGraphLaunch(ge, user));
      CK(hipMemcpyAsync(h, a, 64, hipMemcpyDeviceToHost, user));
      CK(hipStreamSynchronize(user));
    }
    double us = std::chrono::duration<double, std::micro>(now() - t1).count() / 200;
    printf("graph:  %.1f us per chain\n", us);
  }
  // same on a non-default stream
  for (int rep = 0; rep < 2; rep++) {
    auto t1 = now();
    for (int it = 0; it < 200; it++) {
      CK(hipGraphLaunch(ge, cap));
      CK(hipMemcpyAsync(h, a, 64, hipMemcpyDeviceToHost, cap));
      CK(hipStreamSynchronize(cap));
    }
    printf("graph on own stream: %.1f us per chain\n", std::chrono::duration<double, std::micro>(now() - t1).count() / 200);
  }
  for (int rep = 0; rep < 2; rep++) {
    auto t1 = now();
    for (int it = 0; it < 200; it++) {
      if (chain(cap, side, f, j, a, b, n)) return 1;
      CK(hipMemcpyAsync(h, a, 64, hipMemcpyDeviceToHost, cap));
      CK(hipStreamSynchronize(cap));
    }
    printf("direct on own stream: %.1f us per chain\n", std::chrono::duration<double, std::micro>(now() - t1).count() / 200);
  }
  return 0;
}
