// Host-side expansion of a compact join plan (VERDICT r03 "Next round" 8: would a compact-plan download + expansion by
// host threads beat moving the 3.2 GB of pairs over PCIe?).  Synthetic plan of the headline's shape: 10M query rows,
// ~40 pairs each, a 100M-entry sorted-id array; the pairs are written into arrays that have been touched before (as the
// library's pinned pool hands them out).  Prints GB/s of pairs written for several thread counts.
//   g++ -O3 -march=native -pthread -o /tmp/host_expand_probe tools/probes/host_expand_probe.cpp && /tmp/host_expand_probe
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

int main(int argc, char** argv) {
  const size_t nq = 10000000, ns = 100000000;
  std::vector<uint32_t> lo(nq), cnt(nq);
  std::vector<int32_t> qrid(nq), srid(ns);
  std::vector<uint64_t> off(nq + 1);
  uint64_t x = 88172645463325252ull;
  auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (size_t i = 0; i < ns; i++) srid[i] = (int32_t)(rnd() % ns);
  uint64_t total = 0;
  for (size_t i = 0; i < nq; i++) {
    cnt[i] = 20 + (uint32_t)(rnd() % 41);
    lo[i] = (uint32_t)((i * 10ull + rnd() % 8) % (ns - 64));
    qrid[i] = (int32_t)(rnd() % nq);
    off[i] = total;
    total += cnt[i];
  }
  off[nq] = total;
  int32_t* ra = (int32_t*)aligned_alloc(4096, total * 4);
  int32_t* rb = (int32_t*)aligned_alloc(4096, total * 4);
  memset(ra, 0, total * 4);
  memset(rb, 0, total * 4);
  const unsigned hw = std::thread::hardware_concurrency();
  printf("pairs %llu (%.2f GB of output), hardware_concurrency %u\n", (unsigned long long)total, total * 8 / 1e9, hw);
  for (int T : {8, 16, 32, 64, 128}) {
    if ((unsigned)T > hw) break;
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
      const auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
          const size_t q0 = nq * (size_t)t / T, q1 = nq * (size_t)(t + 1) / T;
          for (size_t q = q0; q < q1; q++) {
            const uint64_t o = off[q];
            const uint32_t c = cnt[q], l = lo[q];
            const int32_t id = qrid[q];
            for (uint32_t k = 0; k < c; k++) {
              ra[o + k] = id;
              rb[o + k] = srid[l + k];
            }
          }
        });
      for (auto& t : th) t.join();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms < best) best = ms;
    }
    printf("threads %3d: %.1f ms = %.1f GB/s of pairs written\n", T, best, total * 8 / 1e6 / best);
  }
  return 0;
}
