// what the runtime calls of a fresh context cost on this box: stream creation (by priority), event creation, page-locked and device allocations
// build: hipcc --offload-arch=gfx950 -O2 -o tools/dbg/api_cost tools/dbg/api_cost.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double ms(std::chrono::steady_clock::time_point a) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }
__global__ void k(int* p) { if (p) p[0] = 1; }
int main() {
  auto t = std::chrono::steady_clock::now(); hipFree(0); printf("runtime init %.1f ms\n", ms(t));
  int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
  std::vector<hipStream_t> ss;
  for (int i = 0; i < 10; i++) { hipStream_t s; t = std::chrono::steady_clock::now();
    hipStreamCreateWithPriority(&s, hipStreamNonBlocking, i % 3 == 0 ? hi : i % 3 == 1 ? (lo + hi) / 2 : lo); const double a = ms(t);
    t = std::chrono::steady_clock::now(); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, (int*)nullptr); hipStreamSynchronize(s);
      printf("stream %d: create %.2f ms, first launch + sync %.2f ms\n", i, a, ms(t)); ss.push_back(s); }
  { // four more streams from four threads at once: does creation run in parallel?
    t = std::chrono::steady_clock::now(); std::vector<std::thread> th; hipStream_t ps[4];
    for (int i = 0; i < 4; i++) th.emplace_back([&, i] { hipSetDevice(0); hipStreamCreateWithPriority(&ps[i], hipStreamNonBlocking, lo); });
    for (auto& x : th) x.join(); printf("4 streams from 4 threads: %.2f ms wall\n", ms(t)); }
  t = std::chrono::steady_clock::now(); std::vector<hipEvent_t> ev(32); for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    printf("32 events %.2f ms\n", ms(t));
  for (size_t mb : {1, 16, 32, 64, 128}) { void* p = nullptr; t = std::chrono::steady_clock::now(); hipHostMalloc(&p, mb << 20, hipHostMallocDefault);
    const double a = ms(t); t = std::chrono::steady_clock::now(); hipHostFree(p);
    printf("hipHostMalloc %zu MiB %.2f ms, free %.2f ms\n", mb, a, ms(t)); }
  for (size_t mb : {16, 1024, 5120}) { void* p = nullptr; t = std::chrono::steady_clock::now(); hipMalloc(&p, mb << 20); const double a = ms(t);
    t = std::chrono::steady_clock::now(); hipFree(p); printf("hipMalloc %zu MiB %.2f ms, free %.2f ms\n", mb, a, ms(t)); }
  return 0;
}
