// Microbenchmark: what a grid of near-empty workgroups costs on gfx950 (wave launch rate, LDS allocation, one dependent load).
// Sizes mirror the step kernels: 46 574 x 256 threads + 24 KB LDS (mkp_decode_slots), 1 966 x 1024 threads + 60 KB LDS (mkp_pileup_stream).
// hipcc --offload-arch=gfx950 -O3 -o launch_rate launch_rate.hip && ./launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LDSW> __global__ void __launch_bounds__(256) k_empty256(uint32_t* out) {
  __shared__ uint32_t l[LDSW > 0 ? LDSW : 1];
  if (LDSW > 0) { l[threadIdx.x] = threadIdx.x; __syncthreads(); if (l[(threadIdx.x + 1) & 255] == 0xdeadbeefu) out[0] = 1; }
}
template <int LDSW> __global__ void __launch_bounds__(256) k_load256(const uint4* __restrict__ rec, uint32_t* out) {
  __shared__ uint32_t l[LDSW > 0 ? LDSW : 1];
  const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint4 r = rec[w * 4];   // one 64-byte record per wave (first 16 bytes)
  if (LDSW > 0) l[threadIdx.x] = r.x;
  if ((threadIdx.x & 63) == 0) out[w * 8] = r.x + r.y;   // one 32-byte record out per wave
}
// persistent form: the same per-wave work, waves pull indices from an atomic
template <int LDSW> __global__ void __launch_bounds__(256) k_persist256(const uint4* __restrict__ rec, uint32_t* out, uint32_t* counter, uint32_t n) {
  __shared__ uint32_t l[LDSW > 0 ? LDSW : 1];
  for (;;) {
    uint32_t w = 0; if ((threadIdx.x & 63) == 0) w = atomicAdd(counter, 1u);
    w = __builtin_amdgcn_readfirstlane(w);
    if (w >= n) break;
    const uint4 r = rec[w * 4];
    if (LDSW > 0) l[threadIdx.x] = r.x;
    if ((threadIdx.x & 63) == 0) out[w * 8] = r.x + r.y;
  }
}
__global__ void __launch_bounds__(1024) k_empty1024(uint32_t* out, uint32_t clear_words) {
  extern __shared__ uint32_t dl[];
  for (uint32_t k = threadIdx.x; k < clear_words; k += 1024) dl[k] = 0;
  __syncthreads();
  if (dl[threadIdx.x] == 0xdeadbeefu) out[0] = 1;
}
template <int T> __global__ void __launch_bounds__(T) k_emptyT(uint32_t* out, uint32_t clear_words) {
  extern __shared__ uint32_t dl[];
  for (uint32_t k = threadIdx.x; k < clear_words; k += T) dl[k] = 0;
  __syncthreads();
  if (dl[threadIdx.x] == 0xdeadbeefu) out[0] = 1;
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  uint4* rec; uint32_t* out; uint32_t* ctr;
  const uint32_t NW = 186296;
  CK(hipMalloc(&rec, (size_t)NW * 64)); CK(hipMalloc(&out, (size_t)NW * 32)); CK(hipMalloc(&ctr, 64));
  CK(hipMemset(rec, 0, (size_t)NW * 64)); CK(hipMemset(out, 0, (size_t)NW * 32));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; i++) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; i++) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %.4f ms per launch\n", name, ms / 20);
    return 0;
  };
  const uint32_t G = (NW + 3) / 4;
  timeit("empty 46574 x 256 thr, no LDS", [&] { hipLaunchKernelGGL(k_empty256<0>, dim3(G), dim3(256), 0, 0, out); });
  timeit("empty 46574 x 256 thr, 24 KB LDS", [&] { hipLaunchKernelGGL(k_empty256<6144>, dim3(G), dim3(256), 0, 0, out); });
  timeit("empty 46574 x 256 thr, 12 KB LDS", [&] { hipLaunchKernelGGL(k_empty256<3072>, dim3(G), dim3(256), 0, 0, out); });
  timeit("load+store 46574 x 256 thr, no LDS", [&] { hipLaunchKernelGGL(k_load256<0>, dim3(G), dim3(256), 0, 0, rec, out); });
  timeit("load+store 46574 x 256 thr, 24 KB LDS", [&] { hipLaunchKernelGGL(k_load256<6144>, dim3(G), dim3(256), 0, 0, rec, out); });
  for (uint32_t wg : {256u * 4, 256u * 6, 256u * 8}) {
    char nm[128]; snprintf(nm, sizeof nm, "persistent load+store, %u WGs x 256 thr, 24 KB LDS (+memset)", wg);
    timeit(nm, [&] { hipMemsetAsync(ctr, 0, 4, 0); hipLaunchKernelGGL(k_persist256<6144>, dim3(wg), dim3(256), 0, 0, rec, out, ctr, NW); });
  }
  timeit("memset 4 bytes alone", [&] { hipMemsetAsync(ctr, 0, 4, 0); });
  CK(hipFuncSetAttribute((const void*)k_empty1024, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  timeit("empty 1966 x 1024 thr, 60 KB LDS, no clear", [&] { hipLaunchKernelGGL(k_empty1024, dim3(1966), dim3(1024), 61440, 0, out, 0u); });
  timeit("empty 1966 x 1024 thr, 60 KB LDS, clear 45 KB", [&] { hipLaunchKernelGGL(k_empty1024, dim3(1966), dim3(1024), 61440, 0, out, 11520u); });
  CK(hipFuncSetAttribute((const void*)k_emptyT<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void*)k_emptyT<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  timeit("empty 3932 x 512 thr, 30 KB LDS, clear 22 KB", [&] { hipLaunchKernelGGL(k_emptyT<512>, dim3(3932), dim3(512), 30720, 0, out, 5760u); });
  timeit("empty 7864 x 256 thr, 15 KB LDS, clear 11 KB", [&] { hipLaunchKernelGGL(k_emptyT<256>, dim3(7864), dim3(256), 15360, 0, out, 2880u); });
  timeit("empty 512 x 1024 thr, 60 KB LDS, clear 45 KB", [&] { hipLaunchKernelGGL(k_empty1024, dim3(512), dim3(1024), 61440, 0, out, 11520u); });
  timeit("empty 256 x 256 thr", [&] { hipLaunchKernelGGL(k_empty256<0>, dim3(256), dim3(256), 0, 0, out); });
  return 0;
}
