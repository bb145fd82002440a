// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU against known byte counts, in the access shapes libmkpileup uses:
// MI355X_MICROARCH.md says FETCH_SIZE reports half the bytes of a wide coalesced streaming read on gfx950 and that other widths are
// uncalibrated.  Four kernels over one 1 GiB buffer (far beyond the 256 MiB Infinity Cache):
//   calib_stream16   every lane reads 16 B, coalesced                      -> 1 GiB read
//   calib_byte_s32   every lane reads 1 byte of each 32 B                  -> every 32 B sector touched once
//   calib_byte_s24   every lane reads 1 byte every 24 B (a --cpg walk over 4-bit SEQ touches about that)
//   calib_write16    every lane writes 16 B, coalesced                      -> 1 GiB written
// Build + run: tools/pmc_calib.sh (two rocprofv3 --pmc passes); prints counter KiB / true KiB per kernel.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
extern "C" __global__ void calib_stream16(const uint4* __restrict__ p, size_t n16, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
template <int STRIDE> __device__ void byte_walk(const uint8_t* __restrict__ p, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * STRIDE; i < n; i += (size_t)gridDim.x * blockDim.x * STRIDE) acc += p[i];
  if (acc == 0x12345678u) *sink = acc;
}
extern "C" __global__ void calib_byte_s32(const uint8_t* __restrict__ p, size_t n, uint32_t* sink) { byte_walk<32>(p, n, sink); }
extern "C" __global__ void calib_byte_s24(const uint8_t* __restrict__ p, size_t n, uint32_t* sink) { byte_walk<24>(p, n, sink); }
extern "C" __global__ void calib_write16(uint4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1u, 2u,
      3u);
}
int main() {
  const size_t bytes = 1ull << 30;
  uint8_t* buf; uint32_t* sink;
  CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 4)); CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(calib_stream16, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(calib_byte_s32, dim3(4096), dim3(256), 0, 0, buf, bytes, sink);
    hipLaunchKernelGGL(calib_byte_s24, dim3(4096), dim3(256), 0, 0, buf, bytes, sink);
    hipLaunchKernelGGL(calib_write16, dim3(4096), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
    CHECK(hipDeviceSynchronize());
  }
  printf("buffer_bytes %zu\n", bytes);
  return 0;
}
