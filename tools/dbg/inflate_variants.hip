// Ablation harness for the per-thread BGZF inflate kernels (not product code): times mkp_inflate_blocks / mkp_inflate_blocks2 over every
// block of a BAM, the second edition also with its memory traffic switched off piece by piece (MKP_INFLATE_DBG bits).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMKP_INFLATE_DBG -I modkit_amd/csrc -o /tmp/inflate_variants tools/dbg/inflate_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mkp_inflate.hip"

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
  fseek(f, 0, SEEK_END); const size_t n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> file(n + 64, 0); if (fread(file.data(), 1, n, f) != n) return 2; fclose(f);
  std::vector<MkpBgzfBlock> blks; size_t o = 0; unsigned long long total = 0;
  while (o + 18 <= n) { uint16_t xlen, bs; memcpy(&xlen, &file[o + 10], 2); memcpy(&bs, &file[o + 16], 2); const uint32_t bsize = (uint32_t)bs + 1; uint32_t isize; memcpy(&isize, &file[o + bsize - 4], 4);
    blks.push_back({o + 12 + xlen, total, bsize - xlen - 20, isize}); total += isize; o += bsize; }
  const size_t limit = argc > 2 ? (size_t)atol(argv[2]) : blks.size(); if (limit < blks.size()) { blks.resize(limit); total = blks.back().out_off + blks.back().out_len; }
  uint8_t *din, *dout; MkpBgzfBlock* dblk; uint32_t* dst;
  CK(hipMalloc(&din, n + 64)); CK(hipMalloc(&dout, total + 256)); CK(hipMalloc(&dblk, blks.size() * sizeof(MkpBgzfBlock))); CK(hipMalloc(&dst, blks.size() * 4));
  CK(hipMemcpy(din, file.data(), n + 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dblk, blks.data(), blks.size() * sizeof(MkpBgzfBlock), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t nb = (uint32_t)blks.size();
  printf("%u blocks, %.1f MB -> %.1f MB\n", nb, n / 1e6, total / 1e6);
  for (int kernel = 1; kernel <= 2; kernel++) for (uint32_t dbg : {0u, 1u, 2u, 3u}) {
    if (kernel == 1 && dbg) continue;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(mkp_inflate_dbg), &dbg, 4));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipMemset(dst, 0xff, nb * 4)); CK(hipEventRecord(e0, 0));
      CK(kernel == 1 ? mkp_launch_inflate(0, din, dblk, nb, dout, dst) : mkp_launch_inflate2(0, din, dblk, nb, dout, dst));
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    std::vector<uint32_t> st(nb); CK(hipMemcpy(st.data(), dst, nb * 4, hipMemcpyDeviceToHost)); size_t bad = 0; for (auto v : st) if (v) bad++;
    printf("kernel %d dbg %u: %.2f ms (%.1f GB/s out)%s\n", kernel, dbg, best, total / best / 1e6, dbg ? "" : bad ? "  BAD STATUS" : "  all ok");
  }
  return 0;
}
