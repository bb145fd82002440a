// single-thread timing of the host BGZF inflate: the library's decoder vs zlib over every block of a BAM
// g++ -O2 -std=c++17 -I modkit_amd/csrc -I include tools/dbg/inflate_host_bench.cpp -lz -lpthread -o /tmp/ihb && /tmp/ihb in.bam
#include <chrono>
#include <cstdio>
#include <vector>
#include <zlib.h>
#include "mkp_inflate_host.hpp"
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<uint8_t> raw(n + 16);
    if (fread(raw.data(), 1, n, f) != n) return 1;
  struct B { size_t off, clen, isize; }; std::vector<B> bl; size_t o = 0, tot = 0;
  while (o + 18 <= n) { uint16_t xlen, bs; memcpy(&xlen, &raw[o + 10], 2); memcpy(&bs, &raw[o + 16], 2); uint32_t isz;
    memcpy(&isz, &raw[o + bs + 1 - 4], 4);
    bl.push_back({o + 12 + xlen, (size_t)bs + 1 - xlen - 20, isz}); tot += isz; o += (size_t)bs + 1; }
  std::vector<uint8_t> a(1 << 16), b(1 << 16);
  for (int rep = 0; rep < 2; rep++) {
    auto t0 = std::chrono::steady_clock::now(); size_t declined = 0;
    for (auto& x : bl) if (!mkp::hostinf::inflate(&raw[x.off], x.clen, a.data(), x.isize)) declined++;
    auto t1 = std::chrono::steady_clock::now();
    for (auto& x : bl) { z_stream zs; memset(&zs, 0, sizeof(zs)); inflateInit2(&zs, -15); zs.next_in = &raw[x.off]; zs.avail_in = x.clen;
      zs.next_out = b.data(); zs.avail_out = x.isize; inflate(&zs, Z_FINISH); inflateEnd(&zs); }
    auto t2 = std::chrono::steady_clock::now();
    double s1 = std::chrono::duration<double>(t1 - t0).count(), s2 = std::chrono::duration<double>(t2 - t1).count();
    printf("%zu blocks, %.1f MB inflated: own %.3f s (%.0f MB/s, %zu declined), zlib %.3f s (%.0f MB/s)\n", bl.size(), tot / 1e6, s1, tot / 1e6 / s1,
        declined, s2, tot / 1e6 / s2);
  }
}
