// host-only timing of the bedMethyl writer (mkp_writer.hpp) on synthetic rows: rows -> text on all cores -> file
// build: g++ -O2 -std=c++17 -I include -o /tmp/writer_bench tools/dbg/writer_bench.cpp -lpthread -lz ; run: /tmp/writer_bench <rows> <out>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../modkit_amd/csrc/mkp_writer.hpp"
int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2767406; const char* path = argc > 2 ? argv[2] : "/tmp/writer_bench.bed";
  std::vector<uint32_t> pos(n), code(n), v(n), m(n), c(n), o(n), d(n), fl(n), df(n), nc(n); std::vector<uint8_t> strand(n);
    std::vector<int32_t> motif(n);
  uint64_t x = 88172645463325252ull; auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 33); };
  for (uint64_t i = 0; i < n; i++) { pos[i] = (uint32_t)(i * 23 + rnd() % 20); code[i] = (i & 1) ? 'm' : 'h'; strand[i] = (i & 2) ? '+' : '-';
    motif[i] = -1; v[i] = 20 + rnd() % 20; m[i] = rnd() % 20; c[i] = v[i] - m[i]; o[i] = rnd() % 5; d[i] = rnd() % 3; fl[i] = rnd() % 4;
    df[i] = rnd() % 2; nc[i] = rnd() % 2; }
  mkp_rows r; memset(&r, 0, sizeof(r)); r.n_rows = n; r.pos = pos.data(); r.strand = strand.data(); r.code_repr = code.data();
    r.motif_idx = motif.data(); r.n_valid = v.data(); r.n_mod = m.data(); r.n_canonical = c.data(); r.n_other = o.data();
  r.n_delete = d.data(); r.n_fail = fl.data(); r.n_diff = df.data(); r.n_nocall = nc.data();
  for (int rep = 0; rep < 4; rep++) {
    FILE* f = fopen(path, "w"); if (!f) return 1;
    auto t0 = std::chrono::steady_clock::now();
    { mkp::RowWriter wr; wr.f = f; wr.write("chr20", r); auto t1 = std::chrono::steady_clock::now(); wr.finish();
      auto t2 = std::chrono::steady_clock::now();
      printf("rows %llu: write() returned %.1f ms, finish %.1f ms, total %.1f ms\n", (unsigned long long)n, std::chrono::duration<double,
          std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(), std::chrono::duration<double,
          std::milli>(t2 - t0).count());
        }
    fclose(f);
  }
  return 0;
}
