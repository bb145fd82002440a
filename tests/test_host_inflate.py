"""The device BGZF inflate kernel's decoder (modkit_amd/csrc/mkp_inflate.hip, product code) compiled for the HOST with one-thread
shims of the HIP built-ins and run block by block against zlib: every BGZF block of every BAM fixture (dynamic-Huffman blocks from
htslib / samtools), stored and fixed-Huffman blocks made with zlib, and corrupted blocks (must end in an error code, not hang or
overrun).  The same source runs on the GPU in tests/test_gpu_inflate.py."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>
#define MKP_INFLATE_HOST_SHIM 1
#define __device__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __restrict__
#define __launch_bounds__(x)
struct Idx { unsigned x; };
static Idx blockIdx{0}, threadIdx{0};
#include "mkp_inflate.hip"

static std::vector<uint8_t> slurp(const char* p) { FILE* f = fopen(p, "rb"); std::vector<uint8_t> v; if (!f) return v; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize((size_t)n); if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear(); fclose(f); return v; }
// one block through the kernel body (thread 0 of block 0)
static uint32_t run_block(const uint8_t* in, uint32_t in_len, std::vector<uint8_t>& out, uint32_t out_len) {
  MkpBgzfBlock b; b.in_off = 0; b.out_off = 0; b.in_len = in_len; b.out_len = out_len;
  out.assign((size_t)out_len + 64, 0xEE); uint32_t st = 99;
  mkp_inflate_blocks(in, &b, 1, out.data(), &st);
  for (size_t i = out_len; i < out.size(); i++) if (out[i] != 0xEE) return 100;   // wrote past its slice
  // the second edition (mkp_inflate_blocks2) must give the same status and bytes
  std::vector<uint8_t> out2((size_t)out_len + 64, 0xEE); uint32_t st2 = 99;
  mkp_inflate_blocks2(in, &b, 1, out2.data(), &st2);
  for (size_t i = out_len; i < out2.size(); i++) if (out2[i] != 0xEE) return 101;
  if ((st == 0) != (st2 == 0)) return 102;
  if (st == 0 && memcmp(out.data(), out2.data(), out_len) != 0) return 103;
  out.resize(out_len);
  return st;
}
static bool zinflate(const uint8_t* in, uint32_t in_len, std::vector<uint8_t>& out, uint32_t out_len) {
  z_stream zs; memset(&zs, 0, sizeof(zs)); if (inflateInit2(&zs, -15) != Z_OK) return false;
  out.assign((size_t)out_len + 1, 0); zs.next_in = const_cast<Bytef*>(in); zs.avail_in = in_len; zs.next_out = out.data(); zs.avail_out = out_len + 1;
  int rc = inflate(&zs, Z_FINISH); const bool ok = rc == Z_STREAM_END && zs.total_out == out_len; inflateEnd(&zs); out.resize(out_len); return ok;
}
int main(int argc, char** argv) {
  int fails = 0; size_t blocks = 0, bytes = 0;
  for (int a = 1; a < argc; a++) {   // every BGZF block of the given files
    std::vector<uint8_t> f = slurp(argv[a]); size_t o = 0;
    while (o + 18 <= f.size()) {
      uint16_t xlen; memcpy(&xlen, &f[o + 10], 2); uint16_t bs; memcpy(&bs, &f[o + 16], 2); const uint32_t bsize = (uint32_t)bs + 1;
      uint32_t isize; memcpy(&isize, &f[o + bsize - 4], 4);
      const uint8_t* payload = &f[o + 12 + xlen]; const uint32_t clen = bsize - xlen - 20;
      std::vector<uint8_t> got, want;
      const uint32_t st = run_block(payload, clen, got, isize);
      if (!zinflate(payload, clen, want, isize) || st != 0 || got != want) { printf("FAIL %s block at %zu: status %u\n", argv[a], o, st); fails++; }
      blocks++; bytes += isize; o += bsize;
    }
  }
  // stored, fixed-Huffman and dynamic blocks of synthetic data at several levels / strategies, split into several DEFLATE blocks
  srand(5);
  for (int it = 0; it < 60; it++) {
    const uint32_t n = 1 + (uint32_t)(rand() % 65000); std::vector<uint8_t> raw(n);
    for (uint32_t i = 0; i < n; i++) raw[i] = (it % 3 == 0) ? (uint8_t)rand() : (uint8_t)("ACGT,;0123"[rand() % 10] + (it % 3 == 1 && rand() % 50 == 0 ? 1 : 0));
    if (it % 4 == 1) for (uint32_t i = 300; i < n; i++) if (rand() % 3) raw[i] = raw[i - 1 - (uint32_t)(rand() % 299)];
    z_stream zs; memset(&zs, 0, sizeof(zs));
    const int level = it % 10, strategy = (it % 5 == 4) ? Z_FIXED : (it % 7 == 6 ? Z_HUFFMAN_ONLY : Z_DEFAULT_STRATEGY);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> comp(n + n / 2 + 1024); zs.next_in = raw.data(); zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
    uint32_t fed = 0; while (fed < n) { uint32_t step = 1 + (uint32_t)(rand() % 20000); if (step > n - fed) step = n - fed; zs.avail_in = step; deflate(&zs, (rand() % 2) ? Z_FULL_FLUSH : Z_NO_FLUSH); fed += step; }
    zs.avail_in = 0; deflate(&zs, Z_FINISH); const uint32_t clen = (uint32_t)zs.total_out; deflateEnd(&zs);
    std::vector<uint8_t> got; const uint32_t st = run_block(comp.data(), clen, got, n);
    if (st != 0 || got != raw) { printf("FAIL synthetic %d (level %d strategy %d): status %u\n", it, level, strategy, st); fails++; }
    // corruption: flipped bytes, truncation, wrong output size — an error code or (rarely) a clean decode of other bytes, never an overrun
    for (int k = 0; k < 6; k++) {
      std::vector<uint8_t> bad(comp.begin(), comp.begin() + clen); uint32_t blen = clen, want_n = n;
      if (k < 3) bad[(size_t)rand() % blen] ^= (uint8_t)(1 + rand() % 255); else if (k == 3) blen = (uint32_t)(rand() % blen); else if (k == 4) want_n = n + 1 + (uint32_t)(rand() % 50); else want_n = n > 1 ? n - 1 : 0;
      const uint32_t st2 = run_block(bad.data(), blen, got, want_n);
      if (st2 >= 99) { printf("FAIL corrupt %d/%d: status %u\n", it, k, st2); fails++; }
      if (k >= 3 && st2 == 0) { printf("FAIL corrupt %d/%d decoded cleanly\n", it, k); fails++; }
    }
  }
  // periodic data: matches at every short distance, 258 bytes long, chained (the second edition's pattern / cyclic / stepwise copies)
  for (int d = 1; d <= 48; d++) for (int lvl : {1, 6, 9}) {
    const uint32_t n = 3000 + (uint32_t)(rand() % 60000); std::vector<uint8_t> raw(n), per((size_t)d);
    for (auto& x : per) x = (uint8_t)rand();
    for (uint32_t i = 0; i < n; i++) raw[i] = per[i % (uint32_t)d];
    for (int k = 0; k < 40; k++) raw[(size_t)rand() % n] = (uint8_t)rand();   // a few breaks: literal runs and fresh matches in between
    z_stream zs; memset(&zs, 0, sizeof(zs)); deflateInit2(&zs, lvl, Z_DEFLATED, -15, 9, Z_DEFAULT_STRATEGY);
    std::vector<uint8_t> comp(n + 1024); zs.next_in = raw.data(); zs.avail_in = n; zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
    deflate(&zs, Z_FINISH); const uint32_t clen = (uint32_t)zs.total_out; deflateEnd(&zs);
    std::vector<uint8_t> got; const uint32_t st = run_block(comp.data(), clen, got, n);
    if (st != 0 || got != raw) { printf("FAIL periodic d=%d level %d: status %u\n", d, lvl, st); fails++; }
  }
  if (fails) printf("FAILED %d\n", fails); else printf("ok %zu blocks %zu bytes\n", blocks, bytes);
  return fails ? 1 : 0;
}
'''


def test_inflate_decoder_matches_zlib(tmp_path):
    src = tmp_path / "inflate.cpp"
    src.write_text(SRC)
    exe = tmp_path / "inflate"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "modkit_amd", "csrc"), "-x", "c++", "-o", str(exe), str(src), "-lz"])
    bams = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "modkit_fixtures", "*.bam")))
    assert len(bams) >= 8
    p = subprocess.run([str(exe)] + bams, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().startswith("ok"), p.stdout[-2000:] + p.stderr[-2000:]
