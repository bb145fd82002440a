"""The library's bedMethyl row formatter (modkit_amd/csrc/mkp_format.hpp; used by mkp_pileup_main) against printf:
`{:.2}` of an f32 percentage must be the correctly rounded decimal (ties to even), as Rust prints it
(src/writers.rs:140).  Exhaustive over every (n_mod, n_valid) with n_valid <= 4096, plus f32 edge values."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "mkp_format.hpp"
int main() {
  char a[512], b[512]; unsigned long long n = 0;
  for (unsigned v = 1; v <= 4096; v++) for (unsigned m = 0; m <= v; m++) {
    const float pct = ((float)m / (float)v) * 100.0f;
    char* e = mkp::put_pct2(a, pct); *e = 0;
    snprintf(b, sizeof b, "%.2f", (double)pct);
    if (strcmp(a, b)) { printf("MISMATCH %u/%u: %s vs %s\n", m, v, a, b); return 1; }
    n++;
  }
  // hand-picked f32 values sitting on or next to rounding ties
  const float edge[] = {0.005f, 0.015f, 0.125f, 0.375f, 2.675f, 99.995f, 100.0f, 0.0f, 1e-7f, 33.335f, 66.665f, 50.125f, 12.345f, 0.285f, 1.005f};
  for (float x : edge) { char* e = mkp::put_pct2(a, x); *e = 0; snprintf(b, sizeof b, "%.2f", (double)x); if (strcmp(a, b)) { printf("MISMATCH edge %g: %s vs %s\n", x, a, b); return 1; } }
  // a whole row against the printf formulation
  char* e = mkp::format_row(a, "chr20", 5, "m,CG,0", 6, ' ', 1234567u, '-', 37, 11, 26, 0, 2, 5, 1, 3); *e = 0;
  const float pct = ((float)11 / (float)37) * 100.0f;
  snprintf(b, sizeof b, "%s\t%u\t%u\t%s\t%u\t%c\t%u\t%u\t255,0,0\t%u%c%.2f%c%u%c%u%c%u%c%u%c%u%c%u%c%u\n", "chr20", 1234567u, 1234568u, "m,CG,0", 37u, '-', 1234567u, 1234568u, 37u, ' ',
           (double)pct, ' ', 11u, ' ', 26u, ' ', 0u, ' ', 2u, ' ', 5u, ' ', 1u, ' ', 3u);
  if (strcmp(a, b)) { printf("MISMATCH row:\n%s%s", a, b); return 1; }
  // the decimal writer over every digit count and both short paths (pair table), alone and inside whole rows whose repeated columns
  // (start / end / coverage) are copied rather than converted twice
  unsigned long long x = 88172645463325252ull; auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (unsigned v = 0; v < 200000; v++) { char* q = mkp::put_u32(a, v); *q = 0; snprintf(b, sizeof b, "%u", v); if (strcmp(a, b)) { printf("MISMATCH u32 %u: %s vs %s\n", v, a, b); return 1; } }
  for (int k = 0; k < 200000; k++) {
    const unsigned v = (unsigned)(rnd() >> (rnd() % 33u + 31u)); char* q = mkp::put_u32(a, v); *q = 0; snprintf(b, sizeof b, "%u", v);
    if (strcmp(a, b)) { printf("MISMATCH u32 %u: %s vs %s\n", v, a, b); return 1; }
  }
  for (unsigned v : {0u, 9u, 10u, 99u, 100u, 999u, 1000u, 65535u, 99999u, 100000u, 999999999u, 1000000000u, 4294967294u, 4294967295u}) {
    char* q = mkp::put_u32(a, v); *q = 0; snprintf(b, sizeof b, "%u", v); if (strcmp(a, b)) { printf("MISMATCH u32 %u: %s vs %s\n", v, a, b); return 1; }
  }
  for (int k = 0; k < 100000; k++) {
    const unsigned pos = (unsigned)(rnd() % 4294967295ull), nv = 1u + (unsigned)(rnd() % (k & 1 ? 70000u : 60u)), nm = (unsigned)(rnd() % (nv + 1u));
    const unsigned c[6] = {(unsigned)(rnd() % 100u), (unsigned)(rnd() % 12u), (unsigned)(rnd() % 1000u), (unsigned)(rnd() % 3u), (unsigned)(rnd() % 70000u), (unsigned)(rnd() % 10u)};
    const char sp = (k & 2) ? ' ' : '\t', strand = (k & 4) ? '+' : '-';
    char* q = mkp::format_row(a, "chrUn_KI270742v1", 16, "21839,CG,0", 10, sp, pos, strand, nv, nm, nv - nm, c[0], c[1], c[2], c[3], c[4]); *q = 0;
    const float f = ((float)nm / (float)nv) * 100.0f;
    snprintf(b, sizeof b, "%s\t%u\t%u\t%s\t%u\t%c\t%u\t%u\t255,0,0\t%u%c%.2f%c%u%c%u%c%u%c%u%c%u%c%u%c%u\n", "chrUn_KI270742v1", pos, pos + 1u, "21839,CG,0", nv, strand, pos, pos + 1u, nv, sp,
             (double)f, sp, nm, sp, nv - nm, sp, c[0], sp, c[1], sp, c[2], sp, c[3], sp, c[4]);
    if (strcmp(a, b)) { printf("MISMATCH row %d:\n%s%s", k, a, b); return 1; }
  }
  printf("ok %llu\n", n);
  return 0;
}
'''


def test_percent_formatting_matches_printf_exhaustively(tmp_path):
    src = tmp_path / "fmt.cpp"
    src.write_text(SRC)
    exe = tmp_path / "fmt"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "modkit_amd", "csrc"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)], text=True)
    assert out.startswith("ok 8394752"), out


WRITER_SRC = r'''
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "mkp_writer.hpp"
using namespace mkp;
// three "shards" of rows through RowWriter (format on all cores + writer thread) against the one-thread formulation
int main(int argc, char** argv) {
  std::mt19937 rng(7);
  std::string ref;
  RowWriter wr; wr.mixed = argc > 3 && !strcmp(argv[3], "mixed"); wr.labels = {"CG,0", "CHH,0"};
  wr.f = fopen(argv[1], "w"); if (!wr.f) return 2;
  if (argc > 3 && !strcmp(argv[3], "bgzf")) { wr.bz.reset(new BgzfTabixSink()); wr.bz->f = wr.f; wr.bz->index_path = std::string(argv[1]) + ".tbi"; }
  const size_t sizes[3] = {200000, 10, 70001};
  const char* chroms[3] = {"chr1", "chrUn_KI270742v1", "chrM"};
  for (int s = 0; s < 3; s++) {
    const size_t n = sizes[s];
    std::vector<uint32_t> pos(n), code(n), nv(n), nm(n), nc(n), no(n), nd(n), nf(n), ndf(n), nn(n); std::vector<uint8_t> st(n); std::vector<int32_t> mi(n);
    for (size_t i = 0; i < n; i++) {
      pos[i] = (uint32_t)(i * 3 + rng() % 3); code[i] = (rng() % 5 == 0) ? (0x80000000u | (rng() % 100000u)) : (uint32_t)"mhac"[rng() % 4];
      nm[i] = rng() % 50; nc[i] = rng() % 50; no[i] = rng() % 3; nv[i] = nm[i] + nc[i] + no[i]; if (!nv[i]) { nc[i] = 1; nv[i] = 1; }
      nd[i] = rng() % 4; nf[i] = rng() % 9; ndf[i] = rng() % 5; nn[i] = rng() % 7; st[i] = (uint8_t)"+-."[rng() % 3]; mi[i] = (int32_t)(rng() % 3) - 1;
    }
    mkp_rows r; memset(&r, 0, sizeof(r));
    r.n_rows = n; r.pos = pos.data(); r.strand = st.data(); r.code_repr = code.data(); r.motif_idx = mi.data(); r.n_valid = nv.data(); r.n_mod = nm.data();
    r.n_canonical = nc.data(); r.n_other = no.data(); r.n_delete = nd.data(); r.n_fail = nf.data(); r.n_diff = ndf.data(); r.n_nocall = nn.data();
    wr.write(chroms[s], r);
    // reference formulation: one thread, one row at a time
    const char sp = wr.mixed ? ' ' : '\t';
    for (size_t i = 0; i < n; i++) {
      char name[96]; int k = (code[i] & 0x80000000u) ? snprintf(name, sizeof name, "%u", code[i] & 0x7fffffffu) : snprintf(name, sizeof name, "%c", (char)code[i]);
      if (mi[i] >= 0) k += snprintf(name + k, sizeof(name) - (size_t)k, ",%s", wr.labels[(size_t)mi[i]].c_str());
      char buf[512]; char* e = format_row(buf, chroms[s], strlen(chroms[s]), name, (size_t)k, sp, pos[i], (char)st[i], nv[i], nm[i], nc[i], no[i], nd[i], nf[i], ndf[i], nn[i]);
      ref.append(buf, (size_t)(e - buf));
    }
  }
  wr.finish(); fclose(wr.f);
  FILE* g = fopen(argv[2], "w"); fwrite(ref.data(), 1, ref.size(), g); fclose(g);
  printf("rows %llu bytes %zu\n", (unsigned long long)wr.n, ref.size());
  return 0;
}
'''


def _writer_harness(tmp_path, extra_flags, mixed):
    src = tmp_path / "wr.cpp"
    src.write_text(WRITER_SRC)
    exe = tmp_path / "wr"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread"] + extra_flags + ["-I", os.path.join(ROOT, "modkit_amd", "csrc"), "-o", str(exe), str(src), "-lz"])
    a, b = str(tmp_path / "a.bed"), str(tmp_path / "b.bed")
    out = subprocess.run([str(exe), a, b] + ([mixed] if isinstance(mixed, str) else ["mixed"] if mixed else []), capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("rows 270011"), out.stdout + out.stderr
    assert "ThreadSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-2000:]
    if mixed != "bgzf":
        assert open(a, "rb").read() == open(b, "rb").read()
    return a, b


def test_row_writer_threads_keep_order_and_bytes(tmp_path):
    _writer_harness(tmp_path, [], mixed=False)
    _writer_harness(tmp_path, [], mixed=True)


def test_row_writer_under_thread_sanitizer(tmp_path):
    probe = subprocess.run(["g++", "-fsanitize=thread", "-x", "c++", "-", "-o", str(tmp_path / "probe")], input="int main(){return 0;}", capture_output=True, text=True)
    if probe.returncode != 0:
        import pytest
        pytest.skip("no libtsan in this image")
    _writer_harness(tmp_path, ["-fsanitize=thread"], mixed=False)


# ---- --bgzf: BGZF blocks + TBI index (mkp_bgzf_out.hpp)
def _bgzf_blocks(data):
    import struct
    import zlib
    o, blocks = 0, []
    while o < len(data):
        assert data[o:o + 4] == b"\x1f\x8b\x08\x04"
        xlen = struct.unpack_from("<H", data, o + 10)[0]
        assert data[o + 12:o + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", data, o + 16)[0] + 1
        payload = zlib.decompress(data[o + 12 + xlen:o + bsize - 8], -15)
        crc, isize = struct.unpack_from("<II", data, o + bsize - 8)
        assert isize == len(payload) <= 65536 and crc == (zlib.crc32(payload) & 0xffffffff)
        blocks.append((o, payload))
        o += bsize
    return blocks


def _reg2bins(beg, end):
    end -= 1
    bins = [0]
    for shift, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        bins.extend(range(base + (beg >> shift), base + (end >> shift) + 1))
    return bins


def test_bgzf_output_and_tabix_index(tmp_path):
    import gzip
    import random
    import struct
    a, b = _writer_harness(tmp_path, [], mixed="bgzf")
    raw = open(a, "rb").read()
    text = open(b, "rb").read()
    assert gzip.decompress(raw) == text                    # a gzip reader sees the plain bedMethyl
    blocks = _bgzf_blocks(raw)
    assert blocks[-1][1] == b"" and all(len(p) <= 0xff00 for _, p in blocks)   # EOF marker; bgzip's block size
    by_off = {o: p for o, p in blocks}
    ix = b"".join(p for _, p in _bgzf_blocks(open(a + ".tbi", "rb").read()))
    magic, n_ref, fmt, sc, bc, ec, meta, skip, l_nm = struct.unpack_from("<4siiiiiiii", ix, 0)
    assert (magic, fmt, sc, bc, ec, meta, skip) == (b"TBI\x01", 0x10000, 1, 2, 3, ord("#"), 0)   # tabix -p bed
    names = ix[36:36 + l_nm].split(b"\0")[:-1]
    assert names == [b"chr1", b"chrUn_KI270742v1", b"chrM"] and n_ref == 3
    o = 36 + l_nm
    index = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", ix, o)[0]; o += 4
        bins = {}
        for _ in range(n_bin):
            bn, n_chunk = struct.unpack_from("<Ii", ix, o); o += 8
            bins[bn] = [struct.unpack_from("<QQ", ix, o + 16 * k) for k in range(n_chunk)]; o += 16 * n_chunk
        n_intv = struct.unpack_from("<i", ix, o)[0]; o += 4
        lidx = list(struct.unpack_from("<%dQ" % n_intv, ix, o)); o += 8 * n_intv
        index.append((bins, lidx))
    assert len(ix) - o == 8                                 # n_no_coor
    lines = text.split(b"\n")[:-1]
    per_ref = {nm: [] for nm in names}
    for ln in lines:
        f = ln.split(b"\t")
        per_ref[f[0]].append((int(f[1]), int(f[2]), ln))
    for ri, nm in enumerate(names):
        bins, lidx = index[ri]
        assert bins[37450][1] == (len(per_ref[nm]), 0)      # pseudo-bin: line count
        rng = random.Random(ri)
        span = per_ref[nm][-1][1]
        for _ in range(60):                                 # region queries through the index == a scan of the text
            beg = rng.randrange(0, span); end = min(span + 5, beg + rng.choice([1, 7, 300, 20000, 400000]))
            want = [ln for s0, e0, ln in per_ref[nm] if s0 < end and e0 > beg]
            min_off = lidx[beg >> 14] if (beg >> 14) < len(lidx) else 0
            got = []
            for bn in _reg2bins(beg, end):
                for c0, c1 in bins.get(bn, []):
                    if c1 <= min_off:
                        continue
                    v = c0
                    while v < c1:                           # read lines from virtual offset v
                        coff, uoff = v >> 16, v & 0xffff
                        payload = by_off[coff]
                        nl = payload.index(b"\n", uoff)
                        ln = payload[uoff:nl]
                        f = ln.split(b"\t")
                        if f[0] == nm and int(f[1]) < end and int(f[2]) > beg:
                            got.append((v, ln))
                        if nl + 1 < len(payload):
                            v = (coff << 16) | (nl + 1)
                        else:
                            v = (coff + len(raw[coff:coff + 18]) - 18 + struct.unpack_from("<H", raw, coff + 16)[0] + 1) << 16
            assert [ln for _, ln in sorted(set(got))] == want, (nm, beg, end)
