// Host-side ingest for libmkpileup: BGZF/BAM reader (block-parallel inflate), FASTA (+.fai not
// required), BED.  This is the IO substrate the reference gets from rust-htslib / bio
// (src/pileup/mod.rs:732-743, src/fasta.rs:34,106-111, src/position_filter.rs:230-347); it feeds
// mkp_record views to the packer.  Whole-file residency (decompressed BAM kept in host RAM) is the
// round-1 design; BAI-indexed streaming is listed under "next" in DESIGN.md.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mkpileup.h"

namespace mkp {

struct Error : std::runtime_error {
  int status;
  Error(int st, const std::string& m) : std::runtime_error(m), status(st) {}
};

struct BamIndexEntry {  // one alignment record inside `raw`
  uint64_t off;         // offset of the record's 32-byte core (after block_size)
  int32_t tid, pos, end;  // end = bam_endpos (pos+1 for records without reference length)
  int32_t reflen;
  uint16_t flag;
};

// A byte buffer whose pages are first touched by whoever writes them (the inflate workers), not zero-filled up front;
// anonymous mapping with transparent huge pages requested (one fault per 2 MiB instead of per 4 KiB where the host allows it)
struct ByteBuf {
  uint8_t* p = nullptr; size_t n = 0, cap = 0;
  ByteBuf() = default;
  ByteBuf(ByteBuf&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
  ByteBuf& operator=(ByteBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; } return *this; }
  ByteBuf(const ByteBuf&) = delete; ByteBuf& operator=(const ByteBuf&) = delete;
  ~ByteBuf() { release(); }
  void release() { if (p) munmap(p, cap); p = nullptr; n = cap = 0; }
  void alloc(size_t bytes) {
    release();
    cap = (std::max<size_t>(bytes, 1) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    void* m = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) { cap = 0; throw Error(MKP_E_NOMEM, "out of host memory for the decompressed BAM"); }
    madvise(m, cap, MADV_HUGEPAGE);
    p = (uint8_t*)m; n = bytes;
  }
  size_t size() const { return n; }
  const uint8_t* data() const { return p; }
  uint8_t* data() { return p; }
  const uint8_t& operator[](size_t i) const { return p[i]; }
  uint8_t& operator[](size_t i) { return p[i]; }
};

// Read-only view of a whole file (mmap: the page cache is read in place)
struct FileMap {
  const uint8_t* p = nullptr; size_t n = 0;
  explicit FileMap(const std::string& path) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error(MKP_E_IO, "cannot open " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw Error(MKP_E_IO, "cannot stat " + path); }
    n = (size_t)st.st_size;
    if (n) {
      void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { close(fd); throw Error(MKP_E_IO, "cannot map " + path); }
      madvise(m, n, MADV_SEQUENTIAL);
      p = (const uint8_t*)m;
    }
    close(fd);
  }
  ~FileMap() { if (p) munmap((void*)p, n); }
  FileMap(const FileMap&) = delete; FileMap& operator=(const FileMap&) = delete;
  size_t size() const { return n; }
  const uint8_t& operator[](size_t i) const { return p[i]; }
};

struct BamData {
  std::vector<std::string> ref_names;
  std::vector<uint32_t> ref_lens;
  ByteBuf raw;  // decompressed stream
  std::vector<BamIndexEntry> recs;
  std::vector<size_t> tid_first;  // first record index per tid (+ sentinel), records are coordinate sorted

  int tid_of(const std::string& n) const { for (size_t i = 0; i < ref_names.size(); i++) if (ref_names[i] == n) return (int)i; return -1; }

  mkp_record view(const BamIndexEntry& e) const {
    const uint8_t* c = &raw[e.off];
    mkp_record r;
    memcpy(&r.tid, c, 4); memcpy(&r.pos, c + 4, 4);
    r.l_qname = c[8];
    uint16_t nc; memcpy(&nc, c + 12, 2); r.n_cigar = nc;
    memcpy(&r.flag, c + 14, 2);
    memcpy(&r.l_qseq, c + 16, 4);
    uint32_t bs; memcpy(&bs, c - 4, 4);
    r.l_data = (int32_t)bs - 32;
    r.data = c + 32;
    return r;
  }
};

static inline void inflate_block(const uint8_t* src, size_t clen, uint8_t* dst, size_t dlen) {
  z_stream zs; memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, -15) != Z_OK) throw Error(MKP_E_IO, "zlib init failed");
  zs.next_in = const_cast<Bytef*>(src); zs.avail_in = (uInt)clen; zs.next_out = dst; zs.avail_out = (uInt)dlen;
  int rc = inflate(&zs, Z_FINISH);
  inflateEnd(&zs);
  if (rc != Z_STREAM_END || zs.avail_out != 0) throw Error(MKP_E_IO, "corrupt BGZF block");
}

static inline BamData load_bam(const std::string& path, unsigned threads = 0) {
  FileMap comp(path);
  struct Blk { size_t coff, clen, doff, dlen; };
  std::vector<Blk> blks; size_t o = 0, dtotal = 0;
  while (o + 18 <= comp.size()) {
    if (comp[o] != 31 || comp[o + 1] != 139) throw Error(MKP_E_IO, "not BGZF: " + path);
    uint16_t xlen; memcpy(&xlen, &comp[o + 10], 2);
    size_t x = o + 12, xe = x + xlen; uint32_t bsize = 0; bool found = false;
    if (xe > comp.size()) throw Error(MKP_E_IO, "bad BGZF block in " + path);
    while (x + 4 <= xe) { uint16_t sl; memcpy(&sl, &comp[x + 2], 2); if (comp[x] == 'B' && comp[x + 1] == 'C' && sl == 2) { uint16_t b; memcpy(&b, &comp[x + 4], 2); bsize = (uint32_t)b + 1; found = true; } x += 4 + sl; }
    if (!found || o + bsize > comp.size() || bsize < (uint32_t)xlen + 20u) throw Error(MKP_E_IO, "bad BGZF block in " + path);
    uint32_t isize; memcpy(&isize, &comp[o + bsize - 4], 4);
    blks.push_back({o + 12 + xlen, bsize - xlen - 20, dtotal, isize});
    dtotal += isize; o += bsize;
  }
  BamData bd; bd.raw.alloc(dtotal);
  if (!threads) threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::atomic<size_t> next{0}; std::atomic<bool> bad{false};
  auto work = [&]() { for (;;) { size_t i = next++; if (i >= blks.size()) break; if (!blks[i].dlen) continue; try { inflate_block(&comp[blks[i].coff], blks[i].clen, &bd.raw[blks[i].doff], blks[i].dlen); } catch (...) { bad = true; } } };
  if (threads <= 1 || blks.size() < 4) work(); else { std::vector<std::thread> th; for (unsigned t = 0; t < threads; t++) th.emplace_back(work); for (auto& t : th) t.join(); }
  if (bad) throw Error(MKP_E_IO, "corrupt BGZF data in " + path);
  const ByteBuf& d = bd.raw; o = 0;
  auto need = [&](size_t n) { if (o + n > d.size()) throw Error(MKP_E_IO, "truncated BAM " + path); };
  need(12);
  if (memcmp(&d[0], "BAM\1", 4) != 0) throw Error(MKP_E_IO, "not a BAM file: " + path);
  int32_t l_text; memcpy(&l_text, &d[4], 4); o = 8; need((size_t)l_text + 4); o += (size_t)l_text;
  int32_t n_ref; memcpy(&n_ref, &d[o], 4); o += 4;
  for (int i = 0; i < n_ref; i++) {
    need(4); int32_t ln; memcpy(&ln, &d[o], 4); o += 4; need((size_t)ln + 4);
    bd.ref_names.push_back(std::string((const char*)&d[o], ln > 0 ? (size_t)ln - 1 : 0)); o += (size_t)ln;
    uint32_t lr; memcpy(&lr, &d[o], 4); o += 4; bd.ref_lens.push_back(lr);
  }
  while (o + 4 <= d.size()) {  // record boundaries: one hop per record
    int32_t bs; memcpy(&bs, &d[o], 4); o += 4; need((size_t)bs);
    if (bs < 32) throw Error(MKP_E_IO, "corrupt BAM record");
    BamIndexEntry e; e.off = o;
    memcpy(&e.tid, &d[o], 4); memcpy(&e.pos, &d[o + 4], 4);
    uint8_t lq = d[o + 8]; uint16_t nc; memcpy(&nc, &d[o + 12], 2); memcpy(&e.flag, &d[o + 14], 2);
    int32_t lseq; memcpy(&lseq, &d[o + 16], 4);
    if (lseq < 0 || (uint64_t)32 + lq + 4ull * nc + ((uint64_t)lseq + 1) / 2 + (uint64_t)lseq > (uint64_t)bs) throw Error(MKP_E_IO, "corrupt BAM record");
    if (e.tid < -1 || e.tid >= n_ref || e.pos < -1 || e.pos >= 0x7ffffff0) throw Error(MKP_E_IO, "corrupt BAM record: reference id or position out of range");
    e.reflen = 0; e.end = e.pos + 1;
    bd.recs.push_back(e); o += (size_t)bs;
  }
  {  // reference spans (bam_endpos): CIGAR walks, all cores
    const size_t n = bd.recs.size(); std::atomic<bool> span_bad{false};
    auto span = [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; i++) {
        BamIndexEntry& e = bd.recs[i];
        const uint8_t* c = &d[e.off]; uint16_t nc; memcpy(&nc, c + 12, 2);
        const uint8_t* cg = c + 32 + c[8]; int64_t rl = 0;
        for (uint16_t k = 0; k < nc; k++) { uint32_t w; memcpy(&w, cg + 4 * k, 4); uint32_t op = w & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += w >> 4; }
        if ((int64_t)e.pos + rl > 0x7ffffff0ll) { span_bad = true; rl = 0; }   // alignment runs past 2^31: corrupt record
        e.reflen = (int32_t)rl; e.end = e.pos + (rl > 0 ? (int32_t)rl : 1);
      }
    };
    const unsigned nt = n >= 4096 ? threads : 1u;
    if (nt <= 1) span(0, n);
    else { std::vector<std::thread> th; for (unsigned t = 0; t < nt; t++) th.emplace_back(span, n * t / nt, n * (t + 1) / nt); for (auto& t : th) t.join(); }
    if (span_bad) throw Error(MKP_E_IO, "corrupt BAM record: alignment runs past 2^31");
  }
  bd.tid_first.assign(bd.ref_names.size() + 1, bd.recs.size());
  for (size_t i = bd.recs.size(); i-- > 0;) { int t = bd.recs[i].tid; if (t >= 0 && (size_t)t < bd.ref_names.size()) bd.tid_first[(size_t)t] = i; }
  for (size_t t = bd.ref_names.size(); t-- > 0;) if (bd.tid_first[t] == bd.recs.size() && t + 1 <= bd.ref_names.size()) bd.tid_first[t] = bd.tid_first[t + 1];
  return bd;
}

// ------------------------------------------------------------------------------------ FASTA
struct Fasta {
  std::map<std::string, std::string> seqs;
  static Fasta load(const std::string& path) {
    Fasta f; std::ifstream in(path);
    if (!in) throw Error(MKP_E_IO, "cannot open fasta " + path);
    std::string line; std::string* cur = nullptr;
    while (std::getline(in, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (line.empty()) continue;
      if (line[0] == '>') { std::string n = line.substr(1); size_t sp = n.find_first_of(" \t"); if (sp != std::string::npos) n.resize(sp); cur = &f.seqs[n]; }
      else if (cur) cur->append(line);
    }
    return f;
  }
  const std::string* get(const std::string& name) const { auto it = seqs.find(name); return it == seqs.end() ? nullptr : &it->second; }
};

}  // namespace mkp
