// bgzip + tabix form of the bedMethyl output ("bgzip+tabix of the output").  The reference writes plain text and leaves
// compression to `bgzip` / `tabix -p bed` (htslib); with `--bgzf` the driver writes what that pair would: a BGZF file (SAM spec 4.1)
// and its TBI index (tabix format spec: header with the BED preset, per contig the binning index over 16 kb .. 512 Mb bins, the 16 kb
// linear index, the htslib pseudo-bin).  Blocks are cut at line boundaries and compressed by the threads that format the rows; the
// writer thread lays them out in file order and extends the index line by line.  Host-only C++ (tests/test_format_cpu.py).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace mkp {

static const size_t MKP_BGZF_BLOCK = 0xff00;   // uncompressed bytes per block, as bgzip cuts them

// text[0..n) -> one BGZF block appended to `out`; returns its compressed size.  A zlib failure (or a block that does not fit the
// 64 KiB BGZF limit) leaves `out` as it was and throws: a half-written block must never reach the file or the index.
static inline uint32_t bgzf_block(const uint8_t* text, size_t n, std::vector<uint8_t>* out, int level = 6) {
  const size_t at = out->size(); out->resize(at + n + n / 8 + 64);
  z_stream zs; memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { out->resize(at);
    throw std::runtime_error("bgzf: deflateInit2 failed"); }
  zs.next_in = const_cast<Bytef*>(text); zs.avail_in = (uInt)n; zs.next_out = out->data() + at + 18; zs.avail_out = (uInt)(out->size() - at - 26);
  const int rc = deflate(&zs, Z_FINISH); const size_t clen = zs.total_out; deflateEnd(&zs);
  if (rc != Z_STREAM_END || clen + 26 > 65536) { out->resize(at); throw std::runtime_error("bgzf: block does not compress into 64 KiB"); }
  static const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0};
  memcpy(out->data() + at, hdr, 16);
  const uint16_t bsize = (uint16_t)(clen + 25); memcpy(out->data() + at + 16, &bsize, 2);
  const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), text, (uInt)n), isz = (uint32_t)n;
  memcpy(out->data() + at + 18 + clen, &crc, 4); memcpy(out->data() + at + 22 + clen, &isz, 4);
  out->resize(at + clen + 26);
  return (uint32_t)(clen + 26);
}

// What a formatting thread hands to the writer thread in --bgzf mode: its rows' text as whole-line BGZF blocks, and per line where it sits
struct BgzfPiece {
  std::vector<uint8_t> comp;                       // the blocks, back to back
  std::vector<uint32_t> block_csize, block_usize;
  struct Line { uint32_t pos, block, uoff, len; };
  std::vector<Line> lines;
  // text = n_lines lines; line_len[i] bytes each (incl. the newline); pos[i] = chromStart
  void build(const char* text, const uint32_t* line_len, const uint32_t* pos, size_t n_lines) {
    size_t i = 0, off = 0;
    while (i < n_lines) {
      size_t j = i, bytes = 0;
      while (j < n_lines && bytes + line_len[j] <= MKP_BGZF_BLOCK) {
        lines.push_back({pos[j], (uint32_t)block_csize.size(), (uint32_t)bytes, line_len[j]}); bytes += line_len[j]; j++; }
      // (a line longer than a block cannot happen for bedMethyl rows)
      if (j == i) { lines.push_back({pos[j], (uint32_t)block_csize.size(), 0u, line_len[j]}); bytes = line_len[j]; j++; }
      const uint32_t cs = bgzf_block(reinterpret_cast<const uint8_t*>(text) + off, bytes, &comp);
      block_csize.push_back(cs); block_usize.push_back((uint32_t)bytes);
      off += bytes; i = j;
    }
  }
};

class BgzfTabixSink {
 public:
  FILE* f = nullptr; std::string index_path; bool failed = false;
  uint64_t file_off = 0;
  struct Ref { std::string name; std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins; std::vector<uint64_t> lidx; uint64_t first = 0,
      last = 0, n = 0; };
  std::vector<Ref> refs;
  static uint32_t reg2bin(int64_t beg, int64_t end) {  // SAM spec 5.3
    --end;
    if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
  }
  // one piece (rows of one contig, file order)
  void write(const std::string& chrom, const BgzfPiece& p) {
    if (p.comp.empty()) return;
    if (refs.empty() || refs.back().name != chrom) { Ref r; r.name = chrom; refs.push_back(r); }
    Ref& R = refs.back();
    std::vector<uint64_t> boff(p.block_csize.size() + 1); boff[0] = file_off;
    for (size_t b = 0; b < p.block_csize.size(); b++) boff[b + 1] = boff[b] + p.block_csize[b];
    if (fwrite(p.comp.data(), 1, p.comp.size(), f) != p.comp.size()) failed = true;
    file_off = boff.back();
    for (const BgzfPiece::Line& L : p.lines) {
      const uint64_t v0 = (boff[L.block] << 16) | L.uoff;
      const uint64_t v1 = (L.uoff + L.len < p.block_usize[L.block]) ? ((boff[L.block] << 16) | (L.uoff + L.len)) : (boff[L.block + 1] << 16);
      const int64_t beg = L.pos, end = (int64_t)L.pos + 1;
      auto& ch = R.bins[reg2bin(beg, end)];
      if (!ch.empty() && ch.back().second == v0) ch.back().second = v1; else ch.push_back({v0, v1});   // consecutive lines of one bin form one chunk
      const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
      if (R.lidx.size() <= w1) R.lidx.resize(w1 + 1, 0);
      for (size_t w = w0; w <= w1; w++) if (R.lidx[w] == 0) R.lidx[w] = v0;
      if (R.n == 0) R.first = v0;
      R.last = v1; R.n++;
    }
  }
  // EOF block + the .tbi
  void finish() {
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (fwrite(eof, 1, 28, f) != 28) failed = true;
    std::vector<uint8_t> ix; auto put = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; ix.insert(ix.end(), b, b + n); };
    auto i32 = [&](int32_t v) { put(&v, 4); }; auto u32 = [&](uint32_t v) { put(&v, 4); }; auto u64 = [&](uint64_t v) { put(&v, 8); };
    put("TBI\1", 4); i32((int32_t)refs.size());
    // `tabix -p bed`: TBX_UCSC, sequence / begin / end columns, comment character, lines to skip
    i32(0x10000); i32(1); i32(2); i32(3); i32('#'); i32(0);
    std::string names; for (auto& r : refs) { names += r.name; names.push_back('\0'); }
    i32((int32_t)names.size()); put(names.data(), names.size());
    for (auto& r : refs) {
      i32((int32_t)r.bins.size() + 1);
      for (auto& kv : r.bins) { u32(kv.first); i32((int32_t)kv.second.size()); for (auto& c : kv.second) { u64(c.first); u64(c.second); } }
      u32(37450); i32(2); u64(r.first); u64(r.last); u64(r.n); u64(0);   // htslib's pseudo-bin: the contig's extent in the file and its line count
      // empty windows take the next window's offset, as htslib fills them
      for (size_t w = r.lidx.size(); w-- > 1;) if (r.lidx[w - 1] == 0) r.lidx[w - 1] = r.lidx[w];
      i32((int32_t)r.lidx.size()); for (uint64_t v : r.lidx) u64(v);
    }
    u64(0);   // n_no_coor
    FILE* fi = fopen(index_path.c_str(), "wb");
    if (!fi) { failed = true; return; }
    std::vector<uint8_t> comp;
    try { for (size_t o = 0; o < ix.size(); o += MKP_BGZF_BLOCK) bgzf_block(ix.data() + o, std::min(MKP_BGZF_BLOCK, ix.size() - o), &comp); }
    catch (const std::exception&) { failed = true; fclose(fi); remove(index_path.c_str()); return; }
    comp.insert(comp.end(), eof, eof + 28);
    if (fwrite(comp.data(), 1, comp.size(), fi) != comp.size()) failed = true;
    fclose(fi);
  }
};

}  // namespace mkp
