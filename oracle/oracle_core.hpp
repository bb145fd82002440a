// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product; nothing under
// modkit_amd/ may include, link or execute this.  Only tests/, the smoke() check
// and bench.py's cpu_baseline leg use it, and only as the checker.
//
// CPU restatement of nanoporetech/modkit v0.4.4 `modkit pileup` (reference at
// /root/reference, cited as file:line below).  The reference is a Rust crate and
// cannot be built in this image (no cargo/rustc/htslib), so this restatement is
// pinned against the reference's own golden bedMethyl fixtures
// (tests/test_pileup.rs) — see tests/test_oracle_golden.py.
//
// Third-party semantics restated from their published behaviour (not in tree):
//   rust-htslib 0.46 / htslib pileup engine  (default flag mask, D/N columns)
//   rustc-hash 1.1 FxHasher + hashbrown       (iteration order of small maps)
//   rust-lapper 1.1                           (merge_overlaps / find)
//   bio 1.0 fasta::IndexedReader              (fetch/read)
//   rand 0.8.5 StdRng (rand_chacha 0.3 ChaCha12, rand_core 0.6 seed_from_u64, Bernoulli)   (--seed with --sampling-frac < 1)
//
// Parity unpinned by any reference fixture (derived from code only): ties in
// mod-code probability, >=3 codes per base, ChEBI ordering (derived Ord: Code <
// ChEbi, src/mod_base_code.rs:105), max_depth read dropping (bam_plp_push, restated in oracle_pileup.hpp), QC-fail reads, N CIGAR
// ops, boundary-CpG loss, schedule pruning order (sampling_schedule.rs:225).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace mko {

// ---------------------------------------------------------------- basic types
enum Base : int { BA = 0, BC = 1, BG = 2, BT = 3 };
static inline int base_from_char(char c) {
  switch (c) {
    case 'A': return BA;
    case 'C': return BC;
    case 'G': return BG;
    case 'T': return BT;
    default: return -1;
  }
}
static inline char base_char(int b) { return "ACGT"[b]; }
static inline int complement(int b) { return 3 - b; }  // mod_base_code.rs:198-205

// ModCodeRepr (mod_base_code.rs:105-109).  Code(c) -> c ; ChEbi(x) -> 0x80000000|x.
// Numeric order on this encoding == the *derived* Ord (variant index first:
// Code < ChEbi), which is what `.cmp()` resolves to in pileup/mod.rs:441 and
// the BTreeMap at pileup/mod.rs:526.
typedef uint32_t ModCode;
static inline ModCode code_char(char c) { return (uint32_t)(unsigned char)c; }
static inline ModCode code_chebi(uint32_t x) { return 0x80000000u | x; }
static inline bool is_chebi(ModCode m) { return (m & 0x80000000u) != 0; }
static inline std::string code_str(ModCode m) {
  if (is_chebi(m)) return std::to_string(m & 0x7fffffffu);
  return std::string(1, (char)m);
}
static inline bool parse_mod_code(const std::string& s, ModCode* out) {
  // ModCodeRepr::parse (mod_base_code.rs:112-122): one char, else u32
  if (s.size() == 1) { *out = code_char(s[0]); return true; }
  if (s.empty()) return false;
  uint64_t v = 0;
  for (char c : s) {
    if (c < '0' || c > '9') return false;
    v = v * 10 + (uint64_t)(c - '0');
    if (v > 0xffffffffull) return false;
  }
  *out = code_chebi((uint32_t)v);
  return true;
}

struct MkErr : std::runtime_error {
  explicit MkErr(const std::string& s) : std::runtime_error(s) {}
};

// ----------------------------------------------- FxHashMap<ModCodeRepr, f32>
// rustc-hash 1.1: hash = (rotl(hash,5) ^ word) * 0x517cc1b727220a95 per written
// word; derive(Hash) writes the discriminant (isize) then the payload (u32).
static inline uint64_t fx_add(uint64_t h, uint64_t w) {
  return (((h << 5) | (h >> 59)) ^ w) * 0x517cc1b727220a95ull;
}
static inline uint64_t fx_hash_code(ModCode m) {
  uint64_t h = 0;
  h = fx_add(h, is_chebi(m) ? 1 : 0);
  h = fx_add(h, is_chebi(m) ? (m & 0x7fffffffu) : m);
  return h;
}

// hashbrown RawTable restated for <=14 live entries, insert-only.  Iteration is
// by ascending bucket; placement = first free bucket at or after (hash & mask),
// wrapping (Group::WIDTH 16 on x86-64 makes the probe linear for <=16 buckets).
// Growth 0 -> 4 -> 8 -> 16 buckets (capacity 3 / 7 / 14), re-inserting in
// iteration order.  Pinned only for {h,m} (mod_bam.rs:2250-2258).
struct FxProbMap {
  int nb = 0;  // buckets
  int n = 0;   // items
  ModCode keys[16];
  float vals[16];
  bool used[16];
  FxProbMap() { memset(used, 0, sizeof(used)); }
  static int cap(int nb) { return nb == 0 ? 0 : (nb < 8 ? nb - 1 : nb / 8 * 7); }
  int find(ModCode k) const {
    for (int i = 0; i < nb; i++)
      if (used[i] && keys[i] == k) return i;
    return -1;
  }
  void place(ModCode k, float v) {
    int pos = (int)(fx_hash_code(k) & (uint64_t)(nb - 1));
    for (int s = 0; s < nb; s++) {
      int i = (pos + s) & (nb - 1);
      if (!used[i]) { used[i] = true; keys[i] = k; vals[i] = v; return; }
    }
    throw MkErr("fxmap full");
  }
  // entry(k).or_insert(v): returns bucket index
  int entry(ModCode k, float v_if_new) {
    int i = find(k);
    if (i >= 0) return i;
    if (n == cap(nb)) {
      int nnb = nb == 0 ? 4 : nb * 2;
      if (nnb > 16) throw MkErr("more than 14 mod codes on one base: unsupported by oracle");
      ModCode ok[16]; float ov[16]; int on = 0;
      for (int j = 0; j < nb; j++) if (used[j]) { ok[on] = keys[j]; ov[on] = vals[j]; on++; }
      nb = nnb; memset(used, 0, sizeof(used));
      for (int j = 0; j < on; j++) place(ok[j], ov[j]);
    }
    place(k, v_if_new);
    n++;
    return find(k);
  }
  // insert(k,v) -> previous value if any (returns true if replaced)
  bool insert(ModCode k, float v, float* prev) {
    int i = find(k);
    if (i >= 0) { *prev = vals[i]; vals[i] = v; return true; }
    entry(k, v);
    return false;
  }
  template <class F> void for_each(F f) const {
    for (int i = 0; i < nb; i++) if (used[i]) f(keys[i], vals[i]);
  }
  float sum() const {  // values().sum::<f32>() in iteration order, from 0f32
    float s = 0.0f;
    for (int i = 0; i < nb; i++) if (used[i]) s = s + vals[i];
    return s;
  }
};

// ------------------------------------------------------------------ BAM input
struct BamRecord {
  int32_t tid = -1, pos = -1;
  uint16_t flag = 0;
  int32_t l_seq = 0;
  std::string qname;
  std::vector<uint32_t> cigar;  // len<<4|op  (MIDNSHP=X)
  std::string seq;              // ASCII, as stored (reference orientation)
  std::vector<uint8_t> qual;    // base qualities, as stored
  std::vector<uint8_t> aux;
  int32_t ref_len() const {
    int64_t l = 0;
    for (uint32_t c : cigar) {
      int op = c & 15;
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) l += c >> 4;
    }
    return (int32_t)l;
  }
  int32_t end_pos() const { int32_t l = ref_len(); return pos + (l > 0 ? l : 1); }  // bam_endpos
  bool is_reverse() const { return flag & 16; }
};
struct BamFile {
  std::vector<std::string> ref_names;
  std::vector<uint32_t> ref_lens;
  std::vector<BamRecord> recs;  // file order
  int tid_of(const std::string& name) const {
    for (size_t i = 0; i < ref_names.size(); i++) if (ref_names[i] == name) return (int)i;
    return -1;
  }
};

static inline std::vector<uint8_t> read_gz_all(const std::string& path) {
  gzFile f = gzopen(path.c_str(), "rb");
  if (!f) throw MkErr("cannot open " + path);
  gzbuffer(f, 1 << 20);
  std::vector<uint8_t> out;
  std::vector<uint8_t> buf(1 << 22);
  for (;;) {
    int n = gzread(f, buf.data(), (unsigned)buf.size());
    if (n < 0) { gzclose(f); throw MkErr("gz read error " + path); }
    if (n == 0) break;
    out.insert(out.end(), buf.begin(), buf.begin() + n);
  }
  gzclose(f);
  return out;
}

template <class F> static inline void oracle_parallel_for(size_t n, F f) {
  unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (n < 64) nt = 1;
  std::atomic<size_t> next{0};
  auto work = [&]() { for (;;) { size_t i = next.fetch_add(64); if (i >= n) break; for (size_t k = i; k < std::min(n, i + 64); k++) f(k); } };
  std::vector<std::thread> th; for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
  work(); for (auto& t : th) t.join();
}

// BGZF (SAM spec 4.1): a series of gzip members each carrying its compressed size in a "BC" extra subfield, so the members
// can be inflated independently.  Falls back to a plain sequential gzip read when a member lacks the subfield.
static inline std::vector<uint8_t> read_bgzf_all(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw MkErr("cannot open " + path);
  std::vector<uint8_t> comp; { fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); comp.resize((size_t)n);
    if (n && fread(comp.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f);
      throw MkErr("read error " + path); } }
  fclose(f);
  struct Blk { size_t off, clen; uint32_t isize; uint64_t uoff; };
  std::vector<Blk> blks; size_t o = 0; uint64_t u = 0; bool ok = true;
  while (o < comp.size()) {
    if (o + 18 > comp.size() || comp[o] != 31 || comp[o + 1] != 139 || !(comp[o + 3] & 4)) { ok = false; break; }
    const size_t xlen = comp[o + 10] | (comp[o + 11] << 8); size_t x = o + 12, xe = x + xlen; int bsize = -1;
    if (xe > comp.size()) { ok = false; break; }
    while (x + 4 <= xe) { const size_t sl = comp[x + 2] | (comp[x + 3] << 8);
      if (comp[x] == 'B' && comp[x + 1] == 'C' && sl == 2 && x + 6 <= xe) bsize = comp[x + 4] | (comp[x + 5] << 8);
      x += 4 + sl; }
    if (bsize < 0 || o + (size_t)bsize + 1 > comp.size() || (size_t)bsize + 1 < xlen + 20) { ok = false; break; }
    const size_t total = (size_t)bsize + 1; uint32_t isize; memcpy(&isize, &comp[o + total - 4], 4);
    blks.push_back({o + 12 + xlen, total - xlen - 20, isize, u}); u += isize; o += total;
  }
  if (!ok) return read_gz_all(path);
  std::vector<uint8_t> out((size_t)u); std::atomic<bool> bad{false};
  oracle_parallel_for(blks.size(), [&](size_t i) {
    const Blk& b = blks[i]; if (!b.isize) return;
    z_stream zs; memset(&zs, 0, sizeof(zs)); if (inflateInit2(&zs, -15) != Z_OK) { bad = true; return; }
    zs.next_in = comp.data() + b.off; zs.avail_in = (uInt)b.clen; zs.next_out = out.data() + b.uoff; zs.avail_out = b.isize;
    const int rc = inflate(&zs, Z_FINISH); if (rc != Z_STREAM_END || zs.total_out != b.isize) bad = true;
    inflateEnd(&zs);
  });
  if (bad) throw MkErr("corrupt BGZF block in " + path);
  return out;
}

static inline BamFile read_bam(const std::string& path) {
  std::vector<uint8_t> d = read_bgzf_all(path);
  BamFile bf;
  size_t o = 0;
  auto need = [&](size_t n) { if (o + n > d.size()) throw MkErr("truncated BAM " + path); };
  auto i32 = [&]() { need(4); int32_t v; memcpy(&v, &d[o], 4); o += 4; return v; };
  need(4);
  if (memcmp(&d[0], "BAM\1", 4) != 0) throw MkErr("not a BAM file: " + path);
  o = 4;
  int32_t l_text = i32(); need(l_text); o += l_text;
  int32_t n_ref = i32();
  for (int i = 0; i < n_ref; i++) {
    int32_t l_name = i32(); need(l_name);
    bf.ref_names.push_back(std::string((const char*)&d[o], l_name > 0 ? l_name - 1 : 0));
    o += l_name;
    bf.ref_lens.push_back((uint32_t)i32());
  }
  // record boundaries first (sequential, cheap), then the records themselves on all cores
  std::vector<std::pair<size_t, size_t>> spans;
  while (o + 4 <= d.size()) { int32_t bs = i32(); if (bs < 32) throw MkErr("corrupt BAM record"); need((size_t)bs);
    spans.push_back({o, o + (size_t)bs}); o += (size_t)bs; }
  bf.recs.resize(spans.size());
  static const char* NT16 = "=ACMGRSVTWYHKDBN";
  std::atomic<bool> bad{false};
  oracle_parallel_for(spans.size(), [&](size_t k) {
    size_t p = spans[k].first; const size_t e = spans[k].second;
    auto g32 = [&]() { int32_t v; memcpy(&v, &d[p], 4); p += 4; return v; };
    BamRecord& r = bf.recs[k];
    r.tid = g32(); r.pos = g32();
    uint8_t l_read_name = d[p]; p += 1; p += 1 /*mapq*/; p += 2 /*bin*/;
    uint16_t n_cigar; memcpy(&n_cigar, &d[p], 2); p += 2;
    memcpy(&r.flag, &d[p], 2); p += 2;
    r.l_seq = g32(); p += 12;  // next_refID, next_pos, tlen
    if (r.l_seq < 0 || p + l_read_name + 4 * (size_t)n_cigar + (size_t)(r.l_seq + 1) / 2 + (size_t)r.l_seq > e) { bad = true; return; }
    r.qname = std::string((const char*)&d[p], l_read_name > 0 ? l_read_name - 1 : 0); p += l_read_name;
    r.cigar.resize(n_cigar);
    if (n_cigar) memcpy(r.cigar.data(), &d[p], 4 * (size_t)n_cigar);
    p += 4 * (size_t)n_cigar;
    r.seq.resize(r.l_seq);
    for (int i = 0; i < r.l_seq; i++) { uint8_t b = d[p + i / 2]; r.seq[i] = NT16[(i & 1) ? (b & 15) : (b >> 4)]; }
    p += (size_t)(r.l_seq + 1) / 2;
    r.qual.assign(d.begin() + (std::ptrdiff_t)p, d.begin() + (std::ptrdiff_t)(p + (size_t)r.l_seq));
    p += (size_t)r.l_seq;
    r.aux.assign(d.begin() + (std::ptrdiff_t)p, d.begin() + (std::ptrdiff_t)e);
  });
  if (bad) throw MkErr("corrupt BAM record");
  return bf;
}

// aux lookup (bam_aux_get: first occurrence).  Returns pointer to the type byte.
static inline const uint8_t* aux_find(const std::vector<uint8_t>& aux, const char* tag, const uint8_t** end_out) {
  size_t o = 0, n = aux.size();
  while (o + 3 <= n) {
    const uint8_t* t = &aux[o];
    char ty = (char)aux[o + 2];
    size_t v = o + 3, len = 0;
    switch (ty) {
      case 'A': case 'c': case 'C': len = 1; break;
      case 's': case 'S': len = 2; break;
      case 'i': case 'I': case 'f': len = 4; break;
      case 'd': len = 8; break;
      case 'Z': case 'H': { size_t k = v; while (k < n && aux[k]) k++; len = k - v + 1; break; }
      case 'B': {
        if (v + 5 > n) return nullptr;
        char st = (char)aux[v]; uint32_t cnt; memcpy(&cnt, &aux[v + 1], 4);
        size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
        len = 5 + es * (size_t)cnt; break;
      }
      default: return nullptr;
    }
    if (v + len > n) return nullptr;
    if (t[0] == (uint8_t)tag[0] && t[1] == (uint8_t)tag[1]) { if (end_out) *end_out = &aux[0] + v + len; return t + 2; }
    o = v + len;
  }
  return nullptr;
}

// ------------------------------------------------------- MM / ML tag decoding
enum SkipMode { MODE_EXPLICIT = 0, MODE_IMPLICIT = 1, MODE_DEFAULT_IMPLICIT = 2 };  // mod_bam.rs:326-336

struct MmTagInfo {          // mod_bam.rs:873-879
  char fundamental_base;    // A C G T U N
  int mode;
  bool neg_strand;
  std::vector<ModCode> codes;
  std::vector<uint32_t> deltas;
};

// MmTagInfo::parse (mod_bam.rs:909-1000), parse_int_list 881-889
static inline MmTagInfo parse_one_mm(const std::string& s) {
  MmTagInfo t;
  size_t comma = s.find(',');
  std::string header = s.substr(0, comma);
  if (header.size() < 1) throw MkErr("invalid-MM-tag: no canonical base");
  char fb = header[0];
  if (!(fb == 'A' || fb == 'C' || fb == 'G' || fb == 'T' || fb == 'U' || fb == 'N'))
    throw MkErr("invalid-MM-tag: fundamental base");
  t.fundamental_base = fb;
  if (header.size() < 2) throw MkErr("invalid-MM-tag: no strand");
  if (header[1] == '+') t.neg_strand = false;
  else if (header[1] == '-') t.neg_strand = true;
  else throw MkErr("invalid-strand");
  size_t i = 2, offset = 2;
  bool seen_chebi = false, have_mode = false;
  t.mode = MODE_DEFAULT_IMPLICIT;
  if (i < header.size() && header[i] >= '0' && header[i] <= '9') {
    uint64_t v = 0;
    while (i < header.size() && header[i] >= '0' && header[i] <= '9') {
      v = v * 10 + (uint64_t)(header[i] - '0');
      if (v > 0xffffffffull) throw MkErr("invalid-MM-tag: chebi overflow");
      i++; offset++;
    }
    t.codes.push_back(code_chebi((uint32_t)v));
    seen_chebi = true;
  }
  for (; i < header.size(); i++) {
    char c = header[i];
    if (c == '?' || c == '.') { t.mode = (c == '?') ? MODE_EXPLICIT : MODE_IMPLICIT; have_mode = true; offset++; }
    else if (c >= '0' && c <= '9') throw MkErr("invalid-MM-tag: digit mod code");
    else { if (seen_chebi) throw MkErr("invalid-MM-tag: chebi+code"); t.codes.push_back(code_char(c)); offset++; }
  }
  (void)have_mode;
  if (offset + 1 <= s.size()) {
    // parse_int_list on s[offset+1..]: separated_list1(",", ws* digit1 ws*); trailing junk ignored
    const char* p = s.c_str() + offset + 1;
    bool first = true;
    for (;;) {
      const char* save = p;
      if (!first) { if (*p != ',') break; p++; }
      while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
      if (!(*p >= '0' && *p <= '9')) { if (first) throw MkErr("invalid-MM-tag: delta list"); p = save; break; }
      uint64_t v = 0;
      while (*p >= '0' && *p <= '9') { v = v * 10 + (uint64_t)(*p - '0'); if (v > 0xffffffffull) throw MkErr("invalid-MM-tag: delta overflow"); p++; }
      while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
      t.deltas.push_back((uint32_t)v);
      first = false;
    }
  }
  return t;
}

// MmTagInfo::parse_mm_tag (mod_bam.rs:900-907)
static inline std::vector<MmTagInfo> parse_mm_tag(const std::string& mm) {
  std::vector<MmTagInfo> out;
  size_t s = 0;
  while (s <= mm.size()) {
    size_t e = mm.find(';', s);
    if (e == std::string::npos) e = mm.size();
    if (e > s) out.push_back(parse_one_mm(mm.substr(s, e - s)));
    s = e + 1;
  }
  return out;
}

// BaseModProbs (mod_bam.rs:415-657)
struct BaseModProbs {
  FxProbMap probs;
  bool inferred = false;
  static constexpr float MAX_PROB = 1.01f;  // mod_bam.rs:26
  void add_base_mod_prob(ModCode c, float p) {  // 443-464
    if (inferred && p > 0.0f) throw MkErr("conflict-inferred-prob-greater-than-one");
    int i = probs.entry(c, 0.0f);
    if (probs.vals[i] + p > MAX_PROB) throw MkErr("conflict-explicit-prob-greater-than-one");
    probs.vals[i] = probs.vals[i] + p;
  }
  void add_inferred_canonical(const std::vector<ModCode>& codes) {  // 466-487
    if (!inferred) return;
    for (ModCode c : codes) {
      float prev;
      if (probs.insert(c, 0.0f, &prev) && prev > 0.0f) throw MkErr("conflict-inferred-prob-greater-than-one");
    }
  }
  float canonical_prob() const { return 1.0f - probs.sum(); }  // 507-509
  // argmax_base_mod_call value (489-505): max_by returns the LAST maximum
  float argmax_value() const {
    float can = canonical_prob();
    bool have = false; float best = 0.0f;
    probs.for_each([&](ModCode, float p) { if (!have || !(p < best)) { best = p; have = true; } });
    if (have && best > can) return best;
    return can;
  }
  void combine_checked(const BaseModProbs& o) {  // 629-656
    if (inferred != o.inferred) throw MkErr("conflict-explicit-and-inferred");
    o.probs.for_each([&](ModCode c, float p) { int i = probs.entry(c, 0.0f); probs.vals[i] = probs.vals[i] + p; });
    if (probs.sum() > MAX_PROB) throw MkErr("conflict-explicit-prob-greater-than-one");
  }
};

struct CollapseMethod { bool active = false; ModCode code = 0; };  // only ReDistribute is reachable from `pileup`

// BaseModProbs::into_collapsed, ReDistribute arm (mod_bam.rs:558-600)
static inline BaseModProbs collapse_redistribute(const BaseModProbs& in, ModCode x) {
  float marginal = 0.0f;
  in.probs.for_each([&](ModCode c, float p) { if (c == x) marginal = marginal + p; });
  std::vector<std::pair<ModCode, float>> others;
  in.probs.for_each([&](ModCode c, float p) { if (c != x) others.push_back({c, p}); });
  float n_other = (float)others.size() + 1.0f;
  float redistribute = marginal / n_other;
  BaseModProbs out;
  out.inferred = in.inferred;
  for (auto& kv : others) { float prev; out.probs.insert(kv.first, kv.second + redistribute, &prev); }
  return out;
}

struct SeqPosBaseModProbs {  // mod_bam.rs:1058-1065
  int skip_mode = MODE_EXPLICIT;
  std::map<size_t, BaseModProbs> pos;  // forward-sequence position -> probs (order-independent uses only)
};

static inline char comp_char(char c) {
  switch (c) {  // bio::alphabets::dna::revcomp
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    case 'N': return 'N';
    case 'R': return 'Y'; case 'Y': return 'R'; case 'S': return 'S'; case 'W': return 'W';
    case 'K': return 'M'; case 'M': return 'K'; case 'B': return 'V'; case 'V': return 'B';
    case 'D': return 'H'; case 'H': return 'D';
    default: return c;
  }
}
static inline std::string forward_sequence(const BamRecord& r) {  // util.rs:153-159
  if (!r.is_reverse()) return r.seq;
  std::string s(r.seq.rbegin(), r.seq.rend());
  for (auto& c : s) c = comp_char(c);
  return s;
}

static inline bool fb_matches(char fb, char nt) {  // mod_bam.rs:858-869
  switch (fb) {
    case 'A': return nt == 'A'; case 'C': return nt == 'C'; case 'G': return nt == 'G';
    case 'T': case 'U': return nt == 'T';
    default: return true;
  }
}

struct ModBaseInfo {  // mod_bam.rs:1472-1478
  std::map<int, SeqPosBaseModProbs> pos_strand, neg_strand;  // keyed by DnaBase
  bool is_empty() const {  // 1599-1604
    for (auto& kv : pos_strand) if (!kv.second.pos.empty()) return false;
    for (auto& kv : neg_strand) if (!kv.second.pos.empty()) return false;
    return true;
  }
};

// parse_raw_mod_tags + ModBaseInfo::new_from_record (mod_bam.rs:1388-1577)
static inline ModBaseInfo mod_base_info_from_record(const BamRecord& r) {
  // get_tag (util.rs:174-188): new-style wins, each tag chosen independently
  const uint8_t* mm = aux_find(r.aux, "MM", nullptr);
  if (!mm) mm = aux_find(r.aux, "Mm", nullptr);
  if (!mm) throw MkErr("MM-tag-missing");
  if ((char)mm[0] != 'Z') throw MkErr("invalid-MM-tag: wrong type");
  std::string raw_mm((const char*)mm + 1);
  const uint8_t* ml = aux_find(r.aux, "ML", nullptr);
  if (!ml) ml = aux_find(r.aux, "Ml", nullptr);
  if (!ml) throw MkErr("ML-tag-missing");
  if (!((char)ml[0] == 'B' && (char)ml[1] == 'C')) throw MkErr("invalid-ML-tag: wrong type");
  uint32_t ml_n; memcpy(&ml_n, ml + 2, 4);
  const uint8_t* quals = ml + 6;
  // MN (1416-1449)
  const uint8_t* mn = aux_find(r.aux, "MN", nullptr);
  if (mn) {
    int64_t v;
    switch ((char)mn[0]) {
      case 'c': v = (int8_t)mn[1]; break;
      case 'C': v = mn[1]; break;
      case 's': { int16_t x; memcpy(&x, mn + 1, 2); v = x; break; }
      case 'S': { uint16_t x; memcpy(&x, mn + 1, 2); v = x; break; }
      case 'i': { int32_t x; memcpy(&x, mn + 1, 4); v = x; break; }
      case 'I': { uint32_t x; memcpy(&x, mn + 1, 4); v = x; break; }
      default: throw MkErr("invalid-MN-tag: wrong type");
    }
    if ((uint64_t)v != (uint64_t)r.l_seq) throw MkErr("invalid-MN-tag: length mismatch");
  } else if (r.flag & (256 | 1024 | 2048)) {
    throw MkErr("non-primary-no-MN");
  }
  std::string fwd = forward_sequence(r);
  std::vector<MmTagInfo> tags = parse_mm_tag(raw_mm);

  ModBaseInfo info;
  std::map<char, std::vector<uint32_t>> converters;  // cumulative counts per fundamental base (667-684)
  size_t pointer = 0;
  for (const MmTagInfo& t : tags) {
    std::vector<uint32_t>* cum = nullptr;
    if (t.fundamental_base != 'N') {
      auto it = converters.find(t.fundamental_base);
      if (it == converters.end()) {
        std::vector<uint32_t> c(fwd.size());
        uint32_t count = 0;
        for (size_t i = 0; i < fwd.size(); i++) { if (fb_matches(t.fundamental_base, fwd[i])) count++; c[i] = count; }
        it = converters.emplace(t.fundamental_base, std::move(c)).first;
      }
      cum = &it->second;
    }
    // ---- get_base_mod_probs (1213-1295)
    std::vector<size_t> positions;
    if (t.fundamental_base == 'N') {  // to_positions, N arm (740-763)
      if (!t.deltas.empty()) {
        size_t last = t.deltas[0];
        positions.push_back(last);
        for (size_t i = 1; i < t.deltas.size(); i++) {
          size_t np = last + (size_t)t.deltas[i] + 1;
          if (np >= fwd.size()) throw MkErr("invalid-MM-tag: beyond end of seq");
          positions.push_back(np); last = np;
        }
      }
    } else {  // to_positions_specific (697-733)
      size_t finger = 0; uint64_t n_skips = 0;
      for (uint32_t d : t.deltas) {
        if (finger >= cum->size()) throw MkErr("invalid-MM-tag: beyond end of seq");
        while ((uint64_t)(*cum)[finger] <= (uint64_t)d + n_skips) {
          finger++;
          if (finger >= cum->size()) throw MkErr("invalid-MM-tag: beyond end of seq");
        }
        positions.push_back(finger);
        n_skips += (uint64_t)d + 1;
      }
    }
    size_t stride = t.codes.size();
    size_t end = pointer + t.deltas.size() * stride;
    if (end > (size_t)ml_n) throw MkErr("invalid-ML-tag: too short");
    std::map<int, SeqPosBaseModProbs> base_to_probs;
    for (size_t j = 0; j < positions.size(); j++) {
      size_t position = positions[j];
      if (position >= fwd.size()) throw MkErr("position beyond sequence (reference would panic)");
      int b = base_from_char(fwd[position]);
      if (b < 0) throw MkErr("invalid-DNA-RNA-base");
      auto it = base_to_probs.find(b);
      if (it == base_to_probs.end()) { it = base_to_probs.emplace(b, SeqPosBaseModProbs()).first; it->second.skip_mode = t.mode; }
      for (size_t i = 0; i < stride; i++) {
        float p = ((float)quals[pointer + j * stride + i] + 0.5f) / 256.0f;  // quals_to_probs 808-816
        auto pit = it->second.pos.find(position);
        if (pit != it->second.pos.end()) pit->second.add_base_mod_prob(t.codes[i], p);
        else { BaseModProbs bmp; bmp.probs.entry(t.codes[i], p); it->second.pos.emplace(position, bmp); }  // new_init 424-429
      }
    }
    if (t.mode != MODE_EXPLICIT && cum) {  // implicit fill (1265-1292); N has empty cumulative_counts
      uint32_t cum_sum = 0;
      for (size_t p = 0; p < cum->size(); p++) {
        uint32_t x = (*cum)[p];
        if (x > cum_sum) {
          int b = base_from_char(fwd[p]);
          if (b < 0) throw MkErr("invalid-DNA-RNA-base");
          auto it = base_to_probs.find(b);
          if (it == base_to_probs.end()) { it = base_to_probs.emplace(b, SeqPosBaseModProbs()).first; it->second.skip_mode = t.mode; }
          auto pit = it->second.pos.find(p);
          if (pit != it->second.pos.end()) pit->second.add_inferred_canonical(t.codes);
          else {
            BaseModProbs bmp; bmp.inferred = true;  // new_inferred_canonical 431-440
            for (ModCode c : t.codes) { float prev; bmp.probs.insert(c, 0.0f, &prev); }
            it->second.pos.emplace(p, bmp);
          }
        }
        cum_sum = x;
      }
    }
    // ---- merge into per-strand aggregate (1522-1532, combine_positions_to_probs 1037-1054)
    auto& dest = t.neg_strand ? info.neg_strand : info.pos_strand;
    for (auto& kv : base_to_probs) {
      auto ait = dest.find(kv.first);
      if (ait == dest.end()) { ait = dest.emplace(kv.first, SeqPosBaseModProbs()).first; ait->second.skip_mode = t.mode; }
      SeqPosBaseModProbs& agg = ait->second;
      if (agg.skip_mode != kv.second.skip_mode) agg.skip_mode = MODE_IMPLICIT;
      for (auto& pp : kv.second.pos) {
        auto pit = agg.pos.find(pp.first);
        if (pit != agg.pos.end()) pit->second.combine_checked(pp.second);
        else agg.pos.emplace(pp.first, pp.second);
      }
    }
    pointer += t.deltas.size() * stride;
  }
  return info;
}

struct EdgeFilter {  // mod_bam.rs:1634-1672
  bool active = false;
  size_t start = 0, end = 0;
  bool inverted = false;
  bool read_can_be_trimmed(size_t len) const { return !(len <= start || len <= end); }
  bool keep_position(size_t p, size_t len) const {
    if (inverted) return p < start || p >= len - end;
    return p >= start && p < len - end;
  }
};

// ------------------------------------------------------------ threshold caller
struct BaseModCall {  // mod_bam.rs:370-375
  enum Kind { FILTERED, CANONICAL, MODIFIED } kind = FILTERED;
  float p = 0.0f;
  ModCode code = 0;
};

// BaseModProbs::argmax_base_mod_call (mod_bam.rs:489-505): max_by keeps the LAST maximum; a tie with the canonical probability is canonical
static inline BaseModCall argmax_call(const BaseModProbs& bmp) {
  const float can = bmp.canonical_prob();
  bool have = false; float best = 0.0f; ModCode bc = 0;
  bmp.probs.for_each([&](ModCode c, float p) { if (!have || !(p < best)) { best = p; bc = c; have = true; } });
  BaseModCall m;
  if (have && best > can) { m.kind = BaseModCall::MODIFIED; m.p = best; m.code = bc; } else { m.kind = BaseModCall::CANONICAL; m.p = can; }
  return m;
}

struct ThresholdCaller {  // threshold_mod_caller.rs:8-13
  std::map<int, float> per_base;
  std::map<ModCode, float> per_mod;
  float default_threshold = 0.0f;
  // call (threshold_mod_caller.rs:28-63)
  BaseModCall call(int canonical_base, const BaseModProbs& bmp) const {
    std::vector<BaseModCall> cands;
    bmp.probs.for_each([&](ModCode c, float p) {
      float thr;
      auto a = per_mod.find(c);
      if (a != per_mod.end()) thr = a->second;
      else {
        auto b = per_mod.find(code_char(base_char(canonical_base)));
        if (b != per_mod.end()) thr = b->second;
        else { auto d = per_base.find(canonical_base); thr = d != per_base.end() ? d->second : default_threshold; }
      }
      if (p >= thr) { BaseModCall m; m.kind = BaseModCall::MODIFIED; m.p = p; m.code = c; cands.push_back(m); }
    });
    auto d = per_base.find(canonical_base);
    float can_thr = d != per_base.end() ? d->second : default_threshold;
    float cp = bmp.canonical_prob();
    if (cp >= can_thr) { BaseModCall m; m.kind = BaseModCall::CANONICAL; m.p = cp; cands.push_back(m); }
    if (cands.empty()) return BaseModCall();
    size_t best = 0;  // Iterator::max -> last maximal element (BaseModCall ord = probability, mod_bam.rs:379-397)
    for (size_t i = 1; i < cands.size(); i++) if (!(cands[i].p < cands[best].p)) best = i;
    return cands[best];
  }
};

// percentile_linear_interp (thresholds.rs:17-38); xs sorted ascending
static inline float percentile_linear_interp(const std::vector<float>& xs, float q) {
  if (xs.size() < 2) throw MkErr("not enough datapoints, got " + std::to_string(xs.size()));
  if (q > 1.0f) throw MkErr("invalid quantile");
  if (q == 1.0f) return xs[xs.size() - 1];
  float l = (float)(xs.size() - 1);
  float lq = l * q;
  float left = floorf(lq);
  size_t right = (size_t)ceilf(lq);
  float g = lq - truncf(lq);  // f32::fract
  float y0 = xs[(size_t)left];
  float y1 = xs[right];
  float a = y0 * (1.0f - g);
  float b = y1 * g;
  return a + b;
}

// rand 0.8.5 `StdRng` as RecordSampler uses it (src/reads_sampler/record_sampler.rs:29-38 `StdRng::seed_from_u64`, 80-86 `gen_bool`).
// The crate is a dependency (Cargo.toml:42 `rand = "0.8.5"`, which pulls rand_chacha 0.3.1 / rand_core 0.6.4), not in /root/reference:
// restated from its published algorithm —
//   * rand_core `SeedableRng::seed_from_u64`: the 32 seed bytes are eight PCG32 (XSH-RR) outputs of an LCG over the u64, little endian;
//   * `StdRng` = `ChaCha12Rng`: djb ChaCha, 12 rounds, 256-bit key = the seed, 64-bit block counter (words 12-13) from 0, 64-bit stream
//     id (words 14-15) 0; the block buffer hands out the keystream's u32 words in order, a u64 = (next word) | (word after) << 32;
//   * `Rng::gen_bool(p)` = `Bernoulli::new(p).sample`: p == 1 is always true without a draw, otherwise one u64 < (p * 2^64) as u64.
// Pinned: the block function against the published zero-key keystreams at 20 and 12 rounds (tests/test_oracle_unit_kats.py).  Unpinned:
// the seed expansion and the Bernoulli draw (no reference fixture depends on a seeded sample).
struct ChaChaBlocks {
  uint32_t key[8] = {0}; uint64_t counter = 0; int rounds = 12;
  static inline uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
  void block(uint32_t out[16]) {
    uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; i++) st[4 + i] = key[i];
    st[12] = (uint32_t)counter; st[13] = (uint32_t)(counter >> 32); st[14] = 0; st[15] = 0;
    uint32_t x[16]; for (int i = 0; i < 16; i++) x[i] = st[i];
    auto qr = [&](int a, int b, int c, int d) {
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7); };
    for (int r = 0; r < rounds; r += 2) { qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
                                          qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14); }
    for (int i = 0; i < 16; i++) out[i] = x[i] + st[i];
    counter++;
  }
};
struct StdRng {
  ChaChaBlocks cc; uint32_t buf[16]; int idx = 16;
  static StdRng seed_from_u64(uint64_t state) {
    StdRng r;
    for (int i = 0; i < 8; i++) {
      state = state * 6364136223846793005ull + 11634580027462260723ull;
      const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
      r.cc.key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));   // (the bytes are stored little endian and read back as LE words)
    }
    return r;
  }
  uint32_t next_u32() { if (idx == 16) { cc.block(buf); idx = 0; } return buf[idx++]; }
  uint64_t next_u64() { const uint64_t lo = next_u32(); return lo | ((uint64_t)next_u32() << 32); }
  bool gen_bool(double p) {
    if (!(p >= 0.0 && p <= 1.0)) throw MkErr("Bernoulli: p outside [0, 1]");
    if (p == 1.0) return true;
    const uint64_t p_int = (uint64_t)(p * 18446744073709551616.0);
    return next_u64() < p_int;
  }
};

}  // namespace mko
