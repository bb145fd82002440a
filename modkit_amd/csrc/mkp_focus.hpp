// Host scheduling for libmkpileup: the reference's interval grid and focus positions, flattened
// into one byte per reference position (+ a small motif-id combo table) for the device.
//   ReferenceIntervalsFeeder::next_batch      src/interval_chunks.rs:563-643
//   FocusPositions::{new_motif, new_motif_combine_strands, new_regions}   :62-349
//   MotifLocationsLookup::get_motif_positions src/fasta.rs:92-228
//   find_motif_hits / RegexMotif              src/find_motifs/motif_bed.rs:21-337
//   StrandedPositionFilter                    src/position_filter.rs:27-347
// The grid has to be the reference's own (contig/region start + k*interval_size, ends extended in
// combine-strands mode): with --cpg and no --combine-strands a CpG straddling an interval boundary
// is lost by the reference, and bit-exact output has to lose it too.
#pragma once
#include "mkp_bam.hpp"
#include "mkp_device.h"

namespace mkp {

enum Rule : uint8_t { R_POS = 1, R_NEG = 2, R_BOTH = 3 };
static inline Rule rule_combine(Rule a, Rule b) { return a == b ? a : R_BOTH; }   // StrandRule::combine (util.rs:343-349)
static inline Rule rule_absorb(Rule a, bool neg) { return (a == (neg ? R_NEG : R_POS)) ? a : R_BOTH; }  // StrandRule::absorb (333-341)

struct Span { uint64_t s, e; };
static inline void merge_spans(std::vector<Span>& v) {  // rust-lapper merge_overlaps: touching intervals merge
  std::sort(v.begin(), v.end(), [](const Span& a, const Span& b) { return a.s != b.s ? a.s < b.s : a.e < b.e; });
  std::vector<Span> o;
  for (auto& x : v) { if (o.empty() || o.back().e < x.s) o.push_back(x); else if (o.back().e < x.e) o.back().e = x.e; }
  v.swap(o);
}
static inline bool spans_hit(const std::vector<Span>& v, uint64_t s, uint64_t e) {  // sorted, disjoint after merge
  size_t lo = 0, hi = v.size();
  while (lo < hi) { size_t m = (lo + hi) / 2; if (v[m].e <= s) lo = m + 1; else hi = m; }
  return lo < v.size() && v[lo].s < e;
}

struct BedFilter {
  std::map<uint32_t, std::vector<Span>> pos, neg;
  bool contains(uint32_t tid, uint64_t p, bool neg_strand) const { auto& m = neg_strand ? neg : pos; auto it = m.find(tid);
    return it != m.end() && spans_hit(it->second, p, p + 1); }
  bool overlaps(uint32_t tid, uint64_t s, uint64_t e) const { auto a = pos.find(tid); if (a != pos.end() && spans_hit(a->second, s, e)) return true;
    auto b = neg.find(tid); return b != neg.end() && spans_hit(b->second, s, e); }
  bool has_chrom(int64_t tid) const { return tid >= 0 && (pos.count((uint32_t)tid) || neg.count((uint32_t)tid)); }
  static BedFilter load(const std::string& path, const std::map<std::string, uint32_t>& c2t) {
    std::ifstream in(path); if (!in) throw Error(MKP_E_IO, "cannot open BED " + path);
    BedFilter bf; std::string line; std::map<std::string, bool> unknown;
    while (std::getline(in, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (line.empty()) continue;
      std::istringstream ss(line); std::vector<std::string> f; std::string w; while (ss >> w) f.push_back(w);
      if (f.size() < 3 || unknown.count(f[0])) continue;
      auto num = [](const std::string& s, uint64_t* v) { if (s.empty()) return false; *v = 0; for (char c : s) { if (c < '0' || c > '9') return false;
          *v = *v * 10 + (uint64_t)(c - '0'); } return true; };
      uint64_t s, e; if (!num(f[1], &s) || !num(f[2], &e)) continue;
      bool p, n;
      if (f.size() == 3) p = n = true;
      else if (f.size() >= 6) { if (f[5] == "+") { p = true; n = false; } else if (f[5] == "-") { p = false; n = true;
        } else if (f[5] == ".") p = n = true; else continue; }
      else continue;
      auto it = c2t.find(f[0]); if (it == c2t.end()) { unknown[f[0]] = true; continue; }
      if (p) bf.pos[it->second].push_back({s, e});
      if (n) bf.neg[it->second].push_back({s, e});
    }
    if (bf.pos.empty() && bf.neg.empty()) throw Error(MKP_E_INVALID, "zero valid positions parsed from BED file");
    for (auto& kv : bf.pos) merge_spans(kv.second);
    for (auto& kv : bf.neg) merge_spans(kv.second);
    return bf;
  }
};

struct Contig { uint32_t tid, start, length; std::string name; uint32_t end() const { return start + length; } };

// StrandedPositionFilter::optimize_reference_records (position_filter.rs:103-210)
static inline std::vector<Contig> bed_contigs(const BedFilter& bf, const std::vector<Contig>& recs, uint32_t interval_size) {
  std::map<uint32_t, Contig> lut; for (auto& r : recs) lut[r.tid] = r;
  std::map<uint32_t, bool> tids; for (auto& kv : bf.pos) tids[kv.first] = true; for (auto& kv : bf.neg) tids[kv.first] = true;
  std::vector<Contig> out;
  for (auto& t : tids) {
    auto li = lut.find(t.first); if (li == lut.end()) continue;
    std::vector<Span> v; auto a = bf.pos.find(t.first); if (a != bf.pos.end()) v = a->second;
    auto b = bf.neg.find(t.first); if (b != bf.neg.end()) v.insert(v.end(), b->second.begin(), b->second.end());
    merge_spans(v); if (v.empty()) continue;
    Span cur = v[0]; std::vector<Span> agg;
    for (size_t i = 1; i < v.size(); i++) { if (cur.e - cur.s > interval_size) { agg.push_back(cur); cur = v[i]; continue; } cur.e = v[i].e; }
    agg.push_back(cur);
    for (auto& s : agg) out.push_back({t.first, (uint32_t)s.s, (uint32_t)(s.e - s.s), li->second.name});
  }
  return out;
}

struct Motif {
  std::string raw; size_t fwd_off = 0, rev_off = 0; bool palindrome = false;
  std::vector<uint8_t> fwd, rev;  // per position: allowed-base bitmask (A1 C2 G4 T8; 'U' never matches an upper-cased DNA reference)
  size_t len() const { return raw.size(); }
  std::string label() const { return raw + "," + std::to_string(fwd_off); }
  static uint8_t iupac(char c) {
    switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; case 'U': return 16;
      case 'M': return 3; case 'R': return 5; case 'W': return 9; case 'S': return 6; case 'Y': return 10; case 'K': return 12;
      case 'V': return 7; case 'H': return 11; case 'D': return 13; case 'B': return 14; case 'X': case 'N': return 15;
      default: throw Error(MKP_E_INVALID, std::string("Invalid IUPAC code: ") + c); }
  }
  static Motif parse(const std::string& raw, size_t off) {  // RegexMotif::parse_string (motif_bed.rs:197-223)
    Motif m; m.raw = raw;
    if (raw.size() == 1 && std::string("ACGT").find(raw[0]) == std::string::npos) throw Error(MKP_E_INVALID,
        "degenerate bases are not supported as single base motifs");
    if (raw.size() < off + 1) throw Error(MKP_E_INVALID, "motif not long enough for offset");
    if (raw.size() > MKP_HALO) throw Error(MKP_E_UNSUPPORTED, "motifs longer than 16 bases");
    for (char c : raw) m.fwd.push_back(iupac(c));
    for (size_t i = raw.size(); i-- > 0;) { uint8_t f = m.fwd[i], r = 0; if (f & 1) r |= 8; if (f & 2) r |= 4; if (f & 4) r |= 2; if (f & 8) r |= 1;
      if (f & 16) r |= 1;
      m.rev.push_back(r); }
    m.fwd_off = off; m.rev_off = raw.size() - (off + 1);
    // palindrome iff the two regex *texts* are equal: classes keep their written letter order, and a
    // multi-letter class reversed+complemented is equal text only when it maps onto itself that way
    auto text = [](const std::vector<uint8_t>& cls, bool rc, const std::string& raw_) {
      std::string s; size_t n = cls.size();
      for (size_t i = 0; i < n; i++) {
        char c = rc ? raw_[n - 1 - i] : raw_[i];
        std::string letters;
        switch (c) { case 'M': letters = "AC"; break; case 'R': letters = "AG"; break; case 'W': letters = "AT"; break; case 'S': letters = "CG";
          break; case 'Y': letters = "CT"; break; case 'K': letters = "GT"; break;
          case 'V': letters = "ACG"; break; case 'H': letters = "ACT"; break; case 'D': letters = "AGT"; break; case 'B': letters = "CGT"; break;
            case 'X': case 'N': letters = "ACGT"; break; default: letters = std::string(1, c); }
        if (rc) { std::string t(letters.rbegin(), letters.rend());
          for (auto& ch : t) ch = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch == 'U' ? 'A' : ch;
          letters = t; }
        if (letters.size() == 1) s += letters; else s += "[" + letters + "]";
      }
      return s;
    };
    m.palindrome = text(m.fwd, false, raw) == text(m.fwd, true, raw);
    return m;
  }
  // MotifInfo::negative_strand_position
  bool neg_delta(int* d) const { if (!palindrome) return false; *d = (int)rev_off - (int)fwd_off; return true; }
};

// base -> bit (A1 C2 G4 T8), by table: [0] upper case only (--mask-reference: soft-masked bases never match), [1] either case (the
// reference upper-cases the sequence first, fasta.rs; reading through this table saves a copy of every contig)
struct BaseBits { uint8_t t[2][256]; BaseBits() { memset(t, 0, sizeof(t)); const char* b = "ACGT"; for (int i = 0; i < 4; i++) {
      t[0][(uint8_t)b[i]] = t[1][(uint8_t)b[i]] = (uint8_t)(1 << i); t[1][(uint8_t)(b[i] | 0x20)] = (uint8_t)(1 << i); } } };
static inline const uint8_t* base_bits(bool any_case) { static const BaseBits B; return B.t[any_case ? 1 : 0]; }
static inline uint8_t base_bit(char c) { return base_bits(false)[(uint8_t)c]; }

// find_motif_hits (motif_bed.rs:288-337) folded straight into a position -> rule map relative to `off`
static inline void motif_hits(const char* seq, size_t n, const Motif& m, uint64_t off, uint32_t tid, const BedFilter* bf, std::map<uint32_t,
    Rule>* out, bool any_case = false) {
  const uint8_t* bb = base_bits(any_case);
  auto add = [&](size_t p, bool neg) { uint64_t g = off + p; if (bf && !bf->contains(tid, g, neg)) return; auto it = out->find((uint32_t)g);
    if (it != out->end()) it->second = rule_absorb(it->second, neg);
    else (*out)[(uint32_t)g] = neg ? R_NEG : R_POS;
    };
  size_t L = m.len();
  auto match = [&](const std::vector<uint8_t>& cls, size_t i) { for (size_t j = 0; j < L; j++) if (!(cls[j] & bb[(uint8_t)seq[i + j]])) return false;
    return true; };
  if (m.palindrome) { for (size_t i = 0; i + L <= n; i++) if (match(m.fwd, i)) { add(i + m.fwd_off, false); add(i + m.rev_off, true); } }
  else if (L == 1) { const uint8_t fw = bb[(uint8_t)m.raw[0]], rv = (uint8_t)(fw == 1 ? 8 : fw == 2 ? 4 : fw == 4 ? 2 : 1);
    for (size_t i = 0; i < n; i++) { const uint8_t x = bb[(uint8_t)seq[i]];
      if (x == fw) add(i, false);
      else if (x == rv) add(i, true);
    } }
  else { for (size_t i = 0; i + L <= n; i++) { if (match(m.fwd, i)) add(i + m.fwd_off, false); if (match(m.rev, i)) add(i + m.rev_off, true); } }
}

struct Interval { uint32_t tid, start, end; };

// Walks the reference's interval grid and writes the dense focus bytes for [win_start, win_end) of `tid`.
class FocusBuilder {
 public:
  const Fasta* fasta = nullptr; bool mask = false; std::vector<Motif> motifs; const BedFilter* bed = nullptr; bool combine = false;
  std::vector<mkp_motif_combo> combos;  // [0] = none
  bool fast_single = true;   // one motif: fill_single instead of motif_hits + fill_motif (tests switch it off to compare the two)
  FocusBuilder() { mkp_motif_combo z; memset(&z, 0, sizeof(z)); combos.push_back(z); }
  bool has_focus() const { return !motifs.empty() || bed != nullptr; }

  // Interval list of one contig record, in feeder order; fills `focus` (size end-start of the record,
  // indexed from rec.start) when non-null.
  std::vector<Interval> walk(const Contig& rec, uint32_t interval_size, std::vector<uint8_t>* focus) {
    std::vector<Interval> ivs;
    if (interval_size == 0) throw Error(MKP_E_INVALID, "interval size must be positive");
    if (focus) focus->assign(rec.length, 0);
    const FastaSeq* seq = nullptr;
    // without --mask-reference the reference upper-cases the contig: read it through the either-case table instead of copying it
    const bool any_case = !mask;
    if (!motifs.empty()) {
      seq = fasta->get(rec.name);
      if (!seq) throw Error(MKP_E_IO, "contig " + rec.name + " missing from reference FASTA");
      if (rec.end() > seq->size()) throw Error(MKP_E_IO, "contig " + rec.name + " shorter in FASTA than in the BAM header");
    }
    size_t longest = 0; for (auto& m : motifs) longest = std::max(longest, m.len());
    if (!motifs.empty()) {
      // The grid.  Without strand combining it is fixed (start + k * interval_size).  With it the end of an interval moves past a
      // motif hit that straddles it (get_motif_positions_combine_strands, fasta.rs:92-188) and the next interval starts there, so
      // the ends are found one after the other — but each needs only the hits around the nominal end: a merged run of hits that
      // covers position e-1 is chained hit by hit, and every hit that can chain to e-1 or beyond starts after e - 2 * longest.
      uint32_t pos = rec.start;
      while (pos < rec.end()) {
        uint32_t end = (uint32_t)std::min<uint64_t>((uint64_t)pos + interval_size, rec.end());
        if (combine) {
          uint64_t ref_end = rec.end(), buffer = longest * 5, e = end, end_w = std::min<uint64_t>((uint64_t)end + buffer, ref_end);
          std::vector<std::map<uint32_t, Rule>> locs(motifs.size());
          for (;;) {
            if (end_w > seq->size()) throw Error(MKP_E_UNSUPPORTED,
                "motif run reaches past the contig end while extending an interval (the reference never terminates here)");
            const uint64_t from = std::max<uint64_t>(pos, e > 2 * longest + 2 ? e - 2 * longest - 2 : 0);
            for (auto& l : locs) l.clear();
            for (size_t i = 0; i < motifs.size(); i++) motif_hits(seq->data() + from, (size_t)(end_w - from), motifs[i], (uint32_t)from, rec.tid, bed,
                &locs[i], any_case);
            std::vector<Span> sp;
            for (size_t i = 0; i < motifs.size(); i++) {
              uint64_t adj = motifs[i].len() >= motifs[i].fwd_off ? motifs[i].len() - motifs[i].fwd_off : motifs[i].len();
              for (auto& kv : locs[i]) sp.push_back({kv.first, kv.first + adj});
              }
            merge_spans(sp);
            uint64_t search_end = e, qs = e ? e - 1 : 0;
            for (auto& s2 : sp) if (s2.s < e && s2.e > qs) { search_end = s2.e; break; }
            uint64_t too_close = end_w >= longest ? end_w - longest : 0;
            if (search_end < too_close || end_w >= ref_end) { end = (uint32_t)std::min<uint64_t>(search_end, rec.end()); break; }
            e = end_w; end_w += buffer;
          }
        }
        ivs.push_back({rec.tid, pos, end});
        if (end <= pos) throw Error(MKP_E_INVALID, "interval size must be positive");
        pos = end;
      }
      // The focus bytes: every interval's motif hits depend on its own slice of the reference only (with strand combining the
      // slice runs `longest` past the interval end, so that a hit that starts inside it is seen whole): intervals are filled by all
      // host cores, each block of intervals with its own motif-id combo table; the tables are then interned into the shared one in
      // interval order (the ids a sequential walk would give) and a block whose local ids differ has its bytes renumbered.
      if (focus) {
        const size_t per_block = std::max<size_t>(1, (4u << 20) / interval_size), n_blocks = (ivs.size() + per_block - 1) / per_block;
        std::vector<FocusBuilder> local(n_blocks);
        std::vector<std::unique_ptr<Error>> errs(n_blocks);
        const bool comb = combine;
        parallel_ranges(0, n_blocks, 1, [&](uint64_t b0, uint64_t b1) {
          for (uint64_t b = b0; b < b1; b++) {
            FocusBuilder& L = local[b]; L.fasta = fasta; L.mask = mask; L.motifs = motifs; L.bed = bed; L.combine = comb;
            try {
              std::vector<uint8_t> scratch;
              for (size_t k = b * per_block; k < std::min(ivs.size(), (b + 1) * per_block); k++) {
                const uint64_t slice_end = comb ? std::min<uint64_t>((uint64_t)ivs[k].end + longest, rec.end()) : ivs[k].end;
                if (motifs.size() == 1 && fast_single) {
                  L.fill_single(seq->data(), rec, ivs[k].start, ivs[k].end, slice_end, focus, scratch, any_case); continue; }
                std::vector<std::map<uint32_t, Rule>> locs(motifs.size());
                for (size_t i = 0; i < motifs.size(); i++) motif_hits(seq->data() + ivs[k].start, (size_t)(slice_end - ivs[k].start), motifs[i],
                    ivs[k].start, rec.tid, bed, &locs[i], any_case);
                L.fill_motif(locs, rec, ivs[k].start, ivs[k].end, focus);
              }
            } catch (const Error& e) { errs[b].reset(new Error(e)); }
          }
        });
        for (size_t b = 0; b < n_blocks; b++) {
          if (errs[b]) throw *errs[b];
          std::vector<uint8_t> remap(local[b].combos.size(), 0); bool identity = true;
          for (size_t i = 1; i < local[b].combos.size(); i++) { remap[i] = combo_id(local[b].combos[i]); identity = identity && remap[i] == i; }
          if (identity) continue;
          const size_t k0 = b * per_block, k1 = std::min(ivs.size(), (b + 1) * per_block);
          for (uint64_t p = ivs[k0].start; p < ivs[k1 - 1].end; p++) { uint8_t& f = (*focus)[p - rec.start];
            if (f >> 2) f = (uint8_t)((f & 3u) | (remap[f >> 2] << 2));
            }
        }
      }
      return ivs;
    }
    for (uint32_t pos = rec.start; pos < rec.end();) {   // no motifs: the fixed grid; BED focus bytes per interval
      const uint32_t end = (uint32_t)std::min<uint64_t>((uint64_t)pos + interval_size, rec.end());
      if (bed && focus) {
        auto mark = [&](const std::map<uint32_t, std::vector<Span>>& m, uint8_t bit) { auto it = m.find(rec.tid); if (it == m.end()) return;
          for (auto& s2 : it->second) { uint64_t a = std::max<uint64_t>(s2.s, pos), b = std::min<uint64_t>(s2.e, end);
            for (uint64_t p = a; p < b; p++) (*focus)[p - rec.start] |= bit;
          } };
        mark(bed->pos, 1); mark(bed->neg, 2);
      }
      ivs.push_back({rec.tid, pos, end});
      pos = end;
    }
    return ivs;
  }

  template <class F> static void parallel_ranges(uint64_t lo, uint64_t hi, uint64_t grain, F f) {   // f(a, b) over [lo, hi) on the host pool
    const uint64_t n = hi > lo ? hi - lo : 0, g = std::max<uint64_t>(grain, 1), pieces = (n + g - 1) / g;
    HostPool::get().parallel((size_t)pieces, [&](size_t i) { f(lo + i * g, std::min(hi, lo + (i + 1) * g)); });
  }

  // (public for tests/test_host_focus_kats.py, which walks a contig interval by interval with these and compares with walk())
  uint8_t combo_id(const mkp_motif_combo& c) {
    if (c.n_pos == 0 && c.n_neg == 0) return 0;
    for (size_t i = 1; i < combos.size(); i++) if (memcmp(&combos[i], &c, sizeof(c)) == 0) return (uint8_t)i;
    if (combos.size() >= 64) throw Error(MKP_E_UNSUPPORTED, "more than 63 distinct motif-id combinations");
    combos.push_back(c); return (uint8_t)(combos.size() - 1);
  }
  // One motif (the usual run: --cpg, one --motif): the focus bytes of interval [start, end) from the slice [start, slice_end) of `seq`
  // (whole contig, offset 0) without the per-hit maps below — a rule-bit array over the slice, then one ascending pass.  Same hits as
  // motif_hits (a match must lie inside the slice: the boundary-CpG loss of the reference's per-interval search, SURVEY hazard 3), same
  // combos in the same first-appearance order as fill_motif (positions ascending), hence the same bytes; walk() falls back to the maps
  // for several motifs.  A 3 Gb genome's --cpg focus bytes took 13 s of std::map work on 16 cores.
  void fill_single(const char* seq, const Contig& rec, uint32_t start, uint32_t end, uint64_t slice_end, std::vector<uint8_t>* focus,
      std::vector<uint8_t>& rule, bool any_case) {
    const Motif& m = motifs[0]; const size_t L = m.len(), n = (size_t)(slice_end - start); const char* s = seq + start;
      const uint8_t* bb = base_bits(any_case);
    rule.assign(n, 0);
    auto add = [&](size_t p, bool neg) { if (bed && !bed->contains(rec.tid, (uint64_t)start + p, neg)) return;
      rule[p] |= neg ? (uint8_t)R_NEG : (uint8_t)R_POS; };
    const uint8_t* f0 = m.fwd.data(); const uint8_t* r0 = m.rev.data();
    auto match = [&](const uint8_t* cls, size_t i) { for (size_t j = 0; j < L; j++) if (!(cls[j] & bb[(uint8_t)s[i + j]])) return false; return true;
      };
    if (m.palindrome) { for (size_t i = 0; i + L <= n; i++) if (match(f0, i)) { add(i + m.fwd_off, false); add(i + m.rev_off, true); } }
    else if (L == 1) { const uint8_t fw = bb[(uint8_t)m.raw[0]], rv = (uint8_t)(fw == 1 ? 8 : fw == 2 ? 4 : fw == 4 ? 2 : 1);
      for (size_t i = 0; i < n; i++) { const uint8_t x = bb[(uint8_t)s[i]];
        if (x == fw) add(i, false);
        else if (x == rv) add(i, true);
      } }
    else { for (size_t i = 0; i + L <= n; i++) { if (match(f0, i)) add(i + m.fwd_off, false); if (match(r0, i)) add(i + m.rev_off, true); } }
    int d = 0; const bool has_d = m.neg_delta(&d);
    const size_t upto = std::min<size_t>(n, (size_t)(end - start));
    for (size_t p = 0; p < upto; p++) {
      const uint8_t r = rule[p]; if (!r) continue;
      mkp_motif_combo c; memset(&c, 0, sizeof(c)); for (auto& x : c.pos_delta) x = -128;
      if (combine) {   // FocusPositions::new_motif_combine_strands: a '+' (or both-strand) hit carries the offset of its '-' mate, see fill_motif
        if (r == R_POS || r == R_BOTH) {
          int8_t dd = -128;
          if (has_d) { const int64_t q = (int64_t)start + (int64_t)p + d;
            dd = (q < 0) ? (int8_t)-128 : (q >= (int64_t)start && q < (int64_t)end) ? (int8_t)d : (int8_t)-127; }
          c.pos_ids[0] = 0; c.pos_delta[0] = dd; c.n_pos = 1;
        } else { c.neg_ids[0] = 0; c.n_neg = 1; }
      } else {
        if (r & R_POS) { c.pos_ids[0] = 0; c.n_pos = 1; }
        if (r & R_NEG) { c.neg_ids[0] = 0; c.n_neg = 1; }
      }
      (*focus)[(size_t)start + p - rec.start] = (uint8_t)(r | (combo_id(c) << 2));
    }
  }
  void fill_motif(const std::vector<std::map<uint32_t, Rule>>& locs, const Contig& rec, uint32_t start, uint32_t end, std::vector<uint8_t>* focus) {
    if (motifs.size() > MKP_MAX_MOTIF_IDS) throw Error(MKP_E_UNSUPPORTED, "more than 4 motifs");
    std::map<uint32_t, Rule> positions; std::map<uint32_t, mkp_motif_combo> ids;
    auto cb = [&](uint32_t p) -> mkp_motif_combo& { auto it = ids.find(p); if (it == ids.end()) { mkp_motif_combo z; memset(&z, 0, sizeof(z));
        for (auto& d : z.pos_delta) d = -128;
        it = ids.emplace(p, z).first; } return it->second; };
    auto in = [&](uint32_t p) { return p >= start && p < end; };
    auto set_pos = [&](mkp_motif_combo& c, std::initializer_list<size_t> v) { c.n_pos = 0; for (size_t x : v) c.pos_ids[c.n_pos++] = (uint8_t)x; };
    auto set_neg = [&](mkp_motif_combo& c, std::initializer_list<size_t> v) { c.n_neg = 0; for (size_t x : v) c.neg_ids[c.n_neg++] = (uint8_t)x; };
    bool all_single = true; for (auto& m : motifs) if (m.len() != 1) all_single = false;
    if (combine) {  // FocusPositions::new_motif_combine_strands (interval_chunks.rs:250-297)
      for (size_t id = 0; id < motifs.size(); id++) for (auto& kv : locs[id]) {
        if (!in(kv.first)) continue;
        auto it = positions.find(kv.first); if (it != positions.end()) it->second = rule_combine(it->second, kv.second);
          else positions[kv.first] = kv.second;
        mkp_motif_combo& c = cb(kv.first);
        if (kv.second == R_POS || kv.second == R_BOTH) {
          // pos_delta: -128 = negative_strand_position() is None (motif skipped); -127 = the mate position lies outside this
          // interval, so the reference finds no '-' rows for it (position_feature_counts is per interval, pileup/mod.rs:496-497)
          int d; int8_t dd = -128;
          if (motifs[id].neg_delta(&d)) { int64_t q = (int64_t)kv.first + d;
            dd = (q < 0) ? (int8_t)-128 : (q >= (int64_t)start && q < (int64_t)end) ? (int8_t)d : (int8_t)-127; }
          c.pos_ids[c.n_pos] = (uint8_t)id; c.pos_delta[c.n_pos] = dd; c.n_pos++;
        }
        else c.neg_ids[c.n_neg++] = (uint8_t)id;
      }
    } else if (motifs.size() == 1) {  // new_motif, single motif arms (76-126)
      for (auto& kv : locs[0]) {
        if (!in(kv.first)) continue;
        auto it = positions.find(kv.first); if (it != positions.end() && !all_single) it->second = rule_combine(it->second, kv.second);
          else positions[kv.first] = kv.second;
        mkp_motif_combo& c = cb(kv.first);
        if (kv.second == R_POS || kv.second == R_BOTH) set_pos(c, {0});
        if (kv.second == R_NEG || kv.second == R_BOTH) set_neg(c, {0});
      }
    } else if (all_single) {  // add_single_base_motifs (204-248)
      auto add = [&](const char* top, const char* bottom) {
        int a = -1, t = -1; for (size_t i = 0; i < motifs.size(); i++) { if (motifs[i].raw == top) a = (int)i;
          if (motifs[i].raw == bottom) t = (int)i;
          }
        if (a < 0) return;
        for (auto& kv : locs[(size_t)a]) {
          if (!in(kv.first)) continue;
          mkp_motif_combo& c = cb(kv.first);
          if (t >= 0) { positions[kv.first] = R_BOTH; set_pos(c, {(size_t)a, (size_t)t}); set_neg(c, {(size_t)a, (size_t)t}); }
          else { positions[kv.first] = kv.second; if (kv.second == R_POS) set_pos(c, {(size_t)a});
            else if (kv.second == R_NEG) set_neg(c, {(size_t)a});
            }
        }
      };
      add("A", "T"); add("C", "G");
    } else {  // mixture arm (157-198)
      for (size_t id = 0; id < motifs.size(); id++) for (auto& kv : locs[id]) {
        if (!in(kv.first)) continue;
        auto it = positions.find(kv.first); if (it != positions.end()) it->second = rule_combine(it->second, kv.second);
          else positions[kv.first] = kv.second;
        mkp_motif_combo& c = cb(kv.first);
        if (kv.second == R_POS || kv.second == R_BOTH) c.pos_ids[c.n_pos++] = (uint8_t)id;
        if (kv.second == R_NEG || kv.second == R_BOTH) c.neg_ids[c.n_neg++] = (uint8_t)id;
      }
    }
    for (auto& kv : positions) {
      uint8_t id = 0; auto it = ids.find(kv.first); if (it != ids.end()) { if (!combine) for (auto& d : it->second.pos_delta) d = -128;
        id = combo_id(it->second); }
      (*focus)[kv.first - rec.start] = (uint8_t)(kv.second | (id << 2));
    }
  }
};

}  // namespace mkp
