"""The library's host-side motif / strand-rule / interval code (modkit_amd/csrc/mkp_focus.hpp, product code) against the known
answers of the reference's unit tests: RegexMotif offsets, find_motif_hits, overlapping motifs, palindromes
(src/find_motifs/motif_bed.rs:674-759) and StrandRule algebra (src/util.rs:297-349).  A C++ harness; no device."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include "mkp_focus.hpp"
using namespace mkp;
static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); fails++; } } while (0)
typedef std::vector<std::pair<uint32_t, Rule>> Hits;
static Hits hits(const std::string& seq, const Motif& m, uint64_t off = 0) {
  std::map<uint32_t, Rule> out; motif_hits(seq.data(), seq.size(), m, off, 0, nullptr, &out);
  return Hits(out.begin(), out.end());
}
int main() {
  { // test_regex_motif (motif_bed.rs:674-685)
    Motif a = Motif::parse("CCWGG", 1); CHECK(a.fwd_off == 1); CHECK(a.rev_off == 3); CHECK(a.len() == 5);
    Motif b = Motif::parse("CG", 0); CHECK(b.rev_off == 1);
    Motif c = Motif::parse("CGCG", 2); int d = 99; CHECK(c.neg_delta(&d)); CHECK(d == -1);   // MotifInfo::offset()
  }
  { // test_motif_hits (687-718)
    Motif m = Motif::parse("CGCG", 2);
    Hits h = hits("AACGCGAACGCGA", m);
    CHECK(h == (Hits{{3, R_NEG}, {4, R_POS}, {9, R_NEG}, {10, R_POS}}));
    int d; CHECK(m.neg_delta(&d));
    for (auto& x : h) if (x.second == R_POS) { bool found = false; for (auto& y : h) found |= (y.second == R_NEG && (int)y.first == (int)x.first + d); CHECK(found); }
    CHECK(hits("AACGCGAACGCGA", m, 1000) == (Hits{{1003, R_NEG}, {1004, R_POS}, {1009, R_NEG}, {1010, R_POS}}));   // interval offset
  }
  { // test_overlapping_motifs (720-747)
    Motif m = Motif::parse("CHH", 0);
    CHECK(hits("AACCCCTG", m) == (Hits{{2, R_POS}, {3, R_POS}, {4, R_POS}}));
    CHECK(hits("ACCTAG", m) == (Hits{{1, R_POS}, {2, R_POS}, {5, R_NEG}}));
  }
  { // test_motif_palindrome (749-759)
    CHECK(!Motif::parse("CHH", 0).palindrome); CHECK(Motif::parse("CG", 0).palindrome); CHECK(!Motif::parse("C", 0).palindrome); CHECK(Motif::parse("GATC", 1).palindrome);
    int d; CHECK(!Motif::parse("CHH", 0).neg_delta(&d));
  }
  { // single-base motif: both strands from one pass; degenerate single bases are refused (parse_string 197-206)
    CHECK(hits("ACGT", Motif::parse("C", 0)) == (Hits{{1, R_POS}, {2, R_NEG}}));
    bool threw = false; try { Motif::parse("N", 0); } catch (const Error&) { threw = true; } CHECK(threw);
    threw = false; try { Motif::parse("CG", 2); } catch (const Error&) { threw = true; } CHECK(threw);
  }
  { // StrandRule::combine / absorb (util.rs:333-349)
    CHECK(rule_combine(R_POS, R_POS) == R_POS); CHECK(rule_combine(R_POS, R_NEG) == R_BOTH); CHECK(rule_combine(R_BOTH, R_NEG) == R_BOTH);
    CHECK(rule_absorb(R_POS, false) == R_POS); CHECK(rule_absorb(R_POS, true) == R_BOTH); CHECK(rule_absorb(R_NEG, true) == R_NEG); CHECK(rule_absorb(R_BOTH, false) == R_BOTH);
  }
  { // rust-lapper merge_overlaps as used by StrandedPositionFilter: touching intervals merge; membership is half-open
    std::vector<Span> v = {{10, 20}, {20, 30}, {5, 7}, {40, 50}, {45, 47}};
    merge_spans(v);
    CHECK(v.size() == 3 && v[0].s == 5 && v[0].e == 7 && v[1].s == 10 && v[1].e == 30 && v[2].s == 40 && v[2].e == 50);
    CHECK(spans_hit(v, 29, 30)); CHECK(!spans_hit(v, 30, 31)); CHECK(!spans_hit(v, 7, 10)); CHECK(spans_hit(v, 0, 6));
  }
  printf(fails ? "FAILED %d\n" : "ok\n", fails);
  return fails ? 1 : 0;
}
'''


def test_library_motif_code_matches_reference_unit_tests(tmp_path):
    src = tmp_path / "focus.cpp"
    src.write_text(SRC)
    exe = tmp_path / "focus"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "modkit_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src), "-lz", "-pthread"])
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stdout + p.stderr


# The strand-combining interval walk (get_motif_positions_combine_strands, src/fasta.rs:92-188 + FocusPositions::
# new_motif_combine_strands, src/interval_chunks.rs:250-297): the library finds the interval ends in one sequential pass that looks
# at the hits around each nominal end only and fills the focus bytes on all cores; here the contig is walked the plain way, one
# interval after the other over the whole slice, and intervals, focus bytes and combo tables must be identical.
WALK_SRC = r'''
#include <cstdio>
#include <random>
#include "mkp_focus.hpp"
using namespace mkp;
static std::vector<Interval> walk_plain(FocusBuilder& fb, const Contig& rec, uint32_t interval_size, std::vector<uint8_t>* focus, const std::string& seq) {
  std::vector<Interval> ivs; focus->assign(rec.length, 0);
  size_t longest = 0; for (auto& m : fb.motifs) longest = std::max(longest, m.len());
  uint32_t pos = rec.start;
  while (pos < rec.end()) {
    uint32_t end = (uint32_t)std::min<uint64_t>((uint64_t)pos + interval_size, rec.end());
    std::vector<std::map<uint32_t, Rule>> locs(fb.motifs.size());
    uint64_t ref_end = rec.end(), buffer = longest * 5, e = end, end_w = std::min<uint64_t>((uint64_t)end + buffer, ref_end);
    for (;;) {
      if (end_w > seq.size()) throw Error(MKP_E_UNSUPPORTED, "past the contig end");
      for (auto& l : locs) l.clear();
      for (size_t i = 0; i < fb.motifs.size(); i++) motif_hits(seq.data() + pos, (size_t)(end_w - pos), fb.motifs[i], pos, rec.tid, nullptr, &locs[i]);
      std::vector<Span> sp;
      for (size_t i = 0; i < fb.motifs.size(); i++) { uint64_t adj = fb.motifs[i].len() >= fb.motifs[i].fwd_off ? fb.motifs[i].len() - fb.motifs[i].fwd_off : fb.motifs[i].len(); for (auto& kv : locs[i]) sp.push_back({kv.first, kv.first + adj}); }
      merge_spans(sp);
      uint64_t search_end = e, qs = e ? e - 1 : 0;
      for (auto& s : sp) if (s.s < e && s.e > qs) { search_end = s.e; break; }
      uint64_t too_close = end_w >= longest ? end_w - longest : 0;
      if (search_end < too_close || end_w >= ref_end) { for (auto& l : locs) for (auto it = l.begin(); it != l.end();) { if (it->first <= search_end) ++it; else it = l.erase(it); } end = (uint32_t)std::min<uint64_t>(search_end, rec.end()); break; }
      e = end_w; end_w += buffer;
    }
    fb.fill_motif(locs, rec, pos, end, focus);
    ivs.push_back({rec.tid, pos, end});
    pos = end;
  }
  return ivs;
}
int main() {
  std::mt19937_64 rng(11); int bad = 0, cases = 0;
  for (int it = 0; it < 60; it++) {
    uint32_t L = 2000 + rng() % 30000; std::string seq(L, 'A');
    for (uint32_t i = 0; i < L; i++) seq[i] = "ACGT"[rng() & 3];
    if (it % 3) for (int k = 0; k < 40; k++) { uint32_t p = rng() % (L - 200), n = 2 + rng() % 90; for (uint32_t j = 0; j < n && p + j < L; j++) seq[p + j] = (it % 3 == 2 ? "GATC"[j & 3] : "CG"[j & 1]); }
    Fasta fa; fa.seqs["c"] = seq;
    for (int mset = 0; mset < 3; mset++) for (uint32_t isz : {37u, 100u, 501u, 4096u, 100000u}) {
      FocusBuilder a, b;
      for (FocusBuilder* f : {&a, &b}) {
        f->fasta = &fa; f->combine = true; f->mask = true;   // mask: the sequence is used as it is
        if (mset == 0) f->motifs.push_back(Motif::parse("CG", 0));
        else if (mset == 1) { f->motifs.push_back(Motif::parse("CG", 0)); f->motifs.push_back(Motif::parse("GATC", 1)); }
        else { f->motifs.push_back(Motif::parse("CCGG", 1)); f->motifs.push_back(Motif::parse("CG", 0)); f->motifs.push_back(Motif::parse("GC", 1)); }
      }
      Contig rec; rec.tid = 0; rec.name = "c"; rec.start = (it % 5 == 0) ? 123 : 0; rec.length = L - rec.start - ((it % 7 == 0) ? 57 : 0);
      std::vector<uint8_t> fa_bytes, fb_bytes; std::vector<Interval> ia, ib; bool ea = false, eb = false;
      try { ia = a.walk(rec, isz, &fa_bytes); } catch (const Error&) { ea = true; }
      try { ib = walk_plain(b, rec, isz, &fb_bytes, seq); } catch (const Error&) { eb = true; }
      cases++;
      bool same = ea == eb;
      if (same && !ea) {
        same = ia.size() == ib.size() && fa_bytes == fb_bytes && a.combos.size() == b.combos.size();
        for (size_t k = 0; same && k < ia.size(); k++) same = ia[k].start == ib[k].start && ia[k].end == ib[k].end;
        for (size_t k = 0; same && k < a.combos.size(); k++) same = memcmp(&a.combos[k], &b.combos[k], sizeof(mkp_motif_combo)) == 0;
        FocusBuilder c2; c2.fasta = &fa; c2.combine = true; c2.mask = true; c2.motifs = a.motifs;   // the grid alone (what ranks that do not own a contig compute)
        auto ic = c2.walk(rec, isz, nullptr); same = same && ic.size() == ia.size(); for (size_t k = 0; same && k < ia.size(); k++) same = ic[k].end == ia[k].end;
      }
      if (!same) { bad++; if (bad < 5) printf("MISMATCH it=%d mset=%d isz=%u errs=%d/%d n=%zu/%zu\n", it, mset, isz, ea, eb, ia.size(), ib.size()); }
    }
  }
  printf(bad ? "FAILED %d of %d\n" : "ok %d\n", bad ? bad : cases, cases);
  return bad != 0;
}
'''


def test_combine_strands_walk_matches_interval_by_interval_walk(tmp_path):
    src = tmp_path / "walk.cpp"
    src.write_text(WALK_SRC)
    exe = tmp_path / "walk"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "modkit_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           "-o", str(exe), str(src), "-lz", "-pthread"])
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip().startswith("ok"), p.stdout + p.stderr


FAST_SRC = r"""
#include <cstdio>
#include <random>
#include "mkp_focus.hpp"
using namespace mkp;
// FocusBuilder::walk with the one-motif fast path (fill_single) against the same walk through motif_hits + fill_motif: focus bytes, grid
// and combo tables must be identical for every motif shape (palindromic, IUPAC, single base, off-centre focus), with and without strand
// combining and a BED filter, at interval sizes that cut through motif runs.
int main() {
  std::mt19937_64 rng(29); int bad = 0, cases = 0;
  const char* specs[][2] = {{"CG", "0"}, {"GATC", "1"}, {"CHH", "0"}, {"C", "0"}, {"CCWGG", "1"}, {"A", "0"}, {"GC", "1"}, {"CGCG", "2"}, {"DRACH", "2"}};
  for (int it = 0; it < 40; it++) {
    uint32_t L = 1500 + rng() % 20000; std::string seq(L, 'A');
    for (uint32_t i = 0; i < L; i++) seq[i] = "ACGT"[rng() & 3];
    if (it % 3) for (int k = 0; k < 30; k++) { uint32_t p = rng() % (L - 200), n = 2 + rng() % 90; for (uint32_t j = 0; j < n && p + j < L; j++) seq[p + j] = (it % 3 == 2 ? "GATC"[j & 3] : "CG"[j & 1]); }
    if (it % 4 == 1) for (int k = 0; k < 20; k++) seq[rng() % L] = 'N';
    Fasta fa; fa.seqs["c"] = seq;
    BedFilter bf; { std::vector<Span> p, n; for (int k = 0; k < 25; k++) { uint64_t a = rng() % L, b = a + 1 + rng() % 400; (k & 1 ? p : n).push_back({a, b}); if (k % 5 == 0) { p.push_back({a, b}); n.push_back({a, b}); } }
      merge_spans(p); merge_spans(n); bf.pos[0] = p; bf.neg[0] = n; }
    for (auto& sp : specs) for (int comb = 0; comb < 2; comb++) for (int with_bed = 0; with_bed < 2; with_bed++) for (uint32_t isz : {37u, 100u, 501u, 4096u, 100000u}) {
      Motif m = Motif::parse(sp[0], (size_t)atoi(sp[1]));
      if (comb && !m.palindrome) continue;   // --combine-strands needs a palindromic motif
      FocusBuilder a, b;
      for (FocusBuilder* f : {&a, &b}) { f->fasta = &fa; f->combine = comb; f->mask = true; f->motifs.push_back(m); f->bed = with_bed ? &bf : nullptr; }
      b.fast_single = false;
      Contig rec; rec.tid = 0; rec.name = "c"; rec.start = (it % 5 == 0) ? 123 : 0; rec.length = L - rec.start - ((it % 7 == 0) ? 57 : 0);
      std::vector<uint8_t> xa, xb; std::vector<Interval> ia, ib; bool ea = false, eb = false;
      try { ia = a.walk(rec, isz, &xa); } catch (const Error&) { ea = true; }
      try { ib = b.walk(rec, isz, &xb); } catch (const Error&) { eb = true; }
      cases++;
      bool same = ea == eb;
      if (same && !ea) {
        same = ia.size() == ib.size() && xa == xb && a.combos.size() == b.combos.size();
        for (size_t k = 0; same && k < ia.size(); k++) same = ia[k].start == ib[k].start && ia[k].end == ib[k].end;
        for (size_t k = 0; same && k < a.combos.size(); k++) same = memcmp(&a.combos[k], &b.combos[k], sizeof(mkp_motif_combo)) == 0;
      }
      if (!same) { bad++; if (bad < 6) printf("MISMATCH it=%d motif=%s comb=%d bed=%d isz=%u errs=%d/%d\n", it, sp[0], comb, with_bed, isz, ea, eb); }
    }
  }
  // soft-masked (lower-case) reference: without --mask-reference the contig counts as upper-cased, with it lower-case bases never match
  for (int it = 0; it < 12; it++) {
    uint32_t L = 3000 + rng() % 9000; std::string mixed(L, 'A');
    for (uint32_t i = 0; i < L; i++) mixed[i] = "ACGTacgt"[rng() & 7];
    std::string upper = mixed, masked = mixed;
    for (auto& ch : upper) ch = (char)toupper((unsigned char)ch);
    for (auto& ch : masked) if (ch >= 'a' && ch <= 'z') ch = 'N';
    for (auto& sp : specs) for (int nm = 1; nm <= 2; nm++) for (int mk = 0; mk < 2; mk++) {
      Fasta f1, f2; f1.seqs["c"] = mixed; f2.seqs["c"] = mk ? masked : upper;
      FocusBuilder a, b;
      a.fasta = &f1; b.fasta = &f2; a.mask = mk; b.mask = true;
      for (FocusBuilder* f : {&a, &b}) { f->motifs.push_back(Motif::parse(sp[0], (size_t)atoi(sp[1]))); if (nm == 2) f->motifs.push_back(Motif::parse("CG", 0)); f->combine = false; }
      Contig rec; rec.tid = 0; rec.name = "c"; rec.start = 0; rec.length = L;
      std::vector<uint8_t> xa, xb; a.walk(rec, 777, &xa); b.walk(rec, 777, &xb);
      cases++;
      if (xa != xb) { bad++; if (bad < 6) printf("CASE MISMATCH it=%d motif=%s motifs=%d mask=%d\n", it, sp[0], nm, mk); }
    }
  }
  printf(bad ? "FAILED %d of %d\n" : "ok %d\n", bad ? bad : cases, cases);
  return bad != 0;
}
"""


def test_single_motif_fast_path_equals_the_map_path(tmp_path):
    src = tmp_path / "fast.cpp"
    src.write_text(FAST_SRC)
    exe = tmp_path / "fast"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "modkit_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           "-o", str(exe), str(src), "-lz", "-pthread"])
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip().startswith("ok"), p.stdout + p.stderr
