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
