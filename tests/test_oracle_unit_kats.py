"""Pins the oracle's building blocks against the known answers of the reference's own unit tests (beyond the golden
bedMethyl files): ReDistribute collapse values (src/mod_bam.rs:1956-2022), MultipleThresholdModCaller::call semantics
(src/threshold_mod_caller.rs:204-327), quals_to_probs (808-816), percentile_linear_interp (src/thresholds.rs:17-38,
197-201) and the FxHashMap iteration order pinned by src/mod_bam.rs:2250-2258.  A C++ harness over oracle_core.hpp."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include "oracle_core.hpp"
using namespace mko;
static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); fails++; } } while (0)
static BaseModProbs one(char c, float p) { BaseModProbs b; b.add_base_mod_prob(code_char(c), p); return b; }
static float get(const BaseModProbs& b, char c) { float r = -1.f; b.probs.for_each([&](ModCode k, float p) { if (k == code_char(c)) r = p; }); return r; }
static int count(const BaseModProbs& b) { int n = 0; b.probs.for_each([&](ModCode, float) { n++; }); return n; }
int main() {
  { // test_mod_prob_collapse / _dist_examples (mod_bam.rs:1956-2022)
    BaseModProbs b; b.add_base_mod_prob(code_char('h'), 0.85f); b.add_base_mod_prob(code_char('m'), 0.10f);
    BaseModProbs c = collapse_redistribute(b, code_char('h'));
    CHECK(count(c) == 1); CHECK(get(c, 'm') == 0.52500004f);
    BaseModProbs d = collapse_redistribute(b, code_char('a'));      // absent code: unchanged
    CHECK(count(d) == 2); CHECK(get(d, 'h') == 0.85f); CHECK(get(d, 'm') == 0.10f);
    BaseModProbs e; e.add_base_mod_prob(code_char('h'), 0.05273438f); e.add_base_mod_prob(code_char('m'), 0.03320312f);
    CHECK(get(collapse_redistribute(e, code_char('h')), 'm') == 0.059570313f);
  }
  { // test_multi_threshold_call_semantics, CASE A (threshold_mod_caller.rs:206-230)
    ThresholdCaller k; k.per_mod[code_char('a')] = 0.9f; k.default_threshold = 0.8f;
    CHECK(k.call(0, one('a', 0.8f)).kind == BaseModCall::FILTERED);
    BaseModCall r = k.call(0, one('a', 0.2f)); CHECK(r.kind == BaseModCall::CANONICAL); CHECK(r.p == 0.8f);
    r = k.call(0, one('a', 0.9f)); CHECK(r.kind == BaseModCall::MODIFIED); CHECK(r.p == 0.9f); CHECK(r.code == code_char('a'));
  }
  { // CASE B (232-266)
    ThresholdCaller k; k.per_mod[code_char('a')] = 0.9f; k.per_base[0] = 0.2f; k.default_threshold = 1.0f;
    BaseModCall r = k.call(0, one('a', 0.79f)); CHECK(r.kind == BaseModCall::CANONICAL); CHECK(fabsf(r.p - 0.21f) < 1e-6f);
    r = k.call(0, one('a', 0.6f)); CHECK(r.kind == BaseModCall::CANONICAL); CHECK(fabsf(r.p - 0.4f) < 1e-6f);
    r = k.call(0, one('a', 0.2f)); CHECK(r.kind == BaseModCall::CANONICAL);
    r = k.call(0, one('a', 0.9f)); CHECK(r.kind == BaseModCall::MODIFIED); CHECK(r.code == code_char('a'));
  }
  { // CASE C (268-292): both pass -> most likely
    ThresholdCaller k; k.per_mod[code_char('a')] = 0.8f; k.per_base[0] = 0.2f; k.default_threshold = 1.0f;
    BaseModCall r = k.call(0, one('a', 0.8f)); CHECK(r.kind == BaseModCall::MODIFIED); CHECK(r.p == 0.8f);
    r = k.call(0, one('a', 0.2f)); CHECK(r.kind == BaseModCall::CANONICAL); CHECK(r.p == 0.8f);
    r = k.call(0, one('a', 0.9f)); CHECK(r.kind == BaseModCall::MODIFIED);
  }
  { // test_multi_threshold_passthrough (295-303) and _base_threshold (305-327)
    ThresholdCaller k;
    CHECK(k.call(0, one('a', 0.8f)).kind == BaseModCall::MODIFIED); CHECK(k.call(0, one('a', 0.2f)).kind == BaseModCall::CANONICAL);
    ThresholdCaller j; j.per_mod[code_char('a')] = 0.8f; j.per_base[0] = 0.7f; j.default_threshold = 0.75f;
    CHECK(j.call(0, one('a', 0.75f)).kind == BaseModCall::FILTERED); CHECK(j.call(0, one('a', 0.6f)).kind == BaseModCall::FILTERED);
    CHECK(j.call(0, one('a', 0.2f)).kind == BaseModCall::CANONICAL);
    BaseModCall r = j.call(1, one('m', 0.8f)); CHECK(r.kind == BaseModCall::MODIFIED); CHECK(r.code == code_char('m'));
    CHECK(j.call(1, one('m', 0.72f)).kind == BaseModCall::FILTERED);
  }
  { // test_multi_threshold_call_multiple_mods_semantics (threshold_mod_caller.rs:392-424)
    auto two = [](float m, float h) { BaseModProbs b; b.add_base_mod_prob(code_char('m'), m); b.add_base_mod_prob(code_char('h'), h); return b; };
    ThresholdCaller k; k.per_mod[code_char('m')] = 0.7f; k.per_mod[code_char('h')] = 0.8f; k.per_base[1] = 0.75f; k.default_threshold = 0.0f;
    BaseModCall r = k.call(1, two(0.1f, 0.8f)); CHECK(r.kind == BaseModCall::MODIFIED); CHECK(r.code == code_char('h')); CHECK(r.p == 0.8f);
    CHECK(k.call(1, two(0.2f, 0.7f)).kind == BaseModCall::FILTERED);
    ThresholdCaller j; j.per_mod[code_char('m')] = 0.7f; j.per_mod[code_char('h')] = 0.8f; j.per_base[1] = 0.1f; j.default_threshold = 0.0f;
    r = j.call(1, two(0.2f, 0.7f)); CHECK(r.kind == BaseModCall::CANONICAL); CHECK(fabsf(r.p - 0.1f) < 1e-6f);
  }
  { // FxHashMap order pinned by mod_bam.rs:2250-2258: quals come out [1, 200] = h then m, whichever is inserted first
    BaseModProbs b; b.add_base_mod_prob(code_char('m'), 200.5f / 256.f); b.add_base_mod_prob(code_char('h'), 1.5f / 256.f);
    std::vector<ModCode> order; b.probs.for_each([&](ModCode k, float) { order.push_back(k); });
    CHECK(order.size() == 2 && order[0] == code_char('h') && order[1] == code_char('m'));
  }
  { // percentile_linear_interp (thresholds.rs:17-38; 197-201: q > 1 is an error, < 2 datapoints is an error)
    std::vector<float> xs = {0.25f, 0.5f, 0.75f};
    CHECK(percentile_linear_interp(xs, 0.5f) == 0.5f); CHECK(percentile_linear_interp(xs, 1.0f) == 0.75f);
    CHECK(percentile_linear_interp(xs, 0.25f) == 0.25f * 0.5f + 0.5f * 0.5f);
    bool threw = false; try { percentile_linear_interp(xs, 1.5f); } catch (const MkErr&) { threw = true; } CHECK(threw);
    threw = false; try { percentile_linear_interp(std::vector<float>{0.5f}, 0.1f); } catch (const MkErr&) { threw = true; } CHECK(threw);
  }
  printf(fails ? "FAILED %d\n" : "ok\n", fails);
  return fails ? 1 : 0;
}
'''


def test_oracle_blocks_match_reference_unit_test_vectors(tmp_path):
    src = tmp_path / "kat.cpp"
    src.write_text(SRC)
    exe = tmp_path / "kat"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "oracle"), "-o", str(exe), str(src), "-lz"])
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stdout + p.stderr


FV_SRC = r'''
#include <cstdio>
#include "oracle_core.hpp"
#include "oracle_pileup.hpp"
using namespace mko;
static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); fails++; } } while (0)
// FeatureVector (pileup/mod.rs:226-281): a pair of strand tallies; add_feature routes by (alignment strand == read strand)
// under a StrandRule, exactly as the lambda inside the oracle's process_region does
struct FV {
  Tally pos, neg;
  void add(bool aln_neg, int kind /*0 del,1 filtered,2 nocall,3 can,4 mod*/, int base, ModCode code, bool read_strand_neg, StrandRule rule) {
    const bool to_pos = (aln_neg == read_strand_neg);
    if (rule == RULE_POS && !to_pos) return;
    if (rule == RULE_NEG && to_pos) return;
    Tally& t = to_pos ? pos : neg;
    switch (kind) {
      case 0: t.n_delete++; break;
      case 1: t.n_filtered++; break;
      case 2: t.basecall[base]++; break;
      case 3: t.has_modcall[base] = true; t.canonical[base]++; break;
      case 4: t.has_modcall[base] = true; t.modified[base][code]++; break;
    }
  }
  std::vector<Row> decode(const std::set<ModCode> pos_obs[4], const std::set<ModCode> neg_obs[4]) const {
    PileupOptions o; std::vector<Row> counts;
    add_tally_to_counts(counts, pos, '+', pos_obs, o, nullptr, 0);
    add_tally_to_counts(counts, neg, '-', neg_obs, o, nullptr, 0);
    return counts;
  }
};
int main() {
  const ModCode mc = code_char('m'), hmc = code_char('h');
  { // test_feature_vector_basic, first half (pileup/mod.rs:1039-1103)
    std::set<ModCode> pos_obs[4], neg_obs[4]; pos_obs[BC] = {mc, hmc};
    FV fv;
    fv.add(false, 2, BA, 0, false, RULE_BOTH);
    fv.add(false, 3, BC, 0, false, RULE_BOTH);
    fv.add(false, 4, BC, mc, false, RULE_BOTH);
    fv.add(false, 4, BC, mc, false, RULE_BOTH);
    fv.add(false, 2, BC, 0, false, RULE_BOTH);
    fv.add(true, 2, BG, 0, false, RULE_BOTH);
    fv.add(true, 2, BG, 0, false, RULE_BOTH);
    std::vector<Row> counts = fv.decode(pos_obs, neg_obs);
    CHECK(counts.size() == 2);   // h and m; the negative strand has no mod call
    for (auto& r : counts) { CHECK(r.cov == 3); CHECK(r.n_nocall == 1); CHECK(r.n_diff == 1); CHECK(r.strand == '+'); }
  }
  { // second half (1104-1144)
    std::set<ModCode> pos_obs[4], neg_obs[4]; pos_obs[BC] = {mc, hmc}; neg_obs[BC] = {mc, hmc};
    FV fv;
    fv.add(false, 3, BC, 0, false, RULE_BOTH);
    fv.add(true, 4, BC, mc, false, RULE_BOTH);
    fv.add(true, 2, BG, 0, false, RULE_BOTH);
    fv.add(true, 2, BG, 0, false, RULE_BOTH);
    std::vector<Row> counts = fv.decode(pos_obs, neg_obs);
    CHECK(counts.size() == 4);
    int n_neg = 0; for (auto& r : counts) if (r.strand == '-') { CHECK(r.n_diff == 2); n_neg++; }
    CHECK(n_neg == 2);
  }
  { // test_feature_vector_with_strand_rules (1147-1177): the feature on the wrong strand is ignored
    std::set<ModCode> pos_obs[4], neg_obs[4]; pos_obs[BC] = {mc};
    FV fv;
    fv.add(false, 4, BC, mc, false, RULE_POS);
    fv.add(true, 4, BC, mc, false, RULE_POS);
    std::vector<Row> counts = fv.decode(pos_obs, neg_obs);
    CHECK(counts.size() == 1); CHECK(counts.size() == 1 && counts[0].n_mod == 1);
  }
  { // StrandRule algebra (util.rs:297-349, test_strand_rule_semantics)
    CHECK(rule_combine(RULE_POS, RULE_POS) == RULE_POS); CHECK(rule_combine(RULE_POS, RULE_NEG) == RULE_BOTH); CHECK(rule_combine(RULE_BOTH, RULE_NEG) == RULE_BOTH);
  }
  printf(fails ? "FAILED %d\n" : "ok\n", fails);
  return fails ? 1 : 0;
}
'''


def test_oracle_feature_vector_matches_reference_unit_tests(tmp_path):
    src = tmp_path / "fv.cpp"
    src.write_text(FV_SRC)
    exe = tmp_path / "fv"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "oracle"), "-o", str(exe), str(src), "-lz"])
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stdout + p.stderr


MM_SRC = r'''
#include <cstdio>
#include "oracle_core.hpp"
using namespace mko;
static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); fails++; } } while (0)
static BamRecord rec(const std::string& dna, const std::string& mm, const std::vector<uint8_t>& ml) {
  BamRecord r; r.tid = 0; r.pos = 0; r.flag = 0; r.l_seq = (int32_t)dna.size(); r.qname = "t"; r.seq = dna; r.cigar = {(uint32_t)dna.size() << 4};
  r.aux.insert(r.aux.end(), {'M', 'M', 'Z'}); r.aux.insert(r.aux.end(), mm.begin(), mm.end()); r.aux.push_back(0);
  r.aux.insert(r.aux.end(), {'M', 'L', 'B', 'C'}); uint32_t n = (uint32_t)ml.size(); const uint8_t* p = (const uint8_t*)&n; r.aux.insert(r.aux.end(), p, p + 4);
  r.aux.insert(r.aux.end(), ml.begin(), ml.end());
  return r;
}
static std::vector<size_t> keys(const SeqPosBaseModProbs& s) { std::vector<size_t> k; for (auto& kv : s.pos) k.push_back(kv.first); return k; }
static float get(const BaseModProbs& b, char c) { float r = -1.f; b.probs.for_each([&](ModCode k, float p) { if (k == code_char(c)) r = p; }); return r; }
static std::vector<size_t> edge(const SeqPosBaseModProbs& s, const EdgeFilter& f, size_t len, bool* trimmable) {
  // SeqPosBaseModProbs::edge_filter_positions (mod_bam.rs:1075-1102)
  std::vector<size_t> k; *trimmable = f.read_can_be_trimmed(len);
  if (*trimmable) for (auto& kv : s.pos) if (f.keep_position(kv.first, len)) k.push_back(kv.first);
  return k;
}
int main() {
  const std::vector<uint32_t> DL = {5, 2, 1, 3, 1, 2, 3, 1, 2, 1, 11, 5};
  { // test_mod_parse_base_positions (mod_bam.rs:2193-2227)
    MmTagInfo a = parse_one_mm("C+h?,5,2,1,3,1,2,3,1,2,1,11,5");
    CHECK(a.fundamental_base == 'C' && a.mode == MODE_EXPLICIT && !a.neg_strand && a.codes == std::vector<ModCode>{code_char('h')} && a.deltas == DL);
    MmTagInfo b = parse_one_mm("C+m,5,2,1,3,1,2,3,1,2,1,11,5");
    CHECK(b.mode == MODE_DEFAULT_IMPLICIT && b.codes == std::vector<ModCode>{code_char('m')} && b.deltas == DL);
    MmTagInfo c = parse_one_mm("C+m.,5,2,1,3,1,2,3,1,2,1,11,5");
    CHECK(c.mode == MODE_IMPLICIT && c.deltas == DL);
  }
  const std::string dna = "GATCGACTACGTCGA";
  { // test_get_base_mod_probs (2229-2275): combined and split tags give the same map; quals come out [1, 200] = (h, m)
    ModBaseInfo x = mod_base_info_from_record(rec(dna, "C+hm?,0,1,0;", {1, 200, 1, 200, 1, 200}));
    ModBaseInfo y = mod_base_info_from_record(rec(dna, "C+h?,0,1,0;C+m?,0,1,0;", {1, 1, 1, 200, 200, 200}));
    CHECK(keys(x.pos_strand[BC]) == (std::vector<size_t>{3, 9, 12})); CHECK(keys(y.pos_strand[BC]) == (std::vector<size_t>{3, 9, 12}));
    for (size_t p : {3, 9, 12}) {
      CHECK(get(x.pos_strand[BC].pos[p], 'h') == 1.5f / 256.f); CHECK(get(x.pos_strand[BC].pos[p], 'm') == 200.5f / 256.f);
      CHECK(get(y.pos_strand[BC].pos[p], 'h') == 1.5f / 256.f); CHECK(get(y.pos_strand[BC].pos[p], 'm') == 200.5f / 256.f);
    }
  }
  { // test_extract_positions_to_probs (2277-2326): the A+a block sits between the C blocks in ML
    ModBaseInfo x = mod_base_info_from_record(rec(dna, "C+h?,0,1,0;A+a?,0,1,0;C+m?,0,1,0;", {1, 1, 1, 200, 200, 200, 1, 1, 1}));
    ModBaseInfo y = mod_base_info_from_record(rec(dna, "C+hm?,0,1,0;A+a?,0,1,0;", {1, 1, 1, 1, 1, 1, 200, 200, 200}));
    CHECK(keys(x.pos_strand[BC]).size() == 3);
    for (size_t p : {3, 9, 12}) {
      CHECK(get(x.pos_strand[BC].pos[p], 'h') == 0.005859375f); CHECK(get(x.pos_strand[BC].pos[p], 'm') == 0.005859375f);
      CHECK(get(y.pos_strand[BC].pos[p], 'h') == 0.005859375f); CHECK(get(y.pos_strand[BC].pos[p], 'm') == 0.005859375f);
    }
    CHECK(keys(x.pos_strand[BA]).size() == 3); CHECK(get(x.pos_strand[BA].pos.begin()->second, 'a') == 200.5f / 256.f);
  }
  { // test_seq_pos_base_mod_probs_edge_filter (2621-2699)
    ModBaseInfo x = mod_base_info_from_record(rec(dna, "C+h?,0,1,0;A+a?,0,1,0;C+m?,0,1,0;", {1, 1, 1, 200, 200, 200, 100, 100, 100}));
    bool ok; EdgeFilter f; f.active = true;
    f.start = 4; f.end = 4; CHECK(edge(x.pos_strand[BC], f, dna.size(), &ok) == (std::vector<size_t>{9})); CHECK(ok);
    f.start = 50; f.end = 50; edge(x.pos_strand[BC], f, dna.size(), &ok); CHECK(!ok);
    f.start = 3; f.end = 3; CHECK(edge(x.pos_strand[BC], f, dna.size(), &ok) == (std::vector<size_t>{3, 9}));
  }
  { // test_mod_bam_modbase_info_empty (2704-2765)
    CHECK(mod_base_info_from_record(rec(dna, "C+h?;C+m?;", {})).is_empty());
    CHECK(!mod_base_info_from_record(rec(dna, "C+h.;C+m.;", {})).is_empty());
    const std::string d2 = "GACTCGACTGGACGTCGA";
    CHECK(mod_base_info_from_record(rec(d2, "C+h?;C+m?;G-h?;G-m?;", {})).is_empty());
    ModBaseInfo z = mod_base_info_from_record(rec(d2, "C+h.;C+m.;G-h.;G-m.;", {}));
    CHECK(!z.is_empty()); CHECK(z.pos_strand.size() == 1);
    CHECK(keys(z.pos_strand[BC]) == (std::vector<size_t>{2, 4, 7, 12, 15}));
    for (auto& kv : z.pos_strand[BC].pos) { CHECK(kv.second.inferred); CHECK(kv.second.canonical_prob() == 1.0f); }
    CHECK(keys(z.neg_strand[BG]) == (std::vector<size_t>{0, 5, 9, 10, 13, 16}));
    for (auto& kv : z.neg_strand[BG].pos) { CHECK(kv.second.inferred); CHECK(kv.second.canonical_prob() == 1.0f); }
  }
  { // test_delta_list_converter_n_base / test_generic_mm_tags (2776-2803): `N` deltas count every base
    const std::string d3 = "GCGGATTTCTGAGTTTG";
    ModBaseInfo n = mod_base_info_from_record(rec(d3, "N+b?,5,0,0,1,3,0,0;", {255, 255, 255, 255, 255, 255, 255}));
    std::vector<size_t> all; for (auto& kv : n.pos_strand) for (auto& pk : kv.second.pos) all.push_back(pk.first);
    std::sort(all.begin(), all.end());
    CHECK(all == (std::vector<size_t>{5, 6, 7, 9, 13, 14, 15}));
  }
  { // test_mod_base_info (2327-2470): interleaved C+hm? equals split C+h?/C+m? whatever sits between the blocks in ML
    const float H = 0.005859375f, M = 0.39257813f, A = 0.7832031f;
    for (auto& tq : std::vector<std::pair<std::string, std::vector<uint8_t>>>{
             {"C+hm?,0,1,0;A+a?,0,1,0;", {1, 100, 1, 100, 1, 100, 200, 200, 200}},
             {"C+h?,0,1,0;A+a?,0,1,0;C+m?,0,1,0;", {1, 1, 1, 200, 200, 200, 100, 100, 100}},
             {"C+h?,0,1,0;C+m?,0,1,0;A+a?,0,1,0;", {1, 1, 1, 100, 100, 100, 200, 200, 200}}}) {
      ModBaseInfo x = mod_base_info_from_record(rec(dna, tq.first, tq.second));
      CHECK(keys(x.pos_strand[BC]) == (std::vector<size_t>{3, 9, 12})); CHECK(keys(x.pos_strand[BA]) == (std::vector<size_t>{1, 8, 14}));
      for (auto& kv : x.pos_strand[BC].pos) { CHECK(get(kv.second, 'h') == H); CHECK(get(kv.second, 'm') == M); CHECK(!kv.second.inferred); }
      for (auto& kv : x.pos_strand[BA].pos) CHECK(get(kv.second, 'a') == A);
    }
  }
  { // test_duplex_modbase_info / _implicit (2512-2605): top-strand C calls and bottom-strand G calls of one duplex read
    const std::string d2 = "GACTCGACTGGACGTCGA";
    ModBaseInfo x = mod_base_info_from_record(rec(d2, "C+h?,1,1,0;C+m?,1,1,0;G-h?,1,2,0;G-m?,1,2,0", {100, 100, 100, 1, 1, 1, 150, 150, 150, 2, 2, 2}));
    CHECK(keys(x.pos_strand[BC]) == (std::vector<size_t>{4, 12, 15})); CHECK(keys(x.neg_strand[BG]) == (std::vector<size_t>{5, 13, 16}));
    for (auto& kv : x.pos_strand[BC].pos) { CHECK(get(kv.second, 'h') == 0.39257813f); CHECK(get(kv.second, 'm') == 0.005859375f); }
    for (auto& kv : x.neg_strand[BG].pos) { CHECK(get(kv.second, 'h') == 0.5878906f); CHECK(get(kv.second, 'm') == 0.009765625f); }
    ModBaseInfo y = mod_base_info_from_record(rec(d2, "C+h.,1,1,0;C+m.,1,1,0;G-h.,1,2,0;G-m.,1,2,0", {100, 100, 100, 1, 1, 1, 150, 150, 150, 2, 2, 2}));
    CHECK(y.pos_strand[BC].pos.size() == 5); CHECK(y.neg_strand[BG].pos.size() == 6);
    for (size_t p : {2, 7}) { CHECK(y.pos_strand[BC].pos[p].inferred); CHECK(y.pos_strand[BC].pos[p].canonical_prob() == 1.0f); }
    for (size_t p : {0, 9, 10}) { CHECK(y.neg_strand[BG].pos[p].inferred); CHECK(y.neg_strand[BG].pos[p].canonical_prob() == 1.0f); }
  }
  { // test_generic_mm_tags_multibase_conflict / _mixed_modes / _inferred_conflict (2829-2882)
    const std::string d3 = "GCGGATTTCTGAGTTTG";
    auto throws = [&](const std::string& dn, const std::string& mm, const std::vector<uint8_t>& ml, const char* what) {
      try { mod_base_info_from_record(rec(dn, mm, ml)); } catch (const MkErr& e) { return what == nullptr || std::string(e.what()).find(what) != std::string::npos; }
      return false;
    };
    CHECK(throws(d3, "N+b?,1,3,0,0,1,3,0,0;C+m?,0;", std::vector<uint8_t>(9, 255), nullptr));
    CHECK(throws(d3, "C+m.;N+b?,1,3,0,0,1,3,0,0;", std::vector<uint8_t>(8, 255), nullptr));
    bool ok = true; try { mod_base_info_from_record(rec("CATCACA", "N+b?,0,1;C+m.,0,1;", {200, 255, 50, 0})); } catch (const MkErr&) { ok = false; } CHECK(ok);
    CHECK(throws("CATCACA", "C+mh.,0,1;C+h.,0", {200, 0, 0, 200, 25}, "conflict-explicit-and-inferred"));
  }
  { // test_quals_and_probs (2767-2774): (q + 0.5) / 256 round-trips through prob_to_qual for every byte
    for (int q = 0; q < 256; q++) { float p = ((float)q + 0.5f) / 256.0f; int back = (int)(p * 256.0f); if (back > 255) back = 255; CHECK(back == q); }
  }
  printf(fails ? "FAILED %d\n" : "ok\n", fails);
  return fails ? 1 : 0;
}
'''


def test_oracle_mm_ml_extraction_matches_reference_unit_tests(tmp_path):
    src = tmp_path / "mm.cpp"
    src.write_text(MM_SRC)
    exe = tmp_path / "mm"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "oracle"), "-o", str(exe), str(src), "-lz"])
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stdout + p.stderr
