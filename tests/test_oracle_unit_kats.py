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
