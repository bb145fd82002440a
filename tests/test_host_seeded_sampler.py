"""`--seed` + `--sampling-frac < 1` on unmapped records: the reference asks rand 0.8.5's StdRng for one `gen_bool(frac)` per record
(src/reads_sampler/record_sampler.rs:29-38, 80-86).  rand / rand_chacha / rand_core are dependencies (Cargo.toml:42), not in the
reference tree: both restatements — the oracle's `mko::StdRng` and the product's `mkp::SeededSampler` — follow the crates' published
algorithm.  Pinned here: the ChaCha block function against the published zero-key keystreams (20 rounds: RFC 7539 A.1 #1; 12 rounds:
draft-strombergson-chacha-test-vectors TC1, 256-bit key) and the product's draws against the oracle's, draw for draw.  Unpinned: the
PCG32 seed expansion and the Bernoulli cut (no fixture of the reference depends on a seeded sample)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include "oracle_core.hpp"
#include "mkp_rand.hpp"
static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); fails++; } } while (0)
int main() {
  const uint32_t ks20[16] = {0xade0b876u, 0x903df1a0u, 0xe56a5d40u, 0x28bd8653u, 0xb819d2bdu, 0x1aed8da0u, 0xccef36a8u, 0xc70d778bu,
                             0x7c5941dau, 0x8d485751u, 0x3fe02477u, 0x374ad8b8u, 0xf4b8436au, 0x1ca11815u, 0x69b687c3u, 0x8665eeb2u};
  const uint32_t ks12[16] = {0x6a9af49bu, 0x53f95507u, 0x12ce1f81u, 0xd583265fu, 0xbbc32904u, 0x1474e049u, 0xa589007eu, 0x5f15ae2eu,
                             0x79f86405u, 0xc0e37ad2u, 0x3428e82cu, 0x798cfaacu, 0x2c9f623au, 0x1969dea0u, 0x2fe80b61u, 0xbe261341u};
  { mko::ChaChaBlocks c; c.rounds = 20; uint32_t o[16]; c.block(o); for (int i = 0; i < 16; i++) CHECK(o[i] == ks20[i]); }
  { mko::ChaChaBlocks c; c.rounds = 12; uint32_t o[16]; c.block(o); for (int i = 0; i < 16; i++) CHECK(o[i] == ks12[i]); CHECK(c.counter == 1); }
  // the product's sampler against the oracle's generator: same verdicts, draw for draw, over block boundaries
  const unsigned long long seeds[] = {0ull, 1ull, 42ull, 0xffffffffffffffffull, 1234567890123ull};
  const double ps[] = {0.0, 1e-9, 0.1, 0.25, 0.5, 0.731, 0.999999, 1.0};
  for (auto seed : seeds) for (double p : ps) {
    mko::StdRng a = mko::StdRng::seed_from_u64(seed); mkp::SeededSampler b(seed);
    int kept = 0, diff = 0;
    for (int i = 0; i < 5000; i++) { const bool x = a.gen_bool(p), y = b.keep(p); kept += x; diff += x != y; }
    CHECK(diff == 0);
    if (p == 0.0) CHECK(kept == 0);
    if (p == 1.0) CHECK(kept == 5000);
    if (p == 0.5) CHECK(kept > 2300 && kept < 2700);
  }
  // p == 1 consumes no draw (Bernoulli's ALWAYS_TRUE): the draws after it are the stream's first
  { mko::StdRng a = mko::StdRng::seed_from_u64(9), b = mko::StdRng::seed_from_u64(9); a.gen_bool(1.0); CHECK(a.next_u64() == b.next_u64()); }
  if (fails) { printf("%d failures\n", fails); return 1; }
  printf("ok\n"); return 0;
}
'''


def test_seeded_record_sampler(tmp_path):
    src, exe = tmp_path / "t.cpp", tmp_path / "t"
    src.write_text(SRC)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "oracle"), "-I", os.path.join(ROOT, "modkit_amd", "csrc"), "-o", str(exe), str(src), "-lz", "-lpthread"])
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stdout + p.stderr
