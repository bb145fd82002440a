// The reference's record sampler draws from rand 0.8.5's StdRng (src/reads_sampler/record_sampler.rs:29-38 `StdRng::seed_from_u64`,
// 80-86 `gen_bool(sample_frac)`): with `--seed` its choice of unmapped records under `--sampling-frac < 1` is a pure function of the seed,
// reproduced here from the crates' published algorithms (rand_core 0.6 seed expansion, rand_chacha 0.3 ChaCha12, rand 0.8 Bernoulli).
// Host only: one draw per candidate record, a few thousand per run.
#pragma once
#include <cstdint>
#include <cstring>

namespace mkp {

class SeededSampler {
 public:
  explicit SeededSampler(uint64_t seed) {
    // SeedableRng::seed_from_u64: 32 seed bytes from eight steps of a 64-bit LCG, each output permuted PCG-XSH-RR to 32 bits
    uint64_t s = seed;
    for (int w = 0; w < 8; w++) {
      s = s * 6364136223846793005ull + 11634580027462260723ull;
      uint32_t x = (uint32_t)(((s >> 18) ^ s) >> 27);
      unsigned r = (unsigned)(s >> 59);
      in_[4 + w] = r ? (x >> r) | (x << (32 - r)) : x;
    }
    in_[0] = 0x61707865u; in_[1] = 0x3320646eu; in_[2] = 0x79622d32u; in_[3] = 0x6b206574u;   // "expand 32-byte k"
    in_[12] = in_[13] = in_[14] = in_[15] = 0;                                               // block counter (64 bit), stream id 0
  }
  // Rng::gen_bool -> Bernoulli::new(p): p == 1 needs no draw; else a u64 (two consecutive keystream words, low word first) < p * 2^64
  bool keep(double p) {
    if (p >= 1.0) return true;
    const uint64_t cut = (uint64_t)(p * 18446744073709551616.0);
    const uint64_t lo = word(), hi = word();
    return (lo | (hi << 32)) < cut;
  }

 private:
  uint32_t in_[16], out_[16]; int have_ = 0;
  static void quarter(uint32_t* v, int a, int b, int c, int d) {
    v[a] += v[b]; v[d] ^= v[a]; v[d] = (v[d] << 16) | (v[d] >> 16);
    v[c] += v[d]; v[b] ^= v[c]; v[b] = (v[b] << 12) | (v[b] >> 20);
    v[a] += v[b]; v[d] ^= v[a]; v[d] = (v[d] << 8) | (v[d] >> 24);
    v[c] += v[d]; v[b] ^= v[c]; v[b] = (v[b] << 7) | (v[b] >> 25);
  }
  uint32_t word() {
    if (!have_) {
      memcpy(out_, in_, sizeof out_);
      for (int dr = 0; dr < 6; dr++) {   // ChaCha12: six column + diagonal double rounds
        quarter(out_, 0, 4, 8, 12); quarter(out_, 1, 5, 9, 13); quarter(out_, 2, 6, 10, 14); quarter(out_, 3, 7, 11, 15);
        quarter(out_, 0, 5, 10, 15); quarter(out_, 1, 6, 11, 12); quarter(out_, 2, 7, 8, 13); quarter(out_, 3, 4, 9, 14);
      }
      for (int i = 0; i < 16; i++) out_[i] += in_[i];
      if (++in_[12] == 0) in_[13]++;
      have_ = 16;
    }
    return out_[16 - have_--];
  }
};

}  // namespace mkp
