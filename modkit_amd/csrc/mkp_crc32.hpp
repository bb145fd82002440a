// CRC-32 (gzip polynomial, RFC 1952 §8) of a BGZF block's inflated bytes.  htslib — which the reference reads through
// (src/pileup/mod.rs:732-743) — rejects a block whose CRC does not match; so does every inflate path here.  zlib 1.2.11's crc32 runs at
// ~1.5 GB/s per core, a third of what the block decoder itself costs, so x86 hosts with PCLMULQDQ fold 64 bytes per step instead
// (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009; the constants are x^k mod P
// for the bit-reflected polynomial 0x1db710641).  Results are checked against zlib over random lengths in tests/test_host_deflate.py.
#pragma once
#include <zlib.h>

#include <cstddef>
#include <cstdint>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace mkp {

#if defined(__x86_64__)
// raw (pre/post-inverted by the caller) CRC over n >= 64 bytes, n a multiple of 16
__attribute__((target("pclmul,sse4.1"))) static inline uint32_t crc32_fold_pclmul(const uint8_t* p, size_t n, uint32_t crc) {
  alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};   // x^(4*128+64), x^(4*128) mod P
  alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};   // x^(128+64), x^128 mod P
  alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};   // x^64 mod P
  alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};   // P, floor(x^64 / P)
  __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
  x1 = _mm_loadu_si128((const __m128i*)(p + 0x00)); x2 = _mm_loadu_si128((const __m128i*)(p + 0x10));
  x3 = _mm_loadu_si128((const __m128i*)(p + 0x20)); x4 = _mm_loadu_si128((const __m128i*)(p + 0x30));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
  x0 = _mm_load_si128((const __m128i*)k1k2);
  p += 64; n -= 64;
  while (n >= 64) {   // four 128-bit lanes folded 512 bits forward per step
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00); x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
      x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11); x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
      x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
    y5 = _mm_loadu_si128((const __m128i*)(p + 0x00)); y6 = _mm_loadu_si128((const __m128i*)(p + 0x10));
    y7 = _mm_loadu_si128((const __m128i*)(p + 0x20)); y8 = _mm_loadu_si128((const __m128i*)(p + 0x30));
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
    x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
    p += 64; n -= 64;
  }
  x0 = _mm_load_si128((const __m128i*)k3k4);   // the four lanes into one
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
  while (n >= 16) {
    x2 = _mm_loadu_si128((const __m128i*)p);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    p += 16; n -= 16;
  }
  // 128 -> 64 bits
  x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
  x3 = _mm_setr_epi32(~0, 0, ~0, 0);
  x1 = _mm_srli_si128(x1, 8); x1 = _mm_xor_si128(x1, x2);
  x0 = _mm_loadl_epi64((const __m128i*)k5k0);
  x2 = _mm_srli_si128(x1, 4); x1 = _mm_and_si128(x1, x3); x1 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
  // Barrett reduction 64 -> 32 bits
  x0 = _mm_load_si128((const __m128i*)poly);
  x2 = _mm_and_si128(x1, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x10); x2 = _mm_and_si128(x2, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

// crc32(0, p, n) as zlib defines it
static inline uint32_t crc32_of(const uint8_t* p, size_t n) {
#if defined(__x86_64__)
  static const bool fold = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  if (fold && n >= 64) {
    const size_t body = n & ~(size_t)15;
    const uint32_t c = ~crc32_fold_pclmul(p, body, ~0u);
    return n == body ? c : (uint32_t)::crc32((uLong)c, p + body, (uInt)(n - body));
  }
#endif
  return (uint32_t)::crc32(::crc32(0L, Z_NULL, 0), p, (uInt)n);
}

}  // namespace mkp
