// Per-lane pieces of the speculative wave inflate (mkp_inflate_wave2.hip): what a position of a DEFLATE bit stream decodes to IF a
// literal/length code starts there.  Every lane of a wave decodes the position `bit + lane`; which of the 64 answers are real — the
// chain 0 -> n(0) -> n(0) + n(n(0)) ... — is settled afterwards by a scalar walk.  No cross-lane operation in here: the file also
// compiles for the host, where tests/inflate_wave2_emul.cpp runs the same functions over 64 emulated lanes against zlib.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#define MKP_TOK_HD __host__ __device__ __forceinline__
#else
#define MKP_TOK_HD inline
#endif

namespace {
// RFC 1951 §3.2.5 as arithmetic: length symbols 257..285 (ls = symbol - 257), distance symbols 0..29
MKP_TOK_HD uint32_t len_extra(int ls) { return (ls < 8 || ls == 28) ? 0u : (uint32_t)(ls >> 2) - 1u; }
MKP_TOK_HD uint32_t len_base(int ls) { return ls < 8 ? 3u + (uint32_t)ls : ls == 28 ? 258u : 3u + ((4u + ((uint32_t)ls & 3u)) << ((uint32_t)(ls >> 2) - 1u)); }
MKP_TOK_HD uint32_t dist_extra(int ds) { return ds < 4 ? 0u : (uint32_t)(ds >> 1) - 1u; }
MKP_TOK_HD uint32_t dist_base(int ds) { return ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1u)) << ((uint32_t)(ds >> 1) - 1u)); }

// What a lane reports for its bit position: `a` = [0:5] bits the token takes, [6] WIN: a token the output window takes as it is — a
// literal, or a match no longer than a window (64 bytes) that does not overlap itself (dist >= len) — [7] LIT, [8:16] output bytes
// (1 | match length); `b` = the window entry of a literal (MKP_SV_LITERAL | byte) | the distance of a match.  Everything else — end of
// block, codes longer than the direct tables, long or self-overlapping matches, invalid codes — has WIN clear: the walk stops there and
// decodes that one token on its own.
enum : uint32_t { MKP_TA_WIN = 64u, MKP_TA_LIT = 128u, MKP_SV_LITERAL = 0x80000000u };
enum : uint32_t { MKP_TK_LIT = 0u, MKP_TK_MATCH = 1u, MKP_TK_EOB = 2u };
struct MkpTok { uint32_t a, b; };

// 64 stream bits starting at bit `q` of the block input, from the 1 KiB circular LDS window (256 dwords; dword i holds input bytes
// [4i, 4i + 4) mod 1024).  Reads 12 bytes from byte 4 * (q >> 5) on.
MKP_TOK_HD unsigned long long mkp_tok_window(const uint32_t* inw, uint32_t q) {
  const uint32_t di = (q >> 5) & 255u, sh = q & 31u;
  const uint32_t a0 = inw[di], a1 = inw[(di + 1u) & 255u], a2 = inw[(di + 2u) & 255u];
  unsigned long long r = ((unsigned long long)a0 | ((unsigned long long)a1 << 32)) >> sh;
  if (sh) r |= (unsigned long long)a2 << (64u - sh);
  return r;
}

// lit: 2^11 entries, [0:3] code length (0: none this short), [4:12] symbol; dist: 2^9 entries, [0:3] code length, [4:8] symbol
MKP_TOK_HD MkpTok mkp_tok_decode(unsigned long long bits, const uint16_t* lit, const uint16_t* dist) {
  MkpTok t; t.a = 0; t.b = 0;
  const uint32_t e = lit[(uint32_t)bits & 2047u], l = e & 15u, sym = e >> 4;
  if (!l || sym == 256u) return t;
  if (sym < 256u) { t.a = l | MKP_TA_WIN | MKP_TA_LIT | (1u << 8); t.b = MKP_SV_LITERAL | sym; return t; }
  const int ls = (int)sym - 257;
  if (ls >= 29) return t;
  const uint32_t ex = len_extra(ls);
  const uint32_t len = len_base(ls) + ((uint32_t)(bits >> l) & ((1u << ex) - 1u));
  uint32_t n = l + ex;
  const uint32_t d = dist[(uint32_t)(bits >> n) & 511u], dl = d & 15u; const int ds = (int)(d >> 4);
  if (!dl || ds >= 30) return t;
  const uint32_t dx = dist_extra(ds);
  t.b = dist_base(ds) + ((uint32_t)(bits >> (n + dl)) & ((1u << dx) - 1u));
  n += dl + dx;   // <= 11 + 5 + 9 + 13 = 38
  if (len > 64u || t.b < len) return t;
  t.a = n | MKP_TA_WIN | (len << 8);
  return t;
}
}  // namespace
