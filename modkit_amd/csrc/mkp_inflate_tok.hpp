// Per-lane pieces of the speculative wave inflate (mkp_inflate_wave4.hip): what a position of a DEFLATE bit stream decodes to IF a
// literal/length code starts there.  Every lane of a wave decodes the position `bit + lane`; which of the 64 answers are real — the
// chain 0 -> n(0) -> n(0) + n(n(0)) ... — is settled afterwards by a scalar walk.  No cross-lane operation in here: the file also
// compiles for the host, where tests/inflate_wave4_emul.cpp runs the same functions over 64 emulated lanes against zlib.
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
MKP_TOK_HD uint32_t len_base(int ls) {
  return ls < 8 ? 3u + (uint32_t)ls : ls == 28 ? 258u : 3u + ((4u + ((uint32_t)ls & 3u)) << ((uint32_t)(ls >> 2) - 1u)); }
MKP_TOK_HD uint32_t dist_extra(int ds) { return ds < 4 ? 0u : (uint32_t)(ds >> 1) - 1u; }
MKP_TOK_HD uint32_t dist_base(int ds) { return ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1u)) << ((uint32_t)(ds >> 1) - 1u)); }

enum : uint32_t { MKP_SV_LITERAL = 0x80000000u };
enum : uint32_t { MKP_TK_LIT = 0u, MKP_TK_MATCH = 1u, MKP_TK_EOB = 2u };
}  // namespace

// ---- the chain is only MARKED by the scalar walk; everything a token does to the output is done by all lanes at once
namespace {
// The direct tables carry what a lane needs without a second probe (the wave is latency-bound: every dependent LDS round trip is ~120 cycles).
//   literal/length, 2^11 x u16, by the next 11 stream bits: [0:3] code length (0: no code this short), [4] 0 = literal, 1 = length symbol
//     or stop, [5:12] the literal byte | base length - 3, [13:15] extra bits of the length (0..5), 7 = stop: end of block, or symbols 286 / 287
//   distance, 2^8 x u16: [0:3] code length (0: no code this short, or symbols 30 / 31), [4:8] symbol, [9:12] its extra bits
constexpr uint32_t MKP_W4_LIT_BITS = 11u, MKP_W4_DIST_BITS = 8u, MKP_W4_INW = 128u;   // (input window: 128 dwords, circular)
MKP_TOK_HD uint16_t mkp_w4_lit_entry(uint32_t l, uint32_t sym) {
  if (sym < 256u) return (uint16_t)(l | (sym << 5));
  if (sym == 256u || sym >= 286u) return (uint16_t)(l | 16u | (7u << 13));
  const int ls = (int)sym - 257;
  return (uint16_t)(l | 16u | ((len_base(ls) - 3u) << 5) | (len_extra(ls) << 13));
}
MKP_TOK_HD uint16_t mkp_w4_dist_entry(uint32_t l, uint32_t ds) {
  return ds >= 30u ? (uint16_t)0 : (uint16_t)(l | (ds << 4) | (dist_extra((int)ds) << 9)); }
MKP_TOK_HD uint16_t mkp_w4_plain_entry(uint32_t l, uint32_t sym) { return (uint16_t)(l | (sym << 4)); }   // (the code-length code)

// What lane k reports for bit position pos + k: nx = the bits the token takes if the pass can place it as it is (a literal, or a match of at
// most 64 bytes — self-overlapping ones included: the in-pass references resolve them); otherwise MKP_NX_STOP (end of block, a code longer
// than the direct tables, an invalid code: the walk stops there and the wave decodes that token on its own) or MKP_NX_STOP | bits (a longer
// match the direct tables decode: the walk stops, the lane's answer stands); ol = its output bytes; desc = MKP_SV_LITERAL | byte, or the distance.
struct MkpTok4 { uint32_t nx, ol, desc; };
// nx of a position the walk stops at: any chain offset + 128 is told from any in-range advance (<= 63 + 38)
enum : uint32_t { MKP_NX_STOP = 128u };
// (hi:lo) >> sh, low 32 bits, sh in [0, 31]: v_alignbit_b32
MKP_TOK_HD uint32_t mkp_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
  return (uint32_t)((((unsigned long long)hi << 32) | lo) >> (sh & 31u));
#endif
}
MKP_TOK_HD uint32_t mkp_bfe(uint32_t v, uint32_t off, uint32_t width) {   // width in [0, 31]; off + width <= 32
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ubfe(v, off, width);
#else
  return width ? (v >> off) & ((1u << width) - 1u) : 0u;
#endif
}
// 70 stream bits from bit q on as (lo, hi) of the first 64: three dwords of the circular input window, two funnel shifts (no branch on q & 31)
MKP_TOK_HD void mkp_tok_window2(const uint32_t* inw, uint32_t q, uint32_t* lo, uint32_t* hi) {
  const uint32_t di = (q >> 5) & (MKP_W4_INW - 1u), sh = q & 31u;
  const uint32_t a0 = inw[di], a1 = inw[(di + 1u) & (MKP_W4_INW - 1u)], a2 = inw[(di + 2u) & (MKP_W4_INW - 1u)];
  *lo = mkp_alignbit(a1, a0, sh); *hi = mkp_alignbit(a2, a1, sh);
}
MKP_TOK_HD MkpTok4 mkp_tok_decode4(uint32_t lo, uint32_t hi, const uint16_t* lit, const uint16_t* dist) {
  const uint32_t e = lit[lo & ((1u << MKP_W4_LIT_BITS) - 1u)], l = e & 15u, ex = e >> 13, val = (e >> 5) & 255u;
  const uint32_t len = val + 3u + mkp_bfe(lo, l, ex);   // l + ex <= 11 + 5 where it matters
  const uint32_t n = l + ex;                             // (<= 18 on a stop entry: still a shift below 32)
  const uint32_t rest = mkp_alignbit(hi, lo, n);         // 32 bits from the distance code on (it needs at most 8 + 13)
  const uint32_t d = dist[rest & ((1u << MKP_W4_DIST_BITS) - 1u)], dl = d & 15u, ds = (d >> 4) & 31u, dx = d >> 9;
  const uint32_t dd = 1u + (ds < 4u ? ds : ((2u | (ds & 1u)) << dx)) + mkp_bfe(rest, dl, dx);   // dist_base(ds) + extra
  const bool is_lit = l != 0u && !(e & 16u);
  const bool is_match = l != 0u && (e & 16u) != 0u && ex != 7u && dl != 0u;   // (a longer match: the walk stops, but the lane's answer stands)
  const uint32_t nb = n + dl + dx;   // <= 16 + 8 + 13 = 37
  MkpTok4 t;
  t.nx = is_lit ? l : !is_match ? MKP_NX_STOP : len <= 64u ? nb : (MKP_NX_STOP | nb);
  t.ol = is_lit ? 1u : len;
  t.desc = is_lit ? (MKP_SV_LITERAL | val) : dd;
  return t;
}

// Where output byte j of a pass (output position w + j) comes from, given the descriptor of the token that covers it: a literal, a byte
// produced `dist` bytes earlier in this same pass (MKP_SV_INPASS | lane), a byte of the flushed output (FARM builds: MKP_SV_FAR | output
// position, for distances beyond `near`), or a ring position.
enum : uint32_t { MKP_SV_FAR = 0x40000000u, MKP_SV_INPASS = 0x20000000u };
MKP_TOK_HD uint32_t mkp_w4_source(uint32_t desc, uint32_t j, uint32_t w, uint32_t ring_mask, uint32_t near, bool farm) {
  const uint32_t back = j - desc, at = w + back;   // (selects, not branches: the lanes of a pass disagree on every one of these)
  const uint32_t ringed = (farm && desc > near) ? (MKP_SV_FAR | at) : (at & ring_mask);
  const uint32_t copied = desc <= j ? (MKP_SV_INPASS | back) : ringed;
  return (desc & MKP_SV_LITERAL) ? desc : copied;
}
}  // namespace
