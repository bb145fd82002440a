// Host-side raw DEFLATE decoder for BGZF blocks (RFC 1951; BGZF: SAM spec §4.1 — every block is one complete deflate stream of at most
// 64 KiB of output whose size is known up front).  The shard loop's host cost is dominated by inflating the BAM (zlib: ~0.4 GB/s per core;
// the GPU boxes give a process 16 cores for a 4.4 GB file), so the blocks go through this decoder first and through zlib only when it
// declines (a stream it considers malformed: zlib then produces the error).
//
// Shape: 64-bit bit buffer refilled 7-8 bytes at a time; canonical-Huffman tables with an 11-bit (literal/length) and 8-bit (distance)
// first level and second-level tables for longer codes, entries carrying the base value and the number of extra bits; a fast loop that
// runs while >= 274 output bytes and >= 16 input bytes remain (up to three literals per refill, matches copied 8 bytes at a time), and a
// bounds-checked loop for the rest.  Knowing the whole input and the exact output size is what makes it simpler than a streaming inflate.
#pragma once
#include <cstdint>
#include <cstring>

namespace mkp {
namespace hostinf {

constexpr uint32_t F_LIT = 1u << 31, F_EOB = 1u << 30, F_SUB = 1u << 29, F_LEN = 1u << 28, F_LIT2 = 1u << 11;
// entry = flags | value << 12 (16 bits) | extra_bits << 8 | code_bits.  A literal entry with F_LIT2 carries two literals (value = first |
// second << 8, code_bits = both codes): packed bases and qualities have 4-6 bit codes, so most first-level slots hold a pair.
constexpr int LIT_BITS = 11, DIST_BITS = 8;
constexpr uint32_t LIT_CAP = 4096, DIST_CAP = 1024;

struct Tables { uint32_t lit[LIT_CAP]; uint32_t dist[DIST_CAP]; };

static inline uint32_t rev_bits(uint32_t c, int n) {   // the low n (<= 16) bits of c, reversed
  c = ((c & 0x5555u) << 1) | ((c >> 1) & 0x5555u); c = ((c & 0x3333u) << 2) | ((c >> 2) & 0x3333u); c = ((c & 0x0f0fu) << 4) | ((c >> 4) & 0x0f0fu);
  return (((c & 0xffu) << 8) | ((c >> 8) & 0xffu)) >> (16 - n);
}

// Canonical code of `n` symbols with lengths lens[] (0 = unused) into a two-level table; sym_entry(s) gives the entry without its
// code_bits.  Returns false for an over-subscribed code, an incomplete literal/length code, or a table that does not fit.
template <class SymEntry>
static inline bool build_table(const uint8_t* lens, int n, int root, uint32_t* tab, uint32_t cap, bool allow_incomplete, SymEntry sym_entry) {
  int count[16] = {0};
  for (int i = 0; i < n; i++) count[lens[i]]++;
  count[0] = 0;
  int left = 1; uint32_t next_code[16]; uint32_t code = 0;
  for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return false; code = (code + (uint32_t)count[l - 1]) << 1;
    next_code[l] = code; }
  const bool incomplete = left > 0;
  if (incomplete) {   // zlib's rule (inftrees.c): only "no codes at all" or "one code of one bit", and never for literal/length
    int longest = 0; for (int l = 1; l <= 15; l++) if (count[l]) longest = l;
    if (!allow_incomplete || longest > 1) return false;
  }
  const uint32_t root_size = 1u << root;
  if (incomplete) memset(tab, 0, sizeof(uint32_t) * cap);   // entry 0 = no such code
  // second-level tables: the longest code behind every first-level prefix
  uint8_t submax[1u << LIT_BITS]; bool any_long = false;
  for (int l = root + 1; l <= 15; l++) if (count[l]) any_long = true;
  uint32_t codes[320];
  for (int s = 0; s < n; s++) if (lens[s]) codes[s] = rev_bits(next_code[lens[s]]++, lens[s]);
  uint32_t used = root_size;
  if (any_long) {
    uint16_t prefixes[320]; int n_prefixes = 0;   // first-level slots that lead to a second-level table (at most one per long code)
    memset(submax, 0, root_size);
    for (int s = 0; s < n; s++) if (lens[s] > root) { const uint32_t p = codes[s] & (root_size - 1);
      if (!submax[p]) prefixes[n_prefixes++] = (uint16_t)p;
      if (lens[s] > submax[p]) submax[p] = lens[s];
      }
    for (int k = 0; k < n_prefixes; k++) {
      const uint32_t p = prefixes[k], sb = (uint32_t)submax[p] - (uint32_t)root;
      if (used + (1u << sb) > cap) return false;
      tab[p] = F_SUB | (used << 12) | (sb << 8) | (uint32_t)root;
      used += 1u << sb;
    }
  }
  for (int s = 0; s < n; s++) {
    const int l = lens[s]; if (!l) continue;
    const uint32_t e = sym_entry(s);
    if (l <= root) {
      const uint32_t ent = e | (uint32_t)l;
      for (uint32_t k = codes[s]; k < root_size; k += 1u << l) tab[k] = ent;
    } else {
      const uint32_t p = codes[s] & (root_size - 1), head = tab[p], off = (head >> 12) & 0xffffu, sb = (head >> 8) & 15u;
      const uint32_t ent = e | (uint32_t)(l - root);
      for (uint32_t k = codes[s] >> root; k < (1u << sb); k += 1u << (l - root)) tab[off + k] = ent;
    }
  }
  return true;
}

// first-level literal slots whose remaining index bits determine a second literal become pair entries.  Slot i of a literal with an
// l1-bit code is code | k << l1; the second symbol is whatever slot k (zero-extended) decodes, when its code fits the 11 - l1 bits k has.
static inline void pair_literals(uint32_t* tab) {
  constexpr uint32_t N = 1u << LIT_BITS;
  uint32_t orig[N]; memcpy(orig, tab, sizeof(orig));
  for (uint32_t i = 0; i < N; i++) {
    const uint32_t e = orig[i];
    if (!(e & F_LIT)) continue;
    const uint32_t l1 = e & 0xffu; if (l1 >= (uint32_t)LIT_BITS) continue;
    const uint32_t e2 = orig[i >> l1];
    if (!(e2 & F_LIT) || l1 + (e2 & 0xffu) > (uint32_t)LIT_BITS) continue;
    tab[i] = F_LIT | F_LIT2 | (((e >> 12) & 0xffu) << 12) | (((e2 >> 12) & 0xffu) << 20) | (l1 + (e2 & 0xffu));
  }
}

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227,
    258};
static const uint8_t LEN_XB[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097,
    6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_XB[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t lit_entry(int s) {
  if (s < 256) return F_LIT | ((uint32_t)s << 12);
  if (s == 256) return F_EOB;
  if (s > 285) return F_LEN | (0xffffu << 12);   // 286/287 take part in the code but may not occur: base 65535 trips the length check
  return F_LEN | ((uint32_t)LEN_BASE[s - 257] << 12) | ((uint32_t)LEN_XB[s - 257] << 8);
}
static inline uint32_t dist_entry(int s) {
  if (s > 29) return F_LEN | (0xffffu << 12);    // 30/31: invalid distances (larger than any window)
  return F_LEN | ((uint32_t)DIST_BASE[s] << 12) | ((uint32_t)DIST_XB[s] << 8);
}

struct Fixed {   // the fixed code of block type 1 (RFC 1951 §3.2.6), built once
  Tables t; bool ok;
  Fixed() {
    uint8_t l[288], d[32];
    for (int i = 0; i < 288; i++) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
    for (int i = 0; i < 32; i++) d[i] = 5;
    ok = build_table(l, 288, LIT_BITS, t.lit, LIT_CAP, false, lit_entry) && build_table(d, 32, DIST_BITS, t.dist, DIST_CAP, false, dist_entry);
    if (ok) pair_literals(t.lit);
  }
};

// src[0, clen) -> dst[0, dlen); the 8 bytes after src[clen) must be readable (BGZF: CRC32 + ISIZE follow the payload).
// true = dst holds exactly dlen bytes of a well-formed stream; false = declined (nothing may be assumed about dst).
static inline bool inflate(const uint8_t* src, size_t clen, uint8_t* dst, size_t dlen) {
  static const Fixed fixed;
  Tables dyn;
  const uint8_t* in = src; const uint8_t* const in_end = src + clen;
  uint8_t* out = dst; uint8_t* const out_end = dst + dlen;
  uint64_t bb = 0; uint32_t bc = 0;   // bit buffer: the low `bc` bits are counted as unread stream bits
  auto refill_slow = [&]() { while (bc <= 56 && in < in_end) { bb |= (uint64_t)*in++ << bc; bc += 8; } };
  auto take = [&](uint32_t nbits, uint32_t* v) -> bool {   // header fields: nbits <= 16
    if (bc < nbits) { refill_slow(); if (bc < nbits) return false; }
    *v = (uint32_t)(bb & ((1ull << nbits) - 1)); bb >>= nbits; bc -= nbits; return true;
  };
  for (;;) {
    uint32_t final_blk, type;
    if (!take(1, &final_blk) || !take(2, &type)) return false;
    const Tables* T = nullptr;
    if (type == 0) {   // stored: skip to the byte boundary, LEN / NLEN, raw bytes
      if (bc < 64) bb &= (1ull << bc) - 1;   // drop what a fast refill loaded beyond the counted bits: bytes are about to be taken from `in` directly
      const uint32_t drop = bc & 7u; bb >>= drop; bc -= drop;
      uint32_t len, nlen; if (!take(16, &len) || !take(16, &nlen) || (len ^ 0xffffu) != nlen) return false;
      // bytes still in the bit buffer first, then straight from the input
      while (len && bc >= 8) { if (out >= out_end) return false; *out++ = (uint8_t)bb; bb >>= 8; bc -= 8; len--; }
      if (len) { if (bc != 0) return false; if ((size_t)(in_end - in) < len || (size_t)(out_end - out) < len) return false; memcpy(out, in, len);
        in += len; out += len; }
      if (final_blk) break; else continue;
    } else if (type == 1) { if (!fixed.ok) return false; T = &fixed.t; }
    else if (type == 2) {
      uint32_t hlit, hdist, hclen;
      if (!take(5, &hlit) || !take(5, &hdist) || !take(4, &hclen)) return false;
      hlit += 257; hdist += 1; hclen += 4;
      if (hlit > 286 || hdist > 30) return false;
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t cl[19] = {0};
      for (uint32_t i = 0; i < hclen; i++) { uint32_t v; if (!take(3, &v)) return false; cl[order[i]] = (uint8_t)v; }
      uint32_t cltab[128];   // 7-bit single-level table of the code-length code
      { int count[8] = {0}; for (int i = 0; i < 19; i++) count[cl[i]]++; count[0] = 0;
        int left = 1; uint32_t next_code[8], code = 0;
        for (int l = 1; l <= 7; l++) { left = (left << 1) - count[l]; if (left < 0) return false; code = (code + (uint32_t)count[l - 1]) << 1;
          next_code[l] = code; }
        if (left > 0) memset(cltab, 0, sizeof(cltab));
        for (int s = 0; s < 19; s++) if (cl[s]) { const uint32_t r = rev_bits(next_code[cl[s]]++, cl[s]);
          for (uint32_t k = r; k < 128; k += 1u << cl[s]) cltab[k] = 0x80000000u | ((uint32_t)s << 8) | cl[s];
          } }
      uint8_t lens[288 + 32]; uint32_t at = 0; const uint32_t total = hlit + hdist;
      while (at < total) {
        if (bc < 14) { refill_slow(); }
        const uint32_t e = cltab[bb & 127u]; if (!e || (e & 0xffu) > bc) return false;
        bb >>= (e & 0xffu); bc -= (e & 0xffu);
        const uint32_t s = (e >> 8) & 0xffu;
        if (s < 16) { lens[at++] = (uint8_t)s; continue; }
        uint32_t rep, v; uint8_t fill = 0;
        if (s == 16) { if (at == 0 || !take(2, &v)) return false; rep = 3 + v; fill = lens[at - 1]; }
        else if (s == 17) { if (!take(3, &v)) return false; rep = 3 + v; }
        else { if (!take(7, &v)) return false; rep = 11 + v; }
        if (at + rep > total) return false;
        memset(lens + at, fill, rep); at += rep;
      }
      if (lens[256] == 0) return false;   // no end-of-block code
      uint8_t ll[288], dl[32]; memcpy(ll, lens, hlit); memset(ll + hlit, 0, 288 - hlit); memcpy(dl, lens + hlit, hdist);
        memset(dl + hdist, 0, 32 - hdist);
      if (!build_table(ll, 288, LIT_BITS, dyn.lit, LIT_CAP, false, lit_entry)) return false;
      pair_literals(dyn.lit);
      if (!build_table(dl, 32, DIST_BITS, dyn.dist, DIST_CAP, true, dist_entry)) return false;   // one or no distance code is a legal incomplete code
      T = &dyn;
    } else return false;

    const uint32_t* const lt = T->lit; const uint32_t* const dt = T->dist;
    constexpr uint64_t LMASK = (1u << LIT_BITS) - 1, DMASK = (1u << DIST_BITS) - 1;
    bool eob = false;
    // ---- fast loop: room for three literal pairs, a maximal match and the copy overshoot; input for two refills.  The table entry of the
    // next symbol is looked up before a match is copied, so that its latency hides behind the copy.
#define MKP_REFILL() do { uint64_t w_; memcpy(&w_, in, 8); bb |= w_ << bc; in += (63u - bc) >> 3; bc |= 56u; } while (0)
#define MKP_PUT_LIT(e) do { bb >>= ((e) & 0xffu); bc -= ((e) & 0xffu); const uint16_t v2 = (uint16_t)((e) >> 12); memcpy(out, &v2, 2); out += 1u + (((e) >> 11) & 1u); } while (0)
    if ((size_t)(out_end - out) >= 288 && (size_t)(in_end - in) >= 16) {
      MKP_REFILL();
      uint32_t e = lt[bb & LMASK];
      for (;;) {   // here: >= 56 counted bits, e = entry of the next symbol
        if (e & F_LIT) {
          MKP_PUT_LIT(e); e = lt[bb & LMASK];
          if (e & F_LIT) { MKP_PUT_LIT(e); e = lt[bb & LMASK];
            if (e & F_LIT) { MKP_PUT_LIT(e); goto next_symbol; } }
          MKP_REFILL();   // a length / end-of-block / long code follows the literals: it needs up to 48 bits (the low bits, and so e, are unchanged)
        }
        if (e & F_SUB) { bb >>= LIT_BITS; bc -= LIT_BITS; e = lt[((e >> 12) & 0xffffu) + (uint32_t)(bb & ((1u << ((e >> 8) & 15u)) - 1u))];
          if (e & F_LIT) { MKP_PUT_LIT(e); goto next_symbol; } }
        if (!e) return false;
        bb >>= (e & 0xffu); bc -= (e & 0xffu);
        if (e & F_EOB) { eob = true; break; }
        {
          const uint32_t lxb = (e >> 8) & 15u; const uint32_t len = ((e >> 12) & 0xffffu) + (uint32_t)(bb & ((1u << lxb) - 1u)); bb >>= lxb;
            bc -= lxb;
          if (len > 258) return false;
          uint32_t d = dt[bb & DMASK];
          if (d & F_SUB) { bb >>= DIST_BITS; bc -= DIST_BITS; d = dt[((d >> 12) & 0xffffu) + (uint32_t)(bb & ((1u << ((d >> 8) & 15u)) - 1u))]; }
          if (!d) return false;
          bb >>= (d & 0xffu); bc -= (d & 0xffu);
          const uint32_t dxb = (d >> 8) & 15u; const uint32_t dist = ((d >> 12) & 0xffffu) + (uint32_t)(bb & ((1u << dxb) - 1u)); bb >>= dxb;
            bc -= dxb;
          if (dist > (size_t)(out - dst) || dist > 32768u) return false;
          MKP_REFILL(); e = lt[bb & LMASK];   // the next symbol's entry is on its way while the match is copied
          const uint8_t* from = out - dist; uint8_t* const stop = out + len;
          // 16 bytes without a test (most matches are shorter: no loop branch to mispredict), word by word so that dist < 16 reads what it just wrote
          if (dist >= 8) {
            uint64_t w; memcpy(&w, from, 8); memcpy(out, &w, 8); memcpy(&w, from + 8, 8); memcpy(out + 8, &w, 8);
            if (len > 16) { from += 16; out += 16; do { memcpy(&w, from, 8); memcpy(out, &w, 8); from += 8; out += 8; } while (out < stop); }
          }
          else if (dist == 1) memset(out, *from, len);
          else { do { *out++ = *from++; } while (out < stop); }
          out = stop;
          if ((size_t)(out_end - out) < 288 || (size_t)(in_end - in) < 16) break;
          continue;
        }
      next_symbol:
        if ((size_t)(out_end - out) < 288 || (size_t)(in_end - in) < 16) break;
        MKP_REFILL(); e = lt[bb & LMASK];
      }
    }
#undef MKP_PUT_LIT
#undef MKP_REFILL
    // ---- careful loop: the last bytes of the block / of the stream
    if (!eob) {
      if (bc < 64) bb &= (1ull << bc) - 1;   // bits loaded by the fast refill beyond the counted ones are re-read from `in`
      for (;;) {
        refill_slow();
        uint32_t e = lt[bb & LMASK];
        if (e & F_SUB) { if (bc < (uint32_t)LIT_BITS) return false; bb >>= LIT_BITS; bc -= LIT_BITS;
          e = lt[((e >> 12) & 0xffffu) + (uint32_t)(bb & ((1u << ((e >> 8) & 15u)) - 1u))]; }
        if (!e || (e & 0xffu) > bc) return false;
        bb >>= (e & 0xffu); bc -= (e & 0xffu);
        if (e & F_LIT) { const uint32_t nl = 1u + ((e >> 11) & 1u); if ((size_t)(out_end - out) < nl) return false; out[0] = (uint8_t)(e >> 12);
          if (nl == 2) out[1] = (uint8_t)(e >> 20);
          out += nl; continue; }
        if (e & F_EOB) break;
        refill_slow();
        const uint32_t lxb = (e >> 8) & 15u; if (bc < lxb) return false;
        const uint32_t len = ((e >> 12) & 0xffffu) + (uint32_t)(bb & ((1u << lxb) - 1u)); bb >>= lxb; bc -= lxb;
        if (len > 258) return false;
        uint32_t d = dt[bb & DMASK];
        if (d & F_SUB) { if (bc < (uint32_t)DIST_BITS) return false; bb >>= DIST_BITS; bc -= DIST_BITS;
          d = dt[((d >> 12) & 0xffffu) + (uint32_t)(bb & ((1u << ((d >> 8) & 15u)) - 1u))]; }
        if (!d || (d & 0xffu) > bc) return false;
        bb >>= (d & 0xffu); bc -= (d & 0xffu);
        refill_slow();
        const uint32_t dxb = (d >> 8) & 15u; if (bc < dxb) return false;
        const uint32_t dist = ((d >> 12) & 0xffffu) + (uint32_t)(bb & ((1u << dxb) - 1u)); bb >>= dxb; bc -= dxb;
        if (dist > (size_t)(out - dst) || dist > 32768u || len > (size_t)(out_end - out)) return false;
        const uint8_t* from = out - dist; for (uint32_t i = 0; i < len; i++) *out++ = *from++;
      }
    }
    if (final_blk) break;
  }
  return out == out_end;
}

}  // namespace hostinf
}  // namespace mkp
