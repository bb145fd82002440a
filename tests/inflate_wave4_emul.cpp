// CPU emulation of mkp_inflate_wave4 (modkit_amd/csrc/mkp_inflate_wave4.hip): the kernel's control flow restated over 64 emulated lanes — the
// chain walk, the prefix sum of output lengths, the accepted prefix, the head slots, the covering-token search, the in-pass pointer jumping,
// the ring and its far reads — around the per-lane functions the kernel itself compiles (mkp_inflate_tok.hpp: window, token decode, byte source).
// Test infrastructure: checks the algorithm against zlib where no GPU is at hand.
//   inflate_wave4_emul bgzf FILE...      every BGZF block of the files: output and acceptance must equal zlib's
//   inflate_wave4_emul corpus FILE       records of [u32 in_len][u32 out_len][in bytes]: raw DEFLATE streams; acceptance (and output when
//                                        accepted) must equal zlib's with the record's out_len as the expected size
// RING=<bytes> in the environment picks the ring size (default 4096, the kernel's); far reads are on below 32768.
// Prints "ok <blocks> <bytes> <accepted> <rejected>"; exits 1 at the first difference.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../modkit_amd/csrc/mkp_inflate_tok.hpp"

namespace {
constexpr uint32_t RING_MAX = 32768u, LIT_BITS = MKP_W4_LIT_BITS, DIST_BITS = MKP_W4_DIST_BITS, INW = MKP_W4_INW;
constexpr int W = 64;
uint32_t RING = 4096u; bool FARM = true;

struct Lds {
  uint8_t ring[RING_MAX];
  uint16_t lit[1u << LIT_BITS], dist[1u << DIST_BITS];
  uint16_t lcount[16], dcount[16], lsym[288], dsym[32];
  uint8_t lens[320];
  uint32_t hd[64];
  uint32_t inw[INW];
};

// mkp_inflate_wave4.hip: build()
template <class ENC>
int build(const uint8_t* lens, int n, uint16_t* tab, uint32_t tab_bits, uint16_t* count, uint16_t* syms, ENC enc) {
  for (int i = 0; i < 16; i++) count[i] = 0;
  for (uint32_t i = 0; i < (1u << tab_bits); i++) tab[i] = 0;
  for (int s = 0; s < n; s++) if (lens[s]) count[lens[s]]++;
  uint32_t next_code[16], offs[16]; int left = 1; uint32_t code = 0, off = 0, used = 0;
  next_code[0] = 0; offs[0] = 0;
  for (int l = 1; l <= 15; l++) { const uint32_t c = count[l]; left = left * 2 - (int)c; code = (code + (l > 1 ? count[l - 1] : 0u)) << 1;
    next_code[l] = code; offs[l] = off; off += c; used += c; }
  if (left < 0) return left;
  if (used == 0) return 0;
  for (int s = 0; s < n; s++) {
    const uint32_t l = lens[s]; if (!l) continue;
    const uint32_t c = next_code[l]++, o = offs[l]++;
    syms[o] = (uint16_t)s;
    if (l <= tab_bits) {
      uint32_t rev = 0; for (uint32_t k = 0; k < l; k++) if (c & (1u << k)) rev |= 1u << (l - 1 - k);
      const uint16_t ent = enc(l, (uint32_t)s);
      for (uint32_t k = rev; k < (1u << tab_bits); k += 1u << l) tab[k] = ent;
    }
  }
  return left;
}
uint32_t cl_order(int i) { static const uint8_t o[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15}; return o[i]; }

struct In2 {
  const uint8_t* p; uint32_t n, lo; uint32_t pf[W]; uint32_t* w;
  uint64_t refills = 0, seeks = 0;
  uint32_t load_word(uint32_t off) const { uint32_t v = 0; for (uint32_t k = 0; k < 4u; k++) if (off + k < n) v |= (uint32_t)p[off + k] << (8u * k);
    return v; }
  void seek(uint32_t byte) {
    lo = byte & ~255u; seeks++;
    for (uint32_t k = 0; k < INW / 64u; k++) for (uint32_t lane = 0; lane < W; lane++) w[(((lo >> 2) + 64u * k) + lane) & (INW - 1u)] = load_word(lo + 256u * k + 4u * lane);
    for (uint32_t lane = 0; lane < W; lane++) pf[lane] = load_word(lo + 4u * INW + 4u * lane);
  }
  void ensure(uint32_t pos_bits) {
    const uint32_t byte = pos_bits >> 3;
    if (byte >= lo + 2048u) { seek(byte); return; }
    while (byte >= lo + 256u) {
      const uint32_t base = (lo >> 2) & (INW - 1u); refills++;
      for (uint32_t lane = 0; lane < W; lane++) w[base + lane] = pf[lane];
      lo += 256u;
      for (uint32_t lane = 0; lane < W; lane++) pf[lane] = load_word(lo + 4u * INW + 4u * lane);
    }
  }
  unsigned long long peek(uint32_t pos_bits) const {
    // the window must hold what is read: bytes [4 * (q >> 5), + 12)
    const uint32_t b0 = 4u * (pos_bits >> 5);
    if (b0 < lo || b0 + 12u > lo + 4u * INW) { fprintf(stderr, "window miss: byte %u, window [%u, %u)\n", b0, lo, lo + 4u * INW); exit(3); }
    uint32_t a, b; mkp_tok_window2(w, pos_bits, &a, &b); return (unsigned long long)a | ((unsigned long long)b << 32);
  }
};
struct Hdr {
  unsigned long long cb; uint32_t cpos;
  void load(In2& in, uint32_t pos) { in.ensure(pos); cb = in.peek(pos); cpos = pos; }
  uint32_t get(In2& in, uint32_t& pos, uint32_t k) { if (pos + k > cpos + 64u) load(in, pos);
    const uint32_t v = (uint32_t)(cb >> (pos - cpos)) & ((1u << k) - 1u); pos += k; return v; }
  uint32_t peek16(In2& in, uint32_t pos) { if (pos + 16u > cpos + 64u) load(in, pos); return (uint32_t)(cb >> (pos - cpos)) & 0xffffu; }
};
int canon_sym(uint32_t bits, const uint16_t* count, const uint16_t* syms, uint32_t* l) {
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= 15; len++) {
    code |= (int)(bits & 1u); bits >>= 1;
    const int c = (int)count[len];
    if (code - c < first) { *l = (uint32_t)len; return (int)syms[index + (code - first)]; }
    index += c; first += c; first <<= 1; code <<= 1;
  }
  return -1;
}
struct OneTok { uint32_t err, bits, kind, val, dist; };
OneTok one_token(const In2& in, const Lds& L, uint32_t q) {
  OneTok r; r.err = 0; r.bits = 0; r.kind = MKP_TK_EOB; r.val = 0; r.dist = 0;
  const unsigned long long bits = in.peek(q);
  const uint32_t e = L.lit[(uint32_t)bits & ((1u << LIT_BITS) - 1u)]; uint32_t l = e & 15u, ex, base;
  if (l && !(e & 16u)) { r.bits = l; r.kind = MKP_TK_LIT; r.val = (e >> 5) & 255u; return r; }
  if (l && (e >> 13) != 7u) { ex = e >> 13; base = ((e >> 5) & 255u) + 3u; }
  else {
    const int sym = canon_sym((uint32_t)bits, L.lcount, L.lsym, &l); if (sym < 0) { r.err = 4u; return r; }
    r.bits = l;
    if (sym < 256) { r.kind = MKP_TK_LIT; r.val = (uint32_t)sym; return r; }
    if (sym == 256) return r;
    const int ls = sym - 257;
    if (ls >= 29) { r.err = 4u; return r; }
    ex = len_extra(ls); base = len_base(ls);
  }
  r.kind = MKP_TK_MATCH; r.val = base + ((uint32_t)(bits >> l) & ((1u << ex) - 1u));
  const uint32_t n = l + ex;
  const uint32_t d = L.dist[(uint32_t)(bits >> n) & ((1u << DIST_BITS) - 1u)]; uint32_t dl = d & 15u; int ds = (int)((d >> 4) & 31u);
  if (!dl) { ds = canon_sym((uint32_t)(bits >> n), L.dcount, L.dsym, &dl); if (ds < 0) { r.err = 4u; return r; } }
  if (ds >= 30) { r.err = 4u; return r; }
  const uint32_t dx = dist_extra(ds);
  r.dist = dist_base(ds) + ((uint32_t)(bits >> (n + dl)) & ((1u << dx) - 1u));
  r.bits = n + dl + dx;
  return r;
}

struct Stats { uint64_t far_reads = 0, far_passes = 0, out_passes = 0, out_bytes = 0, passes = 0, tokens = 0, slow = 0, slow_eob = 0, slow_lit = 0,
    slow_match = 0, slow_long = 0, slow_direct = 0, cut_full = 0, cut_other = 0, jump_passes = 0, jump_rounds = 0, refills = 0, seeks = 0; } g_stats;

// the kernel, one block; returns the status, fills `out` (cap bytes)
uint32_t wave4_block(const uint8_t* inp, uint32_t in_len, uint8_t* o, uint32_t cap) {
  static Lds L;
  In2 in; in.p = inp; in.n = in_len; in.w = L.inw; in.seek(0);
  Hdr h; uint32_t pos = 0, w = 0, err = 0, flushed = 0;
  const uint32_t in_bits = 8u * in_len;
  constexpr uint32_t LITERAL = MKP_SV_LITERAL;
  const uint32_t M = RING - 1u, FLQ = RING >= 32768u ? RING / 2u : RING / 4u, NEAR = RING - 128u;
  auto far_byte = [&](uint32_t at) -> uint32_t {   // the flushed output: what has not been flushed is not there yet
    if (at >= flushed) { fprintf(stderr, "far read of byte %u, flushed %u, w %u\n", at, flushed, w); exit(3); }
    g_stats.far_reads++; return o[at]; };
  auto ring_read = [&](uint32_t at) -> uint8_t {   // an output byte through the ring: it must still be there
    return L.ring[at & M]; };
  auto flush = [&](uint32_t from, uint32_t to) { for (uint32_t k = from; k < to; k++) o[k] = L.ring[k & M]; };
  for (uint32_t guard = 0; guard <= in_len && !err; guard++) {
    h.load(in, pos);
    const uint32_t last = h.get(in, pos, 1), type = h.get(in, pos, 2);
    if (type == 0) {
      pos = (pos + 7u) & ~7u;
      const uint32_t len = h.get(in, pos, 16), nlen = h.get(in, pos, 16);
      if ((len ^ 0xffffu) != nlen || w + len > cap) { err = 2; break; }
      const uint32_t at = pos >> 3;
      if ((unsigned long long)at + len > in_len) { err = 1; break; }
      flush(flushed, w);
      for (uint32_t k = 0; k < len; k++) { const uint8_t v = inp[at + k]; o[w + k] = v; L.ring[(w + k) & M] = v; }
      w += len; flushed = w; pos = 8u * (at + len);
    } else if (type == 1 || type == 2) {
      int nlen_codes = 288, ndist_codes = 30;
      if (type == 1) {
        for (int s = 0; s < 288; s++) L.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
        for (int s = 0; s < 32; s++) L.lens[288 + s] = 5;
        ndist_codes = 32;
      } else {
        const int nlen = (int)h.get(in, pos, 5) + 257, ndist = (int)h.get(in, pos, 5) + 1, ncode = (int)h.get(in, pos, 4) + 4;
        if (nlen > 286 || ndist > 30) { err = 3; break; }
        for (int k = 0; k < 19; k++) L.lens[k] = 0;
        for (int idx = 0; idx < ncode; idx++) { const uint32_t v = h.get(in, pos, 3); L.lens[cl_order(idx)] = (uint8_t)v; }
        if (build(L.lens, 19, L.dist, DIST_BITS, L.dcount, L.dsym, mkp_w4_plain_entry) != 0) { err = 3; break; }
        int idx = 0;
        while (idx < nlen + ndist) {
          const uint32_t e = L.dist[h.peek16(in, pos) & ((1u << DIST_BITS) - 1u)];
          if (!(e & 15u)) { err = 4; break; }
          pos += (e & 15u);
          const int sym = (int)(e >> 4);
          if (sym < 16) { L.lens[idx] = (uint8_t)sym; idx++; }
          else {
            int len = 0, rep;
            if (sym == 16) { if (idx == 0) { err = 3; break; } len = L.lens[idx - 1]; rep = 3 + (int)h.get(in, pos, 2); }
            else if (sym == 17) rep = 3 + (int)h.get(in, pos, 3);
            else rep = 11 + (int)h.get(in, pos, 7);
            if (idx + rep > nlen + ndist) { err = 3; break; }
            for (int k = 0; k < rep; k++) L.lens[idx + k] = (uint8_t)len;
            idx += rep;
          }
        }
        if (err) break;
        uint8_t mine[32]; for (int k = 0; k < 32; k++) mine[k] = k < ndist ? L.lens[nlen + k] : 0;
        for (int k = 0; k < ndist; k++) L.lens[288 + k] = mine[k];
        nlen_codes = nlen; ndist_codes = ndist;
        if (L.lens[256] == 0u) { err = 3; break; }
      }
      {
        const int e1 = build(L.lens, nlen_codes, L.lit, LIT_BITS, L.lcount, L.lsym, mkp_w4_lit_entry);
        { uint32_t used1 = 0; for (int l = 1; l <= 15; l++) used1 += L.lcount[l];
          if (e1 < 0 || (e1 > 0 && !(used1 == 1u && L.lcount[1] == 1u))) { err = 3; break; } }
        const int e2 = build(L.lens + 288, ndist_codes, L.dist, DIST_BITS, L.dcount, L.dsym, mkp_w4_dist_entry);
        uint32_t used2 = 0; for (int l = 1; l <= 15; l++) used2 += L.dcount[l];
        if (e2 < 0 || (e2 > 0 && !(used2 == 1u && L.dcount[1] == 1u))) { err = 3; break; }
      }
      bool eob = false;
      while (!eob && !err) {
        if (pos > in_bits + 64u) { err = 1; break; }
        in.ensure(pos);
        MkpTok4 t[W];
        for (uint32_t lane = 0; lane < W; lane++) {
          const uint32_t b0 = 4u * ((pos + lane) >> 5);   // the window must hold what is read: bytes [4 * (q >> 5), + 12)
          if (b0 < in.lo || b0 + 12u > in.lo + 4u * INW) { fprintf(stderr, "window miss: byte %u, window [%u, %u)\n", b0, in.lo, in.lo + 4u * INW);
            exit(3); }
          uint32_t lo, hi; mkp_tok_window2(L.inw, pos + lane, &lo, &hi);
          t[lane] = mkp_tok_decode4(lo, hi, L.lit, L.dist);
        }
        g_stats.passes++;
        // the walk
        uint32_t i = 0; unsigned long long chain = 0;
        do { chain |= 1ull << i; i += t[i].nx; } while (i < 64u);
        uint32_t stop_nx = 0;
        if (i >= MKP_NX_STOP) { i = 63u - (uint32_t)__builtin_clzll(chain); chain &= ~(1ull << i); stop_nx = t[i].nx; }
        // prefix sum of the output lengths over the chain
        uint32_t ol[W], incl[W], excl[W]; uint32_t run = 0;
        for (uint32_t lane = 0; lane < W; lane++) { ol[lane] = ((chain >> lane) & 1ull) ? t[lane].ol : 0u; run += ol[lane]; incl[lane] = run;
          excl[lane] = run - ol[lane]; }
        unsigned long long rej = 0; bool rej_full = false;
        for (uint32_t lane = 0; lane < W; lane++) {
          const bool ok = incl[lane] <= 64u && w + incl[lane] <= cap && ((t[lane].desc & LITERAL) || t[lane].desc <= w + excl[lane]);
          if (((chain >> lane) & 1ull) && !ok) { if (!rej) rej_full = incl[lane] > 64u; rej |= 1ull << lane; }
        }
        unsigned long long acc = chain; uint32_t adv = i, n_out; bool special;
        if (rej) { const uint32_t first = (uint32_t)__builtin_ctzll(rej); acc = chain & ((1ull << first) - 1ull); adv = first; n_out = excl[first];
          special = first == 0u; if (rej_full) g_stats.cut_full++; else g_stats.cut_other++; }
        else { n_out = incl[63]; special = i < 64u; }
        g_stats.tokens += (uint64_t)__builtin_popcountll(acc);
        if (n_out) {
          g_stats.out_passes++; g_stats.out_bytes += n_out;
          for (uint32_t lane = 0; lane < W; lane++) L.hd[lane] = 0;
          for (uint32_t lane = 0; lane < W; lane++) if ((acc >> lane) & 1ull) { if (L.hd[excl[lane]]) {
              fprintf(stderr, "head slot %u taken twice\n", excl[lane]); exit(3); } L.hd[excl[lane]] = t[lane].desc; }
          unsigned long long heads = 0; for (uint32_t lane = 0; lane < W; lane++) if (L.hd[lane]) heads |= 1ull << lane;
          if (!(heads & 1ull)) { fprintf(stderr, "no head at lane 0\n"); exit(3); }
          uint32_t sv[W];
          for (uint32_t lane = 0; lane < W; lane++) {
            const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
            const uint32_t hj = 63u - (uint32_t)__builtin_clzll(heads & le);
            sv[lane] = lane < n_out ? mkp_w4_source(L.hd[hj], lane, w, M, NEAR, FARM) : LITERAL;
          }
          uint32_t rounds = 0;
          for (;;) {
            bool any = false; for (uint32_t lane = 0; lane < W; lane++) any |= (sv[lane] & (LITERAL | MKP_SV_INPASS)) == MKP_SV_INPASS;
            if (!any) break;
            if (++rounds > 7) { fprintf(stderr, "in-pass references do not resolve\n"); exit(3); }
            uint32_t other[W]; for (uint32_t lane = 0; lane < W; lane++) other[lane] = sv[sv[lane] & 63u];
            for (uint32_t lane = 0; lane < W; lane++) if ((sv[lane] & (LITERAL | MKP_SV_INPASS)) == MKP_SV_INPASS) {
              if ((sv[lane] & 63u) >= lane) { fprintf(stderr, "in-pass reference forward\n"); exit(3); }
              sv[lane] = other[lane];
            }
          }
          if (rounds) { g_stats.jump_passes++; g_stats.jump_rounds += rounds; }
          uint32_t r[W]; bool anyfar = false;
          for (uint32_t lane = 0; lane < W; lane++) r[lane] = L.ring[sv[lane] & M];
          for (uint32_t lane = 0; lane < n_out; lane++) {
            if ((sv[lane] & (LITERAL | MKP_SV_FAR)) == MKP_SV_FAR) { r[lane] = far_byte(sv[lane] & 0xfffffu); anyfar = true; }
            else if (!(sv[lane] & LITERAL)) {   // a ring read: the position it names must be a byte already written and not yet overwritten
              // the lane knows the ring index only; what it stands for is w + lane - dist of its token, checked through the result below (zlib
              // comparison)
            }
          }
          if (anyfar) g_stats.far_passes++;
          for (uint32_t lane = 0; lane < n_out; lane++) L.ring[(w + lane) & M] = (uint8_t)((sv[lane] & LITERAL) ? sv[lane] : r[lane]);
          w += n_out;
        }
        pos += adv;
        if (special) {
          g_stats.slow++; g_stats.tokens++;
          OneTok k;
          if (!rej && (stop_nx & (MKP_NX_STOP - 1u))) {
            k.err = 0; k.kind = MKP_TK_MATCH; k.bits = stop_nx & (MKP_NX_STOP - 1u); k.val = t[adv].ol; k.dist = t[adv].desc; g_stats.slow_direct++;
            const OneTok chk = one_token(in, L, pos);   // (the lane's answer must be the wave's)
            if (chk.err || chk.kind != k.kind || chk.bits != k.bits || chk.val != k.val || chk.dist != k.dist) {
              fprintf(stderr, "lane-decoded long match differs from one_token\n"); exit(3); }
          } else k = one_token(in, L, pos);
          if (k.err) { err = k.err; break; }
          pos += k.bits;
          if (k.kind == MKP_TK_EOB) { eob = true; g_stats.slow_eob++; }
          else if (k.kind == MKP_TK_LIT) {
            g_stats.slow_lit++;
            if (w >= cap) { err = 6; break; }
            L.ring[w & M] = (uint8_t)k.val;
            w++;
          } else {
            const uint32_t len = k.val, dist = k.dist; g_stats.slow_match++; if (len > 64u) g_stats.slow_long++;
            if (dist > w) { err = 5; break; }
            if (w + len > cap) { err = 6; break; }
            const uint32_t src0 = w - dist;
            for (uint32_t k0 = 0; k0 < len; k0 += 64u) {   // lanes in steps of 64: all loads of a step before its stores
              uint8_t v[W];
              for (uint32_t k2 = k0; k2 < k0 + 64u && k2 < len; k2++) {
                // the kernel's walk of the source cycle: lane mod dist, then steps of 64 mod dist
                uint32_t r = (k2 - k0) % dist; const uint32_t step = 64u % dist;
                for (uint32_t q = 0; q < k0; q += 64u) { r += step; r -= r >= dist ? dist : 0u; }
                if (r != k2 % dist) { fprintf(stderr, "cycle walk off\n"); exit(3); }
                v[k2 - k0] = (FARM && dist > NEAR) ? (uint8_t)far_byte(src0 + k2) : dist >= len ? L.ring[(src0 + k2) & M] : dist == 1u
                    ? L.ring[src0 & M] : L.ring[(src0 + r) & M];
              }
              for (uint32_t k2 = k0; k2 < k0 + 64u && k2 < len; k2++) L.ring[(w + k2) & M] = v[k2 - k0];
            }
            w += len;
          }
        }
        if ((w & ~(FLQ - 1u)) > flushed) { const uint32_t upto = w & ~(FLQ - 1u); flush(flushed, upto); flushed = upto; }
        if (w - flushed > RING) { fprintf(stderr, "ring overrun: %u bytes unflushed\n", w - flushed); exit(3); }
        if (FARM && w - flushed + 64u + 258u > NEAR) { fprintf(stderr, "unflushed tail %u reaches the far zone\n", w - flushed); exit(3); }
      }
    } else { err = 2; break; }
    if (err || last) break;
  }
  flush(flushed, w);
  if (!err && w != cap) err = 6;
  if (!err && pos > in_bits) err = 1;
  g_stats.refills += in.refills; g_stats.seeks += in.seeks;
  (void)ring_read;
  return err;
}

// zlib on a raw DEFLATE stream that must produce exactly `cap` bytes and end inside the input: 0 accepted
int zlib_block(const uint8_t* inp, uint32_t in_len, uint8_t* o, uint32_t cap) {
  z_stream z; memset(&z, 0, sizeof(z));
  if (inflateInit2(&z, -15) != Z_OK) return -100;
  std::vector<uint8_t> spill(16);
  z.next_in = const_cast<uint8_t*>(inp); z.avail_in = in_len; z.next_out = o; z.avail_out = cap;
  int rc = inflate(&z, Z_FINISH);
  if (rc == Z_BUF_ERROR && z.avail_out == 0) {   // the expected size is full: a stream that goes on is a size mismatch
    z.next_out = spill.data(); z.avail_out = (uInt)spill.size(); rc = inflate(&z, Z_FINISH);
    if (rc == Z_STREAM_END && z.total_out == cap) { inflateEnd(&z); return 0; }
    inflateEnd(&z); return 1;
  }
  const bool ok = rc == Z_STREAM_END && z.total_out == cap;
  inflateEnd(&z);
  return ok ? 0 : 1;
}

std::vector<uint8_t> slurp(const char* path) {
  FILE* f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
  std::vector<uint8_t> v; uint8_t buf[1 << 16]; size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f); return v;
}
}  // namespace

int main(int argc, char** argv) {
  if (getenv("RING")) { RING = (uint32_t)atoi(getenv("RING")); FARM = RING < 32768u; if (RING < 1024u || RING > RING_MAX || (RING & (RING - 1u))) {
      fprintf(stderr, "RING must be a power of two in [1024, 32768]\n"); return 2; } }
  if (argc < 3) { fprintf(stderr, "usage: inflate_wave4_emul bgzf FILE... | corpus FILE\n"); return 2; }
  const std::string mode = argv[1];
  uint64_t blocks = 0, bytes = 0, accepted = 0, rejected = 0;
  auto one = [&](const uint8_t* inp, uint32_t in_len, uint32_t cap, const char* what, uint64_t at) {
    std::vector<uint8_t> a(cap + 64, 0xAA), b(cap + 64, 0xBB);
    const uint32_t st = wave4_block(inp, in_len, a.data(), cap);
    const int zr = zlib_block(inp, in_len, b.data(), cap);
    if ((st == 0) != (zr == 0)) {
      fprintf(stderr, "%s @%llu: kernel status %u, zlib %s (in %u bytes, out %u)\n", what, (unsigned long long)at, st,
          zr == 0 ? "accepts" : "rejects", in_len, cap);
      exit(1); }
    if (st == 0 && memcmp(a.data(), b.data(), cap) != 0) {
      uint32_t k = 0; while (k < cap && a[k] == b[k]) k++;
      fprintf(stderr, "%s @%llu: output differs at byte %u of %u\n", what, (unsigned long long)at, k, cap); exit(1);
    }
    blocks++; bytes += cap; if (st == 0) accepted++; else rejected++;
  };
  if (mode == "bgzf") {
    for (int f = 2; f < argc; f++) {
      const std::vector<uint8_t> d = slurp(argv[f]);
      size_t off = 0;
      while (off + 18 <= d.size()) {
        if (d[off] != 0x1f || d[off + 1] != 0x8b) { fprintf(stderr, "%s: not BGZF at %zu\n", argv[f], off); return 2; }
        const uint32_t xlen = d[off + 10] | (d[off + 11] << 8);
        uint32_t bsize = 0; for (uint32_t x = 0; x + 4 <= xlen;) { const uint8_t* e = &d[off + 12 + x]; const uint32_t sl = e[2] | (e[3] << 8);
          if (e[0] == 'B' && e[1] == 'C') bsize = (e[4] | (e[5] << 8)) + 1u;
          x += 4 + sl; }
        if (!bsize || off + bsize > d.size()) { fprintf(stderr, "%s: bad block at %zu\n", argv[f], off); return 2; }
        const uint32_t hdr = 12 + xlen, clen = bsize - hdr - 8;
        uint32_t isize; memcpy(&isize, &d[off + bsize - 4], 4);
        one(&d[off + hdr], clen, isize, argv[f], off);
        off += bsize;
      }
    }
  } else if (mode == "corpus") {
    const std::vector<uint8_t> d = slurp(argv[2]);
    size_t off = 0; uint64_t rec = 0;
    while (off + 8 <= d.size()) {
      uint32_t in_len, out_len; memcpy(&in_len, &d[off], 4); memcpy(&out_len, &d[off + 4], 4); off += 8;
      if (off + in_len > d.size()) { fprintf(stderr, "corpus truncated\n"); return 2; }
      one(&d[off], in_len, out_len, "record", rec++);
      off += in_len;
    }
  } else return 2;
  printf("ok %llu %llu %llu %llu\n", (unsigned long long)blocks, (unsigned long long)bytes, (unsigned long long)accepted,
      (unsigned long long)rejected);
  fprintf(stderr,
      "ring %u: far reads %llu in %llu passes; passes %llu (%.2f tokens, %.1f bytes each; %llu with output) cut by the 64-byte limit %llu, by size / distance %llu; in-pass jumps in %llu passes (%.2f rounds); special %llu (eob %llu lit %llu match %llu of which > 64 bytes %llu, lane-decoded %llu) refills %llu seeks %llu\n",
          RING, (unsigned long long)g_stats.far_reads, (unsigned long long)g_stats.far_passes, (unsigned long long)g_stats.passes,
              g_stats.passes ? (double)g_stats.tokens / (double)g_stats.passes : 0.0,
          g_stats.passes ? (double)g_stats.out_bytes / (double)g_stats.passes : 0.0, (unsigned long long)g_stats.out_passes,
              (unsigned long long)g_stats.cut_full, (unsigned long long)g_stats.cut_other,
          (unsigned long long)g_stats.jump_passes, g_stats.jump_passes ? (double)g_stats.jump_rounds / (double)g_stats.jump_passes : 0.0,
              (unsigned long long)g_stats.slow, (unsigned long long)g_stats.slow_eob,
          (unsigned long long)g_stats.slow_lit, (unsigned long long)g_stats.slow_match, (unsigned long long)g_stats.slow_long,
              (unsigned long long)g_stats.slow_direct, (unsigned long long)g_stats.refills, (unsigned long long)g_stats.seeks);
  return 0;
}
