// BGZF block inflate on the device — first stage of SURVEY §8 (f1), "BAM ingest on device".  The reference reads BAM through
// rust-htslib's IndexedReader (src/pileup/mod.rs:732-743); the block format is BGZF (SAM spec 4.1: gzip members of at most 64 KiB
// with a BC extra field) and the payload is raw DEFLATE (RFC 1951), which htslib hands to zlib.  Neither library is part of the
// reference checkout: the decoder below restates RFC 1951 section 3.2 directly (stored / fixed / dynamic blocks, canonical
// Huffman codes decoded length by length as in section 3.2.2).  The length-by-length walk of decode_sym (`code - count < first`, with
// running `first` / `index`) is the canonical-code decoder of Mark Adler's puff.c (zlib's contrib/puff, zlib licence) — the algorithm
// is his; the code here is written for one GPU thread per block (packed tables in LDS, 64-bit bit buffer, 8-byte match copies).
//
// One THREAD per BGZF block: blocks are independent, a 1 GB BAM has ~17 000 of them, and a bit-serial decoder has no parallelism
// inside a block worth the bookkeeping.  Every thread keeps its two code tables (counts per length + symbols in canonical order)
// in LDS (432 B per thread, one wave per workgroup, five workgroups per CU); the compressed bytes are read through a 64-bit
// bit buffer, the output is written byte by byte into the block's own slice of the output (LZ77 copies read it back — a thread
// sees its own stores).  Every loop is bounded by the block's input and output sizes, so corrupt data ends in an error code, never
// in a hang.  Host side: mkp_bgzf_inflate (mkp_api.cpp) builds the block table, checks every block's CRC32 of what came back.
#ifndef MKP_INFLATE_HOST_SHIM   // tests/test_host_inflate.py compiles this file for the host with one-thread shims of the HIP built-ins
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

struct MkpBgzfBlock { unsigned long long in_off; unsigned long long out_off; uint32_t in_len; uint32_t out_len; };

namespace {
struct Bits {
  const uint8_t* p; uint32_t n, at; unsigned long long buf; uint32_t cnt; bool over;
  // refill with one unaligned 8-byte load while at least 8 input bytes remain (a byte-by-byte refill costs a memory round trip per byte)
  __device__ __forceinline__ void fill() {
    if (at + 8u <= n) {
      unsigned long long v; __builtin_memcpy(&v, p + at, 8);
      buf |= v << cnt;
      const uint32_t take = (63u - cnt) >> 3;   // whole bytes that fit above the `cnt` bits held
      at += take; cnt += take * 8u;
      if (cnt < 64u) buf &= (1ull << cnt) - 1ull;   // drop the partial byte that was shifted in with them
    } else while (cnt <= 56u && at < n) { buf |= (unsigned long long)p[at++] << cnt; cnt += 8u; }
  }
  __device__ __forceinline__ uint32_t get(uint32_t k) {   // k <= 16
    if (cnt < k) { fill(); if (cnt < k) { over = true; return 0; } }
    const uint32_t v = (uint32_t)(buf & ((1ull << k) - 1ull)); buf >>= k; cnt -= k; return v;
  }
};
// One Huffman code: count[len] = codes of that length, symbols in canonical order (9-bit symbols as a byte + a bit array)
struct Code {
  uint16_t* count; uint8_t* lo; uint32_t* hi;
  __device__ __forceinline__ int sym(int i) const { return (int)lo[i] | (int)(((hi[i >> 5] >> (i & 31)) & 1u) << 8); }
  __device__ __forceinline__ void put(int i, int s) const { lo[i] = (uint8_t)s; const uint32_t m = 1u << (i & 31); if (s & 256) hi[i >> 5] |= m; else hi[i >> 5] &= ~m; }
};

// canonical Huffman decode (RFC 1951 3.2.2) over up to 15 peeked bits; returns -1 on an invalid code or input exhaustion
__device__ __forceinline__ int decode_sym(Bits& b, const Code& h) {
  if (b.cnt < 15u) b.fill();
  uint32_t bits = (uint32_t)b.buf; const uint32_t avail = b.cnt;   // bits above `avail` are zero
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= 15; len++) {
    code |= (int)(bits & 1u); bits >>= 1;
    const int count = h.count[len];
    if (code - count < first) {
      if ((uint32_t)len > avail) { b.over = true; return -1; }
      b.buf >>= len; b.cnt -= (uint32_t)len;
      return h.sym(index + (code - first));
    }
    index += count; first += count; first <<= 1; code <<= 1;
  }
  return -1;
}
// build count[] / symbols from code lengths; returns 0 for a complete code, > 0 incomplete, < 0 over-subscribed
__device__ __forceinline__ int construct(const Code& h, const uint8_t* length, int n) {
  uint16_t offs[16];   // (private memory: indexed by code length; built once per DEFLATE block)
  for (int len = 0; len <= 15; len++) h.count[len] = 0;
  for (int s = 0; s < n; s++) h.count[length[s]]++;
  if (h.count[0] == n) return 0;
  int left = 1;
  for (int len = 1; len <= 15; len++) { left <<= 1; left -= h.count[len]; if (left < 0) return left; }
  offs[1] = 0;
  for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + h.count[len]);
  for (int s = 0; s < n; s++) if (length[s] != 0) h.put(offs[length[s]]++, s);
  return left;
}
// RFC 1951 3.2.5 tables as arithmetic (a table in global memory costs a memory round trip per symbol):
//   length symbol 257+ls: ls < 8 -> 3+ls; ls = 28 -> 258; else extra = ls/4 - 1, base = 3 + ((4 + ls%4) << extra)
//   distance symbol ds:   ds < 4 -> 1+ds;                  else extra = ds/2 - 1, base = 1 + ((2 + ds%2) << extra)
__device__ __forceinline__ uint32_t len_extra(int ls) { return (ls < 8 || ls == 28) ? 0u : (uint32_t)(ls >> 2) - 1u; }
__device__ __forceinline__ uint32_t len_base(int ls) { return ls < 8 ? 3u + (uint32_t)ls : ls == 28 ? 258u : 3u + ((4u + ((uint32_t)ls & 3u)) << ((uint32_t)(ls >> 2) - 1u)); }
__device__ __forceinline__ uint32_t dist_extra(int ds) { return ds < 4 ? 0u : (uint32_t)(ds >> 1) - 1u; }
__device__ __forceinline__ uint32_t dist_base(int ds) { return ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1u)) << ((uint32_t)(ds >> 1) - 1u)); }
// order of the code-length code's lengths (3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15, five bits each
__device__ __forceinline__ uint32_t cl_order(int i) {
  const unsigned long long lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
  const unsigned long long hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
  return (uint32_t)((i < 12 ? lo >> (5 * i) : hi >> (5 * (i - 12))) & 31ull);
}
}  // namespace

#define MKP_INFLATE_THREADS 64
#define MKP_INFLATE_TAB_BYTES 432   // per thread: two count tables (2 x 32 B), symbol high bits (40 + 8 B), 288 + 32 symbol low bytes: 27 KiB per workgroup

// status[i]: 0 ok, 1 input exhausted, 2 bad block type / stored length, 3 bad code lengths, 4 bad symbol, 5 distance too far, 6 output size mismatch
extern "C" __global__ void __launch_bounds__(MKP_INFLATE_THREADS)
mkp_inflate_blocks(const uint8_t* __restrict__ in, const MkpBgzfBlock* __restrict__ blocks, uint32_t n_blocks, uint8_t* __restrict__ out, uint32_t* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) uint8_t tab[MKP_INFLATE_THREADS][MKP_INFLATE_TAB_BYTES];
  const uint32_t bi = blockIdx.x * MKP_INFLATE_THREADS + threadIdx.x;
  if (bi >= n_blocks) return;
  const MkpBgzfBlock bk = blocks[bi];
  uint8_t* t = tab[threadIdx.x];
  const Code lencode{reinterpret_cast<uint16_t*>(t), t + 112, reinterpret_cast<uint32_t*>(t + 64)};
  const Code distcode{reinterpret_cast<uint16_t*>(t + 32), t + 400, reinterpret_cast<uint32_t*>(t + 104)};
  uint8_t lengths[320];   // code lengths of the block being set up (private memory)
  Bits b; b.p = in + bk.in_off; b.n = bk.in_len; b.at = 0; b.buf = 0; b.cnt = 0; b.over = false;
  uint8_t* __restrict__ o = out + bk.out_off;
  const uint32_t cap = bk.out_len;
  uint32_t w = 0, err = 0;
  for (uint32_t guard = 0; guard <= bk.in_len && !err; guard++) {   // a DEFLATE block takes at least 3 bits: at most 8 * in_len / 3 of them
    const uint32_t last = b.get(1), type = b.get(2);
    if (b.over) { err = 1; break; }
    if (type == 0) {   // stored
      const uint32_t drop = b.cnt & 7u; b.buf >>= drop; b.cnt -= drop;
      const uint32_t len = b.get(16), nlen = b.get(16);
      if (b.over) { err = 1; break; }
      if ((len ^ 0xffffu) != nlen || w + len > cap) { err = 2; break; }
      for (uint32_t k = 0; k < len; k++) { const uint32_t v = b.get(8); if (b.over) { err = 1; break; } o[w++] = (uint8_t)v; }
    } else if (type == 1 || type == 2) {
      if (type == 1) {   // fixed codes (3.2.6)
        int s = 0;
        for (; s < 144; s++) lengths[s] = 8;
        for (; s < 256; s++) lengths[s] = 9;
        for (; s < 280; s++) lengths[s] = 7;
        for (; s < 288; s++) lengths[s] = 8;
        construct(lencode, lengths, 288);
        for (s = 0; s < 30; s++) lengths[s] = 5;
        construct(distcode, lengths, 30);
      } else {           // dynamic codes (3.2.7)
        const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
        if (b.over) { err = 1; break; }
        if (nlen > 286 || ndist > 30) { err = 3; break; }
        int idx = 0;
        for (; idx < ncode; idx++) lengths[cl_order(idx)] = (uint8_t)b.get(3);
        for (; idx < 19; idx++) lengths[cl_order(idx)] = 0;
        if (b.over) { err = 1; break; }
        if (construct(lencode, lengths, 19) != 0) { err = 3; break; }   // the code-length code must be complete
        idx = 0;
        while (idx < nlen + ndist) {
          int sym = decode_sym(b, lencode);
          if (sym < 0) { err = b.over ? 1 : 4; break; }
          if (sym < 16) lengths[idx++] = (uint8_t)sym;
          else {
            int len = 0, rep;
            if (sym == 16) { if (idx == 0) { err = 3; break; } len = lengths[idx - 1]; rep = 3 + (int)b.get(2); }
            else if (sym == 17) rep = 3 + (int)b.get(3);
            else rep = 11 + (int)b.get(7);
            if (b.over) { err = 1; break; }
            if (idx + rep > nlen + ndist) { err = 3; break; }
            while (rep--) lengths[idx++] = (uint8_t)len;
          }
        }
        if (err) break;
        if (lengths[256] == 0) { err = 3; break; }   // no end-of-block code
        // the distance lengths follow the literal/length ones in the same array: build the length code first, then move on
        int e1 = construct(lencode, lengths, nlen);
        if (e1 != 0) { err = 3; break; }   // an incomplete literal/length code is never valid (the host decoder's and zlib's rule)
        // (lengths of the distance code start at lengths[nlen]; construct reads them in place)
        int e2 = construct(distcode, lengths + nlen, ndist);
        if (e2 < 0 || (e2 > 0 && !(ndist - distcode.count[0] == 1 && distcode.count[1] == 1))) { err = 3; break; }   // incomplete distance code: only a single one-bit code
      }
      // literal / length + distance symbols until end of block (3.2.5)
      for (uint32_t g2 = 0; g2 <= cap + 1u; g2++) {   // every symbol but the last emits at least one byte
        const int sym = decode_sym(b, lencode);
        if (sym < 0) { err = b.over ? 1 : 4; break; }
        if (sym < 256) { if (w >= cap) { err = 6; break; } o[w++] = (uint8_t)sym; }
        else if (sym == 256) break;
        else {
          const int ls = sym - 257;
          if (ls >= 29) { err = 4; break; }
          const uint32_t len = len_base(ls) + b.get(len_extra(ls));
          const int ds = decode_sym(b, distcode);
          if (ds < 0 || ds >= 30) { err = b.over ? 1 : 4; break; }
          const uint32_t dist = dist_base(ds) + b.get(dist_extra(ds));
          if (b.over) { err = 1; break; }
          if (dist > w) { err = 5; break; }
          if (w + len > cap) { err = 6; break; }
          // LZ77 copy.  Far enough back: 8 bytes per step, the loads of a step independent of the stores before it (dist >= 8), so the
          // round trips of one match overlap instead of adding up byte by byte; closer than that the pattern repeats: byte by byte.
          if (dist >= 8u) {
            uint32_t k = 0;
            for (; k + 8u <= len; k += 8u) { unsigned long long v; __builtin_memcpy(&v, o + w - dist + k, 8); __builtin_memcpy(o + w + k, &v, 8); }
            for (; k < len; k++) o[w + k] = o[w - dist + k];
            w += len;
          } else for (uint32_t k = 0; k < len; k++) { o[w] = o[w - dist]; w++; }
        }
      }
    } else { err = 2; break; }
    if (last) break;
  }
  if (!err && w != cap) err = 6;
  status[bi] = err;
}

// ---- second edition of the one-thread-per-block decoder (round 4).  Same tables and symbol decode; what changed is the memory traffic of
// a lane, which is what a block's 100 ms were made of (every global access of the first edition is a round trip of one lane with nothing
// else to run meanwhile):
//   * input: the next aligned qword is always in flight (two-qword window), a refill is register work;
//   * literals gather in a register and leave as one store per eight (loads and stores share vmcnt on gfx9: a wait for a load also drains
//     every store behind it);
//   * LZ77 copies never read what the same copy wrote: dist >= len reads its source four qwords at a time ahead of the stores;
//     dist < 8 builds the repeating pattern in a register once and only stores; 8 <= dist < len reads the `dist` source bytes cyclically.
//     Tails are byte stores out of a register, not byte round trips.
//   * the codes-per-length counts of both tables in registers: the canonical walk touches LDS only for the symbol itself.
// Reads may run up to 7 bytes past a block's output slice and 8 past its input (never written there): both buffers carry that slack.
// Measured on the C3 BAM (54 450 blocks; tools/dbg/inflate_variants.hip): 117 ms first edition, 89 ms this one — 55 ms of it decode, 34 ms
// copies (literal stores: nothing); 6 000 blocks take 78 ms: a block is a serial chain and the kernel's time is one lane's time.  A
// three-state machine (one decode site per pass for literal/length and distance codes, copies in 32-byte steps) was 146 ms: the extra
// passes and table selects cost more than the second decode site.
#ifdef MKP_INFLATE_DBG   // ablation builds only (tools/dbg/inflate_variants.hip): 1 = no LZ77 copies, 2 = no literal stores
__device__ uint32_t mkp_inflate_dbg;
#define MKP_DBG(bit) (mkp_inflate_dbg & (bit))
#else
#define MKP_DBG(bit) 0
#endif
namespace {
struct Bits2 {
  const uint8_t* p; uint32_t n; uint32_t at;        // block input; next unread byte
  unsigned long long cur, nxt; uint32_t q;           // qwords q and q + 1 of the input (zero past the end)
  unsigned long long buf; uint32_t cnt; bool over;
  __device__ __forceinline__ unsigned long long load_q(uint32_t qi) const {
    const uint32_t off = qi * 8u;
    if (off >= n) return 0ull;
    unsigned long long v; __builtin_memcpy(&v, p + off, 8);   // (may run into the block's trailer / the buffer's slack: masked below)
    if (off + 8u > n) v &= (1ull << (8u * (n - off))) - 1ull;
    return v;
  }
  __device__ __forceinline__ void init() { at = 0; q = 0; cur = load_q(0); nxt = load_q(1); buf = 0; cnt = 0; over = false; }
  __device__ __forceinline__ void fill() {            // whole bytes that fit above the `cnt` bits held
    const uint32_t avail = at < n ? n - at : 0u;
    uint32_t take = (63u - cnt) >> 3; if (take > avail) take = avail;
    if (!take) return;
    const uint32_t sh = (at & 7u) * 8u;
    unsigned long long v = cur >> sh; if (sh) v |= nxt << (64u - sh);
    if (take < 8u) v &= (1ull << (8u * take)) - 1ull;
    buf |= v << cnt; cnt += take * 8u; at += take;
    if ((at >> 3) != q) { q++; cur = nxt; nxt = load_q(q + 1u); }   // (take <= 7: at most one qword boundary is crossed)
  }
  __device__ __forceinline__ uint32_t get(uint32_t k) {   // k <= 16
    if (cnt < k) { fill(); if (cnt < k) { over = true; return 0; } }
    const uint32_t v = (uint32_t)(buf & ((1ull << k) - 1ull)); buf >>= k; cnt -= k; return v;
  }
};
// the codes-per-length counts of one table in registers (two per dword): the length-by-length walk is then register work, only the
// final symbol fetch touches LDS (the first edition read count[len] from LDS at every step: ten dependent round trips per symbol)
struct Counts { uint32_t w[8]; __device__ __forceinline__ void load(const uint16_t* c) { for (int i = 0; i < 8; i++) w[i] = (uint32_t)c[2 * i] | ((uint32_t)c[2 * i + 1] << 16); } };
__device__ __forceinline__ int decode_sym2(Bits2& b, const Code& h, const Counts& cn) {
  if (b.cnt < 15u) b.fill();
  uint32_t bits = (uint32_t)b.buf; const uint32_t avail = b.cnt;
  int code = 0, first = 0, index = 0;
#pragma unroll
  for (int len = 1; len <= 15; len++) {
    code |= (int)(bits & 1u); bits >>= 1;
    const int count = (int)((cn.w[len >> 1] >> (16 * (len & 1))) & 0xffffu);
    if (code - count < first) {
      if ((uint32_t)len > avail) { b.over = true; return -1; }
      b.buf >>= len; b.cnt -= (uint32_t)len;
      return h.sym(index + (code - first));
    }
    index += count; first += count; first <<= 1; code <<= 1;
  }
  return -1;
}
__device__ __forceinline__ unsigned long long ld8(const uint8_t* p) { unsigned long long v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void st8(uint8_t* p, unsigned long long v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ void st_tail(uint8_t* p, unsigned long long v, uint32_t n) { for (uint32_t j = 0; j < n; j++) p[j] = (uint8_t)(v >> (8u * j)); }   // n < 8
}  // namespace

extern "C" __global__ void __launch_bounds__(MKP_INFLATE_THREADS)
mkp_inflate_blocks2(const uint8_t* __restrict__ in, const MkpBgzfBlock* __restrict__ blocks, uint32_t n_blocks, uint8_t* __restrict__ out, uint32_t* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) uint8_t tab[MKP_INFLATE_THREADS][MKP_INFLATE_TAB_BYTES];
  const uint32_t bi = blockIdx.x * MKP_INFLATE_THREADS + threadIdx.x;
  if (bi >= n_blocks) return;
  const MkpBgzfBlock bk = blocks[bi];
  uint8_t* t = tab[threadIdx.x];
  const Code lencode{reinterpret_cast<uint16_t*>(t), t + 112, reinterpret_cast<uint32_t*>(t + 64)};
  const Code distcode{reinterpret_cast<uint16_t*>(t + 32), t + 400, reinterpret_cast<uint32_t*>(t + 104)};
  uint8_t lengths[320];
  Counts lc, dc;
  Bits2 b; b.p = in + bk.in_off; b.n = bk.in_len; b.init();
  uint8_t* o = out + bk.out_off;
  const uint32_t cap = bk.out_len;
  uint32_t w = 0, err = 0;
  unsigned long long lit = 0; uint32_t nlit = 0;   // literals not stored yet: output bytes [w - nlit, w)
  for (uint32_t guard = 0; guard <= bk.in_len && !err; guard++) {
    const uint32_t last = b.get(1), type = b.get(2);
    if (b.over) { err = 1; break; }
    if (type == 0) {   // stored
      const uint32_t drop = b.cnt & 7u; b.buf >>= drop; b.cnt -= drop;
      const uint32_t len = b.get(16), nlen = b.get(16);
      if (b.over) { err = 1; break; }
      if ((len ^ 0xffffu) != nlen || w + len > cap) { err = 2; break; }
      for (uint32_t k = 0; k < len; k++) { const uint32_t v = b.get(8); if (b.over) { err = 1; break; } o[w++] = (uint8_t)v; }
    } else if (type == 1 || type == 2) {
      if (type == 1) {
        int s = 0;
        for (; s < 144; s++) lengths[s] = 8;
        for (; s < 256; s++) lengths[s] = 9;
        for (; s < 280; s++) lengths[s] = 7;
        for (; s < 288; s++) lengths[s] = 8;
        construct(lencode, lengths, 288);
        for (s = 0; s < 30; s++) lengths[s] = 5;
        construct(distcode, lengths, 30);
      } else {
        const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
        if (b.over) { err = 1; break; }
        if (nlen > 286 || ndist > 30) { err = 3; break; }
        int idx = 0;
        for (; idx < ncode; idx++) lengths[cl_order(idx)] = (uint8_t)b.get(3);
        for (; idx < 19; idx++) lengths[cl_order(idx)] = 0;
        if (b.over) { err = 1; break; }
        if (construct(lencode, lengths, 19) != 0) { err = 3; break; }
        lc.load(lencode.count);
        idx = 0;
        while (idx < nlen + ndist) {
          int sym = decode_sym2(b, lencode, lc);
          if (sym < 0) { err = b.over ? 1 : 4; break; }
          if (sym < 16) lengths[idx++] = (uint8_t)sym;
          else {
            int len = 0, rep;
            if (sym == 16) { if (idx == 0) { err = 3; break; } len = lengths[idx - 1]; rep = 3 + (int)b.get(2); }
            else if (sym == 17) rep = 3 + (int)b.get(3);
            else rep = 11 + (int)b.get(7);
            if (b.over) { err = 1; break; }
            if (idx + rep > nlen + ndist) { err = 3; break; }
            while (rep--) lengths[idx++] = (uint8_t)len;
          }
        }
        if (err) break;
        if (lengths[256] == 0) { err = 3; break; }
        int e1 = construct(lencode, lengths, nlen);
        if (e1 != 0) { err = 3; break; }   // an incomplete literal/length code is never valid (the host decoder's and zlib's rule)
        int e2 = construct(distcode, lengths + nlen, ndist);
        if (e2 < 0 || (e2 > 0 && !(ndist - distcode.count[0] == 1 && distcode.count[1] == 1))) { err = 3; break; }   // incomplete distance code: only a single one-bit code
      }
      lc.load(lencode.count); dc.load(distcode.count);
      for (uint32_t g2 = 0; g2 <= cap + 1u; g2++) {
        const int sym = decode_sym2(b, lencode, lc);
        if (sym < 0) { err = b.over ? 1 : 4; break; }
        if (sym < 256) {
          if (w >= cap) { err = 6; break; }
          lit |= (unsigned long long)(uint32_t)sym << (8u * nlit); nlit++; w++;
          if (nlit == 8u) { if (!MKP_DBG(2)) st8(o + w - 8u, lit); lit = 0; nlit = 0; }
        } else {
          if (nlit) { st_tail(o + w - nlit, lit, nlit); lit = 0; nlit = 0; }
          if (sym == 256) break;
          const int ls = sym - 257;
          if (ls >= 29) { err = 4; break; }
          const uint32_t len = len_base(ls) + b.get(len_extra(ls));
          const int ds = decode_sym2(b, distcode, dc);
          if (ds < 0 || ds >= 30) { err = b.over ? 1 : 4; break; }
          const uint32_t dist = dist_base(ds) + b.get(dist_extra(ds));
          if (b.over) { err = 1; break; }
          if (dist > w) { err = 5; break; }
          if (w + len > cap) { err = 6; break; }
          const uint8_t* src = o + w - dist; uint8_t* dst = o + w;
          if (MKP_DBG(1)) { /* ablation: the copy is skipped, the decode goes on */ }
          else if (dist >= len) {
            uint32_t k = 0;
            for (; k + 32u <= len; k += 32u) { const unsigned long long a0 = ld8(src + k), a1 = ld8(src + k + 8u), a2 = ld8(src + k + 16u), a3 = ld8(src + k + 24u);
                                                st8(dst + k, a0); st8(dst + k + 8u, a1); st8(dst + k + 16u, a2); st8(dst + k + 24u, a3); }
            if (k < len) {   // up to 31 bytes: all loads first (the last one may run past the source's end, into bytes that are not used)
              const uint32_t r = len - k; unsigned long long a0 = ld8(src + k), a1 = 0, a2 = 0, a3 = 0;
              if (r > 8u) a1 = ld8(src + k + 8u);
              if (r > 16u) a2 = ld8(src + k + 16u);
              if (r > 24u) a3 = ld8(src + k + 24u);
              if (r >= 8u) st8(dst + k, a0); else st_tail(dst + k, a0, r);
              if (r >= 16u) st8(dst + k + 8u, a1); else if (r > 8u) st_tail(dst + k + 8u, a1, r - 8u);
              if (r >= 24u) st8(dst + k + 16u, a2); else if (r > 16u) st_tail(dst + k + 16u, a2, r - 16u);
              if (r > 24u) st_tail(dst + k + 24u, a3, r - 24u);
            }
          } else if (dist < 8u) {
            // the `dist` bytes before w repeat: eight bytes of that pattern in a register, stored at every multiple of the period
            unsigned long long pat = ld8(src) & ((1ull << (8u * dist)) - 1ull);
            for (uint32_t sft = dist; sft < 8u; sft <<= 1) pat |= pat << (8u * sft);
            const uint32_t step = dist * (8u / dist);   // the largest multiple of the period within eight bytes
            uint32_t k = 0;
            for (; k + 8u <= len; k += step) st8(dst + k, pat);
            if (k < len) st_tail(dst + k, pat, len - k);
          } else {
            // 8 <= dist < len: output byte k is source byte k mod dist; the source bytes are final, so no load waits for a store of this copy
            uint32_t off = 0, k = 0;
            for (; k < len; k += 8u) {
              unsigned long long v;
              if (off + 8u <= dist) v = ld8(src + off);
              else { const uint32_t head = dist - off; v = (ld8(src + off) & ((1ull << (8u * head)) - 1ull)) | (ld8(src) << (8u * head)); }
              if (k + 8u <= len) st8(dst + k, v); else st_tail(dst + k, v, len - k);
              off += 8u; if (off >= dist) off -= dist;
            }
          }
          w += len;
        }
      }
      if (nlit) { st_tail(o + w - nlit, lit, nlit); lit = 0; nlit = 0; }
    } else { err = 2; break; }
    if (last) break;
  }
  if (!err && w != cap) err = 6;
  status[bi] = err;
}

#ifndef MKP_INFLATE_HOST_SHIM
extern "C" hipError_t mkp_launch_inflate2(hipStream_t st, const uint8_t* in, const MkpBgzfBlock* blocks, uint32_t n_blocks, uint8_t* out, uint32_t* status) {
  if (!n_blocks) return hipSuccess;
  hipLaunchKernelGGL(mkp_inflate_blocks2, dim3((n_blocks + MKP_INFLATE_THREADS - 1) / MKP_INFLATE_THREADS), dim3(MKP_INFLATE_THREADS), 0, st, in, blocks, n_blocks, out, status);
  return hipGetLastError();
}
#endif

#ifndef MKP_INFLATE_HOST_SHIM
extern "C" hipError_t mkp_launch_inflate(hipStream_t st, const uint8_t* in, const MkpBgzfBlock* blocks, uint32_t n_blocks, uint8_t* out, uint32_t* status) {
  if (!n_blocks) return hipSuccess;
  hipLaunchKernelGGL(mkp_inflate_blocks, dim3((n_blocks + MKP_INFLATE_THREADS - 1) / MKP_INFLATE_THREADS), dim3(MKP_INFLATE_THREADS), 0, st, in, blocks, n_blocks, out, status);
  return hipGetLastError();
}
#endif
