// BGZF block inflate, one wave per block, speculative token decode with a PARALLEL output step (RFC 1951, SAM spec §4.1).
//
// Lane k decodes the token (literal | length + distance) that would start at bit `pos + k` — two LDS table probes per lane plus one probe
// each of the length / distance alphabet tables (mkp_inflate_tok.hpp) — and a scalar walk follows the real chain 0 -> n(0) -> n(0) + n(n(0))
// -> ... through the 64 answers.  Round 4's kernels (mkp_inflate_wave2 / _wave3) did all of a token's work inside that walk, ~48 scalar and
// vector instructions per token on a wave that issues one instruction every ~10 cycles.  Here the walk only MARKS the chain (one
// v_readlane, one s_bitset, one add per token) and the rest is done for all tokens of the pass at once:
//   * output offsets: a DPP prefix sum of the output lengths over the marked lanes; the pass takes the longest prefix of the chain whose
//     bytes fit 64 output lanes and the block's size (the rest is decoded again by the next pass);
//   * byte j of the pass finds its token: the accepted tokens drop their descriptor at LDS slot [offset]; a ballot of the non-empty slots and
//     a find-first-set over the lanes at or below j give the covering token, one ds_bpermute fetches its descriptor;
//   * every lane then knows where its byte comes from — a literal, a ring position, a byte of the flushed output (8 KiB ring: matches further
//     back than the ring), or a byte produced earlier IN THIS PASS (a match that reaches into the pass's own output, self-overlapping
//     matches included).  In-pass references are resolved by pointer jumping over the lanes (ds_bpermute until no lane refers to a lane;
//     0 rounds for 4 passes out of 5), so short-distance matches no longer end a pass or flush anything;
//   * one LDS load + one LDS store per lane move the pass's bytes into the ring.
// End of block, codes longer than the 11- / 9-bit direct tables, matches longer than 64 bytes and anything invalid end the chain and are
// decoded by the wave as one token (one_token).  The compressed bytes reach the lanes through a 1 KiB circular LDS window (512 bytes ahead
// in registers); the ring goes out a quarter at a time.  LDS 15.6 KiB per wave with the 8 KiB ring: ten waves per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mkp_inflate_tok.hpp"

struct MkpBgzfBlock { unsigned long long in_off; unsigned long long out_off; uint32_t in_len; uint32_t out_len; };

namespace {
constexpr uint32_t LIT_BITS = MKP_W4_LIT_BITS, DIST_BITS = MKP_W4_DIST_BITS, INW = MKP_W4_INW;

// order this wave's LDS traffic only: a fence without the address space also drains the global stores of the output bytes (vmcnt(0)),
// ~1 us at every use
#define LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); } while (0)

__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long sgpr64(unsigned long long v) {
  return (unsigned long long)sgpr((uint32_t)v) | ((unsigned long long)sgpr((uint32_t)(v >> 32)) << 32);
}

// inclusive prefix sum over the 64 lanes: row-wise shifts by 1, 2, 4, 8, then the row totals carried across (six v_add_u32_dpp)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /*row_shr:2*/, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /*row_shr:4*/, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /*row_shr:8*/, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /*row_bcast:31*/, 0xc, 0xf, false);
  return v;
}

template <uint32_t RINGSZ>
struct Spec4Lds {   // 6 144 bytes + the ring: 10 240 with a 4 KiB ring = sixteen waves per CU
  uint8_t ring[RINGSZ];
  uint16_t lit[1u << LIT_BITS];
  uint16_t dist[1u << DIST_BITS];
  uint16_t lcount[16], dcount[16];   // canonical fallback: codes per length ...
  uint16_t lsym[288], dsym[32];      // ... and symbols in canonical order
  union { uint8_t lens[320]; uint32_t hd[64]; };   // code lengths while a block's tables are built; the pass's head slots while it is decoded
  uint32_t inw[INW];   // input bytes [lo, lo + 512), byte x at inw-byte x mod 512
};

// direct table + canonical lists from lens[0, n): all lanes.  ENC(code length, symbol) = the table entry.  Returns 0 complete, > 0
// incomplete, < 0 over-subscribed.
template <class ENC>
__device__ __forceinline__ int build(const uint8_t* lens, int n, uint16_t* tab, uint32_t tab_bits, uint16_t* count, uint16_t* syms, int lane,
    ENC enc) {
  if (lane < 16) count[lane] = 0;
  for (uint32_t i = (uint32_t)lane; i < (1u << tab_bits); i += 64u) tab[i] = 0;
  LDS_SYNC();
  // two u16 counters per dword
  for (int s = lane; s < n; s += 64) { const uint32_t l = lens[s];
    if (l) atomicAdd(reinterpret_cast<uint32_t*>(count) + (l >> 1), (l & 1u) ? 0x10000u : 1u);
    }
  LDS_SYNC();
  uint32_t next_code[16], offs[16]; int left = 1; uint32_t code = 0, off = 0; uint32_t used = 0;
  next_code[0] = 0; offs[0] = 0;
  for (int l = 1; l <= 15; l++) {
    const uint32_t c = sgpr(count[l]);
    left = (left << 1) - (int)c; code = (code + (l > 1 ? sgpr(count[l - 1]) : 0u)) << 1; next_code[l] = code; offs[l] = off; off += c; used += c;
  }
  if (left < 0) return left;
  if (used == 0) return 0;
  // symbols in order: the code of a symbol is next_code[len]++ (uniform walk); its table slots are filled by all lanes
  for (int s = 0; s < n; s++) {
    const uint32_t l = sgpr(lens[s]); if (!l) continue;
    uint32_t c = 0, o = 0;
#pragma unroll
    for (int k = 1; k <= 15; k++) if ((uint32_t)k == l) { c = next_code[k]++; o = offs[k]++; }   // (register arrays: constant indices only)
    if (lane == 0) syms[o] = (uint16_t)s;
    if (l <= tab_bits) {
      const uint32_t rev = __builtin_bitreverse32(c) >> (32u - l);
      const uint16_t ent = enc(l, (uint32_t)s);
      for (uint32_t k = rev + ((uint32_t)lane << l); k < (1u << tab_bits); k += 64u << l) tab[k] = ent;
    }
  }
  LDS_SYNC();
  return left;
}

__device__ __forceinline__ uint32_t cl_order(int i) {
  const unsigned long long lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
  const unsigned long long hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
  return (uint32_t)((i < 12 ? lo >> (5 * i) : hi >> (5 * (i - 12))) & 31ull);
}

// ring -> global, output bytes [from, to): 16 bytes per lane and step (from is a multiple of 16 unless it follows a stored block)
template <uint32_t RINGSZ>
__device__ __forceinline__ void flush_ring(const uint8_t* ring, uint8_t* __restrict__ o, uint32_t from, uint32_t to, int lane) {
  uint32_t a = from;
  if (a & 15u) { const uint32_t head = min(to, (a + 15u) & ~15u);
    for (uint32_t k = a + (uint32_t)lane; k < head; k += 64u) o[k] = ring[k & (RINGSZ - 1u)];
    a = head; }
  const uint32_t units = (to - a) >> 4;
  for (uint32_t u = (uint32_t)lane; u < units; u += 64u) { const uint32_t at = a + 16u * u;
    uint4 v = *reinterpret_cast<const uint4*>(ring + (at & (RINGSZ - 1u))); __builtin_memcpy(o + at, &v, 16); }
  for (uint32_t k = a + 16u * units + (uint32_t)lane; k < to; k += 64u) o[k] = ring[k & (RINGSZ - 1u)];
}

// an output byte this wave flushed earlier, read back from global memory past the vector L1 (the flush's stores are write-through; an
// agent-scope load does not look at L1 lines that may predate them)
__device__ __forceinline__ uint32_t far_byte(const uint8_t* p) { return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the block's compressed bytes behind the circular LDS window: 512 bytes, the next 256 in registers
struct In2 {
  const uint8_t* p; uint32_t n;
  uint32_t lo;          // window start (multiple of 256; uniform)
  uint32_t pf;          // this lane's dword of [lo + 512, lo + 768)
  uint32_t* w;
  __device__ __forceinline__ uint32_t load_word(uint32_t off) const {   // dword at byte `off`; zero past the end
    if (off + 4u <= n) { uint32_t v; __builtin_memcpy(&v, p + off, 4); return v; }
    uint32_t v = 0; for (uint32_t k = 0; k < 4u; k++) if (off + k < n) v |= (uint32_t)p[off + k] << (8u * k);
    return v;
  }
  __device__ __forceinline__ void seek(uint32_t byte, int lane) {   // window around `byte`, from scratch
    lo = byte & ~255u;
    LDS_SYNC();
#pragma unroll
    for (uint32_t k = 0; k < INW / 64u; k++) w[(((lo >> 2) + 64u * k) + (uint32_t)lane) & (INW - 1u)] = load_word(lo + 256u * k + 4u * (uint32_t)lane);
    pf = load_word(lo + 4u * INW + 4u * (uint32_t)lane);
    LDS_SYNC();
  }
  __device__ __forceinline__ void ensure(uint32_t pos_bits, int lane) {   // the 24 bytes from pos_bits / 8 on are in the window
    const uint32_t byte = pos_bits >> 3;
    if (byte >= lo + 2048u) { seek(byte, lane); return; }
    while (byte >= lo + 256u) {   // the first half is behind the read position: the prefetched 256 bytes take its place
      const uint32_t base = (lo >> 2) & (INW - 1u);   // 0 or 64
      LDS_SYNC();
      w[base + (uint32_t)lane] = pf;
      lo += 256u;
      pf = load_word(lo + 4u * INW + 4u * (uint32_t)lane);
      LDS_SYNC();
    }
  }
  __device__ __forceinline__ unsigned long long peek(uint32_t pos_bits) const {   // uniform: 64 stream bits from pos_bits on
    uint32_t a, b; mkp_tok_window2(w, pos_bits, &a, &b); return (unsigned long long)sgpr(a) | ((unsigned long long)sgpr(b) << 32);
  }
};

// uniform reader of the block headers: 64 bits cached at cpos
struct Hdr {
  unsigned long long cb; uint32_t cpos;
  __device__ __forceinline__ void load(In2& in, uint32_t pos, int lane) { in.ensure(pos, lane); cb = in.peek(pos); cpos = pos; }
  __device__ __forceinline__ uint32_t get(In2& in, uint32_t& pos, uint32_t k, int lane) {   // k <= 16
    if (pos + k > cpos + 64u) load(in, pos, lane);
    const uint32_t v = (uint32_t)(cb >> (pos - cpos)) & ((1u << k) - 1u);
    pos += k; return v;
  }
  __device__ __forceinline__ uint32_t peek16(In2& in, uint32_t pos, int lane) { if (pos + 16u > cpos + 64u) load(in, pos, lane);
    return (uint32_t)(cb >> (pos - cpos)) & 0xffffu; }
};

// canonical decode (RFC 1951 §3.2.2) of the code starting at the low end of `bits`; returns the symbol or -1, its length in *l
__device__ __forceinline__ int canon_sym(uint32_t bits, const uint16_t* count, const uint16_t* syms, uint32_t* l) {
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= 15; len++) {
    code |= (int)(bits & 1u); bits >>= 1;
    const int c = (int)sgpr(count[len]);
    if (code - c < first) { *l = (uint32_t)len; return (int)sgpr(syms[index + (code - first)]); }
    index += c; first += c; first <<= 1; code <<= 1;
  }
  return -1;
}

// the token at bit q, decoded by the wave as one (uniform; any code length): err = 0 or the status to report
struct OneTok { uint32_t err, bits, kind, val, dist; };   // kind MKP_TK_*; val = literal byte | match length
template <class LDS>
__device__ __forceinline__ OneTok one_token(const In2& in, const LDS& L, uint32_t q) {
  OneTok r; r.err = 0; r.bits = 0; r.kind = MKP_TK_EOB; r.val = 0; r.dist = 0;
  const unsigned long long bits = in.peek(q);
  const uint32_t e = sgpr(L.lit[(uint32_t)bits & ((1u << LIT_BITS) - 1u)]); uint32_t l = e & 15u, ex, base;
  if (l && !(e & 16u)) { r.bits = l; r.kind = MKP_TK_LIT; r.val = (e >> 5) & 255u; return r; }
  if (l && (e >> 13) != 7u) { ex = e >> 13; base = ((e >> 5) & 255u) + 3u; }
  else {   // no code this short, or a stop entry: the canonical lists know every symbol
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
  const uint32_t d = sgpr(L.dist[(uint32_t)(bits >> n) & ((1u << DIST_BITS) - 1u)]); uint32_t dl = d & 15u; int ds = (int)((d >> 4) & 31u);
  if (!dl) { ds = canon_sym((uint32_t)(bits >> n), L.dcount, L.dsym, &dl); if (ds < 0) { r.err = 4u; return r; } }
  if (ds >= 30) { r.err = 4u; return r; }
  const uint32_t dx = dist_extra(ds);
  r.dist = dist_base(ds) + ((uint32_t)(bits >> (n + dl)) & ((1u << dx) - 1u));
  r.bits = n + dl + dx;   // <= 15 + 5 + 15 + 13 = 48
  return r;
}
}  // namespace

// status[i]: 0 ok, 1 input exhausted, 2 bad block type / stored length, 3 bad code lengths, 4 bad symbol, 5 distance too far, 6 output size mismatch
//
// RINGSZ = 4096 (mkp_inflate_wave4): sixteen waves per CU; the ring goes out a quarter at a time, a match further back than the ring minus 128
// takes its bytes from the flushed output (always flushed: the unflushed tail is at most a quarter + one pass + one long match), one
// agent-scope load for all such lanes of a pass.
template <uint32_t RINGSZ, bool FARM>
__device__ __forceinline__ void inflate_wave_par(const uint8_t* __restrict__ in_bytes, const MkpBgzfBlock* __restrict__ blocks, uint32_t n_blocks,
    uint8_t* __restrict__ out, uint32_t* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) Spec4Lds<RINGSZ> L;
  constexpr uint32_t RING = RINGSZ, FLQ = RINGSZ >= 32768u ? RINGSZ / 2u : RINGSZ / 4u, NEAR = RINGSZ - 128u;
  const uint32_t bi = blockIdx.x;
  if (bi >= n_blocks) return;
  const int lane = (int)threadIdx.x;
  const MkpBgzfBlock bk = blocks[bi];
  uint8_t* __restrict__ o = out + bk.out_off;
  const uint32_t cap = bk.out_len, in_bits = 8u * bk.in_len;
  In2 in; in.p = in_bytes + bk.in_off; in.n = bk.in_len; in.w = L.inw; in.seek(0, lane);
  Hdr h;
  uint32_t pos = 0;                       // bit position in the block input (uniform)
  uint32_t w = 0, err = 0, flushed = 0;   // uniform; output bytes [flushed, w) are in the ring only
  constexpr uint32_t LITERAL = MKP_SV_LITERAL, M = RING - 1u;
  // lanes at or below this one, as two dwords (the covering-token search)
  const uint32_t le_lo = lane >= 31 ? 0xffffffffu : (2u << lane) - 1u, le_hi = lane < 32 ? 0u : lane == 63 ? 0xffffffffu : (2u << (lane - 32)) - 1u;
  for (uint32_t guard = 0; guard <= bk.in_len && !err; guard++) {
    h.load(in, pos, lane);
    const uint32_t last = h.get(in, pos, 1, lane), type = h.get(in, pos, 2, lane);
    if (type == 0) {   // stored: byte-aligned LEN / NLEN, then raw bytes, copied by all lanes straight from the input
      pos = (pos + 7u) & ~7u;
      const uint32_t len = h.get(in, pos, 16, lane), nlen = h.get(in, pos, 16, lane);
      if ((len ^ 0xffffu) != nlen || w + len > cap) { err = 2; break; }
      const uint32_t at = pos >> 3;
      if ((unsigned long long)at + len > bk.in_len) { err = 1; break; }
      LDS_SYNC(); flush_ring<RINGSZ>(L.ring, o, flushed, w, lane);
      for (uint32_t k = (uint32_t)lane; k < len; k += 64u) { const uint8_t v = in.p[at + k]; o[w + k] = v; L.ring[(w + k) & (RING - 1u)] = v; }
      w += len; flushed = w; pos = 8u * (at + len);
      if (FARM) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // (later far reads may want these bytes)
    } else if (type == 1 || type == 2) {
      int nlen_codes = 288, ndist_codes = 30;
      if (type == 1) {   // fixed codes (§3.2.6)
        LDS_SYNC();
        for (int s = lane; s < 288; s += 64) L.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
        if (lane < 32) L.lens[288 + lane] = 5;
        ndist_codes = 32;
      } else {           // dynamic codes (§3.2.7)
        const int nlen = (int)h.get(in, pos, 5, lane) + 257, ndist = (int)h.get(in, pos, 5, lane) + 1, ncode = (int)h.get(in, pos, 4, lane) + 4;
        if (nlen > 286 || ndist > 30) { err = 3; break; }
        LDS_SYNC();
        if (lane < 19) L.lens[lane] = 0;
        LDS_SYNC();
        for (int idx = 0; idx < ncode; idx++) { const uint32_t v = h.get(in, pos, 3, lane); if (lane == 0) L.lens[cl_order(idx)] = (uint8_t)v; }
        LDS_SYNC();
        // the code-length code, in the distance table's storage; must be complete
        if (build(L.lens, 19, L.dist, DIST_BITS, L.dcount, L.dsym, lane, mkp_w4_plain_entry) != 0) { err = 3; break; }
        __builtin_amdgcn_wave_barrier();
        int idx = 0;
        while (idx < nlen + ndist) {
          // (a code-length code has at most 7 bits: always in the table)
          const uint32_t e = sgpr(L.dist[h.peek16(in, pos, lane) & ((1u << DIST_BITS) - 1u)]);
          if (!(e & 15u)) { err = 4; break; }
          pos += (e & 15u);
          const int sym = (int)(e >> 4);
          if (sym < 16) { if (lane == 0) L.lens[idx] = (uint8_t)sym; idx++; }
          else {
            int len = 0, rep;
            if (sym == 16) { if (idx == 0) { err = 3; break; } LDS_SYNC();
              len = (int)sgpr(L.lens[idx - 1]); rep = 3 + (int)h.get(in, pos, 2, lane); }
            else if (sym == 17) rep = 3 + (int)h.get(in, pos, 3, lane);
            else rep = 11 + (int)h.get(in, pos, 7, lane);
            if (idx + rep > nlen + ndist) { err = 3; break; }
            for (int k = lane; k < rep; k += 64) L.lens[idx + k] = (uint8_t)len;
            idx += rep;
          }
        }
        if (err) break;
        LDS_SYNC();
        const uint8_t mine = lane < ndist ? L.lens[nlen + lane] : 0;   // the distance lengths move to lens[288, 288 + ndist)
        LDS_SYNC();
        if (lane < ndist) L.lens[288 + lane] = mine;
        nlen_codes = nlen; ndist_codes = ndist;
        LDS_SYNC();
        if (sgpr(L.lens[256]) == 0u) { err = 3; break; }   // no end-of-block code
      }
      LDS_SYNC();
      {
        const int e1 = build(L.lens, nlen_codes, L.lit, LIT_BITS, L.lcount, L.lsym, lane, mkp_w4_lit_entry);
        // an incomplete literal/length code is only valid as a single one-bit code (zlib's inflate_table: max == 1 — a block that holds
        // nothing but its end-of-block symbol); the unused code then decodes to nothing and is an error where it is met
        { uint32_t used1 = 0; for (int l = 1; l <= 15; l++) used1 += sgpr(L.lcount[l]);
          if (e1 < 0 || (e1 > 0 && !(used1 == 1u && sgpr(L.lcount[1]) == 1u))) { err = 3; break; } }
        const int e2 = build(L.lens + 288, ndist_codes, L.dist, DIST_BITS, L.dcount, L.dsym, lane, mkp_w4_dist_entry);
        uint32_t used2 = 0; for (int l = 1; l <= 15; l++) used2 += sgpr(L.dcount[l]);
        // incomplete distance code: only a single one-bit code
        if (e2 < 0 || (e2 > 0 && !(used2 == 1u && sgpr(L.dcount[1]) == 1u))) { err = 3; break; }
      }
      LDS_SYNC();   // (lens is dead from here on: its storage is the passes' head slots)
      // tokens until end of block (§3.2.5), 64 bit positions per pass
      bool eob = false;
      while (!eob && !err) {
        if (pos > in_bits + 64u) { err = 1; break; }   // ran off the input (zeros follow it in the window)
        in.ensure(pos, lane);
        uint32_t b_lo, b_hi; mkp_tok_window2(L.inw, pos + (uint32_t)lane, &b_lo, &b_hi);
        const MkpTok4 t = mkp_tok_decode4(b_lo, b_hi, L.lit, L.dist);
        // the chain: which of the 64 answers are tokens.  The walk marks and follows; a position the pass cannot place reports MKP_NX_STOP,
        // which carries i past every in-range value: one compare per token
        uint32_t i = 0; unsigned long long chain = 0;
        do { chain |= 1ull << i; i += (uint32_t)__builtin_amdgcn_readlane((int)t.nx, (int)i); } while (i < 64u);
        // i = bit offset of the first position beyond the chain; stop_nx = what that position reported, if the walk stopped on it
        uint32_t stop_nx = 0;
        if (i >= MKP_NX_STOP) { i = 63u - (uint32_t)__builtin_clzll(chain); chain &= ~(1ull << i);
          stop_nx = (uint32_t)__builtin_amdgcn_readlane((int)t.nx, (int)i); }
        const bool in_chain = (chain >> lane) & 1ull;
        const uint32_t ol = in_chain ? t.ol : 0u;
        const uint32_t incl = wave_incl_scan(ol), excl = incl - ol;
        // the pass takes the longest prefix of the chain that fits 64 output lanes, the block's size, and whose distances reach no further than the
        // output so far
        const bool ok = incl <= 64u && w + incl <= cap && ((t.desc & LITERAL) || t.desc <= w + excl);
        const unsigned long long rej = __builtin_amdgcn_ballot_w64(in_chain && !ok);
        unsigned long long acc = chain; uint32_t adv = i, n_out; bool special;
        if (rej) { const uint32_t first = (uint32_t)__builtin_ctzll(rej); acc = chain & ((1ull << first) - 1ull); adv = first;
          n_out = (uint32_t)__builtin_amdgcn_readlane((int)excl, (int)first); special = first == 0u; }
        else { n_out = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63); special = i < 64u; }
        if (n_out) {
          // byte j -> its token: descriptors dropped at the tokens' first output lanes, then "the nearest non-empty slot at or below j"
          L.hd[lane] = 0;
          if ((acc >> lane) & 1ull) L.hd[excl] = t.desc;
          LDS_SYNC();   // (another lane's store: without the fence the compiler forwards this lane's own zero)
          const uint32_t hv = L.hd[lane];
          const unsigned long long heads = __builtin_amdgcn_ballot_w64(hv != 0u);
          const uint32_t mhi = (uint32_t)(heads >> 32) & le_hi, mlo = (uint32_t)heads & le_lo;
          // (lane 0 is a head whenever n_out > 0)
          const uint32_t hj = mhi ? 63u - (uint32_t)__builtin_clz(mhi) : 31u - (uint32_t)__builtin_clz(mlo | 1u);
          const uint32_t desc = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(hj << 2), (int)hv);
          uint32_t sv = (uint32_t)lane < n_out ? mkp_w4_source(desc, (uint32_t)lane, w, M, NEAR, FARM) : LITERAL;
          // bytes produced earlier in this pass: follow the references until every lane names a literal, the ring or the flushed output
          while (__builtin_amdgcn_ballot_w64((sv & (LITERAL | MKP_SV_INPASS)) == MKP_SV_INPASS)) {
            const uint32_t other = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((sv & 63u) << 2), (int)sv);
            if ((sv & (LITERAL | MKP_SV_INPASS)) == MKP_SV_INPASS) sv = other;
          }
          uint32_t r_ = L.ring[sv & M];
          if (FARM) { const bool far_ = (sv & (LITERAL | MKP_SV_FAR)) == MKP_SV_FAR; if (__builtin_amdgcn_ballot_w64(far_)) {
              if (far_) r_ = far_byte(o + (sv & 0xfffffu));
            } }
          if ((uint32_t)lane < n_out) L.ring[(w + (uint32_t)lane) & M] = (uint8_t)((sv & LITERAL) ? sv : r_);
          w += n_out;
        }
        pos += adv;
        if (special) {   // the token at pos, on its own
          OneTok k;
          if (!rej && (stop_nx & (MKP_NX_STOP - 1u))) {   // a long match the lane at the stop already decoded
            k.err = 0; k.kind = MKP_TK_MATCH; k.bits = stop_nx & (MKP_NX_STOP - 1u);
            k.val = (uint32_t)__builtin_amdgcn_readlane((int)t.ol, (int)adv); k.dist = (uint32_t)__builtin_amdgcn_readlane((int)t.desc, (int)adv);
          } else k = one_token(in, L, pos);
          if (k.err) { err = k.err; break; }
          pos += k.bits;
          if (k.kind == MKP_TK_EOB) eob = true;
          else if (k.kind == MKP_TK_LIT) {
            if (w >= cap) { err = 6; break; }
            if (lane == 0) L.ring[w & M] = (uint8_t)k.val;
            w++;
          } else {
            const uint32_t len = k.val, dist = k.dist;
            if (dist > w) { err = 5; break; }
            if (w + len > cap) { err = 6; break; }
            const uint32_t src0 = w - dist;
            if (FARM && dist > NEAR) {   // (dist >= len here: the flushed output is the source)
              for (uint32_t k2 = (uint32_t)lane; k2 < len; k2 += 64u) L.ring[(w + k2) & M] = (uint8_t)far_byte(o + src0 + k2);
            } else if (dist >= len) {
              for (uint32_t k2 = (uint32_t)lane; k2 < len; k2 += 64u) L.ring[(w + k2) & M] = L.ring[(src0 + k2) & M];
            } else if (dist == 1u) {
              const uint8_t v = L.ring[src0 & M];
              for (uint32_t k2 = (uint32_t)lane; k2 < len; k2 += 64u) L.ring[(w + k2) & M] = v;
            } else {   // the source cycle, lane k at k mod dist: one division, then steps of 64 mod dist
              const uint32_t step = 64u % dist; uint32_t r = (uint32_t)lane % dist;
              for (uint32_t k2 = (uint32_t)lane; k2 < len; k2 += 64u) { L.ring[(w + k2) & M] = L.ring[(src0 + r) & M]; r += step;
                r -= r >= dist ? dist : 0u; }
            }
            w += len;
          }
        }
        // a quarter of the ring is complete: it goes out in one coalesced sweep, long before the write position comes round to it again
        if ((w & ~(FLQ - 1u)) > flushed) { const uint32_t upto = w & ~(FLQ - 1u); LDS_SYNC(); flush_ring<RINGSZ>(L.ring, o, flushed, upto, lane);
          flushed = upto;
          if (FARM) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
      }
    } else { err = 2; break; }
    if (err || last) break;
  }
  LDS_SYNC(); flush_ring<RINGSZ>(L.ring, o, flushed, w, lane);
  if (!err && w != cap) err = 6;
  if (!err && pos > in_bits) err = 1;
  if (lane == 0) status[bi] = err;
}

extern "C" __global__ void __launch_bounds__(64)
mkp_inflate_wave4(const uint8_t* __restrict__ in_bytes, const MkpBgzfBlock* __restrict__ blocks, uint32_t n_blocks, uint8_t* __restrict__ out,
    uint32_t* __restrict__ status) {
  inflate_wave_par<4096u, true>(in_bytes, blocks, n_blocks, out, status);
}
// ring-size variants for A/B runs (MKP_INFLATE_KERNEL=wave4_8k | wave4_2k): 8 KiB = 11 waves per CU and fewer far reads, 2 KiB = more of them (and no
// more waves: 16 per CU is the VGPR limit)
extern "C" __global__ void __launch_bounds__(64)
mkp_inflate_wave4_8k(const uint8_t* __restrict__ in_bytes, const MkpBgzfBlock* __restrict__ blocks, uint32_t n_blocks, uint8_t* __restrict__ out,
    uint32_t* __restrict__ status) {
  inflate_wave_par<8192u, true>(in_bytes, blocks, n_blocks, out, status);
}
extern "C" __global__ void __launch_bounds__(64)
mkp_inflate_wave4_2k(const uint8_t* __restrict__ in_bytes, const MkpBgzfBlock* __restrict__ blocks, uint32_t n_blocks, uint8_t* __restrict__ out,
    uint32_t* __restrict__ status) {
  inflate_wave_par<2048u, true>(in_bytes, blocks, n_blocks, out, status);
}

extern "C" hipError_t mkp_launch_inflate_wave4(hipStream_t st, const uint8_t* in, const void* blocks, uint32_t n_blocks, uint8_t* out,
    uint32_t* status, int variant) {
  if (!n_blocks) return hipSuccess;
  if (variant == 8) hipLaunchKernelGGL(mkp_inflate_wave4_8k, dim3(n_blocks), dim3(64), 0, st, in, (const MkpBgzfBlock*)blocks, n_blocks, out, status);
  else if (variant == 2) hipLaunchKernelGGL(mkp_inflate_wave4_2k, dim3(n_blocks), dim3(64), 0, st, in, (const MkpBgzfBlock*)blocks, n_blocks, out,
      status);
  else hipLaunchKernelGGL(mkp_inflate_wave4, dim3(n_blocks), dim3(64), 0, st, in, (const MkpBgzfBlock*)blocks, n_blocks, out, status);
  return hipGetLastError();
}
