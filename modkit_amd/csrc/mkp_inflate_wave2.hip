// BGZF block inflate, one wave per block, SPECULATIVE symbol decode (RFC 1951, SAM spec §4.1).
//
// mkp_inflate_wave decodes one symbol at a time with a wave-uniform state: bit buffer -> LDS table -> extra bits -> LDS table -> copy,
// ~440 cycles of dependent latency per symbol with 63 lanes watching.  Here the lanes decode AHEAD of the chain: lane k decodes the
// token (literal | length + distance | end of block) that starts at bit `pos + k` as if one started there — two LDS table probes per
// lane, all 64 at once (mkp_inflate_tok.hpp) — and a scalar walk then follows the real chain 0 -> n(0) -> n(0) + n(n(0)) -> ...
// through the 64 answers with v_readlane, ~6 tokens per pass on BAM data (10 bits per token).  A wave alone on its SIMD pays ~10
// cycles per instruction whatever it is (SQ counters: 1.0 M instructions = 10 M cycles per block for the first version of this walk), so
// the walk is kept to one exit and a few dozen instructions per token:
//   * output bytes are not written token by token: lane j of a 64-byte OUTPUT WINDOW notes where byte j comes from — a literal value,
//     or a ring position for a match byte — with a handful of VALU per token, and the window goes out in one LDS load + one LDS store
//     when it is full (or when a match needs bytes that are still in it).  No LDS round trip sits on the per-token path;
//   * the lanes pre-digest their tokens (output length, literal already in window form, "the window can take this" flag); anything
//     else — end of block, codes longer than the direct tables, matches that overlap themselves or are longer than a window, anything
//     invalid or out of bounds — ends the pass, is decoded by the wave as one token (one_token) and handled the plain way.
// The compressed bytes reach the lanes through a 1 KiB circular LDS window (512 bytes ahead in registers, loaded a refill early);
// the output ring, the tables and the table builder are mkp_inflate_wave's.  LDS 39.0 KiB per wave: four waves per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mkp_inflate_wave_common.hpp"

struct MkpBgzfBlock { unsigned long long in_off; unsigned long long out_off; uint32_t in_len; uint32_t out_len; };

namespace {
template <uint32_t RINGSZ>
struct SpecLds {
  uint8_t ring[RINGSZ];
  uint16_t lit[1u << LIT_BITS];
  uint16_t dist[1u << DIST_BITS];
  uint16_t lcount[16], dcount[16];
  uint16_t lsym[288], dsym[32];
  uint8_t lens[320];
  uint32_t inw[256];   // input bytes [lo, lo + 1024), byte x at inw-byte x mod 1024
};

// ring -> global, output bytes [from, to): 16 bytes per lane and step (mkp_inflate_wave_common.hpp's flush for a ring of RINGSZ bytes)
template <uint32_t RINGSZ>
__device__ __forceinline__ void flush_ring(const uint8_t* ring, uint8_t* __restrict__ o, uint32_t from, uint32_t to, int lane) {
  uint32_t a = from;
  if (a & 15u) { const uint32_t head = min(to, (a + 15u) & ~15u); for (uint32_t k = a + (uint32_t)lane; k < head; k += 64u) o[k] = ring[k & (RINGSZ - 1u)]; a = head; }
  const uint32_t units = (to - a) >> 4;
  for (uint32_t u = (uint32_t)lane; u < units; u += 64u) { const uint32_t at = a + 16u * u; uint4 v = *reinterpret_cast<const uint4*>(ring + (at & (RINGSZ - 1u))); __builtin_memcpy(o + at, &v, 16); }
  for (uint32_t k = a + 16u * units + (uint32_t)lane; k < to; k += 64u) o[k] = ring[k & (RINGSZ - 1u)];
}

// an output byte this wave flushed earlier, read back from global memory past the vector L1 (the flush's stores are write-through; an
// agent-scope load does not look at L1 lines that may predate them)
__device__ __forceinline__ uint32_t far_byte(const uint8_t* p) { return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ unsigned long long sgpr64(unsigned long long v) {
  return (unsigned long long)sgpr((uint32_t)v) | ((unsigned long long)sgpr((uint32_t)(v >> 32)) << 32);
}

// the block's compressed bytes behind the LDS window
struct In2 {
  const uint8_t* p; uint32_t n;
  uint32_t lo;          // window start (multiple of 512; uniform)
  uint32_t pf0, pf1;    // this lane's two dwords of [lo + 1024, lo + 1536)
  uint32_t* w;
  __device__ __forceinline__ uint32_t load_word(uint32_t off) const {   // dword at byte `off`; zero past the end
    if (off + 4u <= n) { uint32_t v; __builtin_memcpy(&v, p + off, 4); return v; }
    uint32_t v = 0; for (uint32_t k = 0; k < 4u; k++) if (off + k < n) v |= (uint32_t)p[off + k] << (8u * k);
    return v;
  }
  __device__ __forceinline__ void seek(uint32_t byte, int lane) {   // window around `byte`, from scratch
    lo = byte & ~511u;
    LDS_SYNC();
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) w[(((lo >> 2) + 64u * k) + (uint32_t)lane) & 255u] = load_word(lo + 256u * k + 4u * (uint32_t)lane);
    pf0 = load_word(lo + 1024u + 4u * (uint32_t)lane); pf1 = load_word(lo + 1280u + 4u * (uint32_t)lane);
    LDS_SYNC();
  }
  __device__ __forceinline__ void ensure(uint32_t pos_bits, int lane) {   // the 24 bytes from pos_bits / 8 on are in the window
    const uint32_t byte = pos_bits >> 3;
    if (byte >= lo + 4096u) { seek(byte, lane); return; }
    while (byte >= lo + 512u) {   // the first half is behind the read position: the prefetched 512 bytes take its place
      const uint32_t base = (lo >> 2) & 255u;   // 0 or 128
      LDS_SYNC();
      w[base + (uint32_t)lane] = pf0; w[base + 64u + (uint32_t)lane] = pf1;
      lo += 512u;
      pf0 = load_word(lo + 1024u + 4u * (uint32_t)lane); pf1 = load_word(lo + 1280u + 4u * (uint32_t)lane);
      LDS_SYNC();
    }
  }
  __device__ __forceinline__ unsigned long long peek(uint32_t pos_bits) const { return sgpr64(mkp_tok_window(w, pos_bits)); }   // uniform
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
  __device__ __forceinline__ uint32_t peek16(In2& in, uint32_t pos, int lane) { if (pos + 16u > cpos + 64u) load(in, pos, lane); return (uint32_t)(cb >> (pos - cpos)) & 0xffffu; }
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
  uint32_t e = sgpr(L.lit[(uint32_t)bits & ((1u << LIT_BITS) - 1u)]), l = e & 15u; int sym = (int)(e >> 4);
  if (!l) { sym = canon_sym((uint32_t)bits, L.lcount, L.lsym, &l); if (sym < 0) { r.err = 4u; return r; } }
  r.bits = l;
  if (sym < 256) { r.kind = MKP_TK_LIT; r.val = (uint32_t)sym; return r; }
  if (sym == 256) return r;
  const int ls = sym - 257;
  if (ls >= 29) { r.err = 4u; return r; }
  const uint32_t ex = len_extra(ls);
  r.kind = MKP_TK_MATCH; r.val = len_base(ls) + ((uint32_t)(bits >> l) & ((1u << ex) - 1u));
  uint32_t n = l + ex;
  const uint32_t d = sgpr(L.dist[(uint32_t)(bits >> n) & ((1u << DIST_BITS) - 1u)]); uint32_t dl = d & 15u; int ds = (int)(d >> 4);
  if (!dl) { ds = canon_sym((uint32_t)(bits >> n), L.dcount, L.dsym, &dl); if (ds < 0) { r.err = 4u; return r; } }
  if (ds >= 30) { r.err = 4u; return r; }
  const uint32_t dx = dist_extra(ds);
  r.dist = dist_base(ds) + ((uint32_t)(bits >> (n + dl)) & ((1u << dx) - 1u));
  r.bits = n + dl + dx;   // <= 15 + 5 + 15 + 13 = 48
  return r;
}
}  // namespace

// status[i]: as mkp_inflate_wave — 0 ok, 1 input exhausted, 2 bad block type / stored length, 3 bad code lengths, 4 bad symbol, 5 distance too far, 6 output size mismatch
//
// RINGSZ = 32768, FARM = false: the whole DEFLATE window in LDS, four waves per CU (mkp_inflate_wave2).
// RINGSZ = 8192, FARM = true (mkp_inflate_wave3, MKP_INFLATE_KERNEL=wave3): ten waves per CU; the ring goes out a quarter at a time, a match
// further back than the ring takes its bytes from the flushed output (always flushed: it lies more than RINGSZ - 128 back, the unflushed
// tail is at most a quarter + one pass), one agent-scope load for all such lanes of a window.
template <uint32_t RINGSZ, bool FARM>
__device__ __forceinline__ void inflate_wave_spec(const uint8_t* __restrict__ in_bytes, const MkpBgzfBlock* __restrict__ blocks, uint32_t n_blocks, uint8_t* __restrict__ out, uint32_t* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) SpecLds<RINGSZ> L;
  constexpr uint32_t RING = RINGSZ, FLQ = RINGSZ >= 32768u ? RINGSZ / 2u : RINGSZ / 4u, NEAR = RINGSZ - 128u, FARBIT = 0x40000000u;
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
  // the output window: lane j < fill owes ring[(w - fill + j) & M] its byte — sv = LITERAL | value, or the ring position it is copied from
  constexpr uint32_t LITERAL = MKP_SV_LITERAL, M = RING - 1u;
  uint32_t sv = 0, fill = 0;
#define WINDOW_BYTES(N) do { uint32_t r_ = L.ring[sv & M]; \
    if (FARM) { const bool far_ = (uint32_t)lane < (N) && (sv & (LITERAL | FARBIT)) == FARBIT; if (__builtin_amdgcn_ballot_w64(far_)) { if (far_) r_ = far_byte(o + (sv & 0xfffffu)); } } \
    if ((uint32_t)lane < (N)) L.ring[(w - (N) + (uint32_t)lane) & M] = (uint8_t)((sv & LITERAL) ? sv : r_); } while (0)
#define WINDOW_OUT() do { if (fill) { WINDOW_BYTES(fill); fill = 0; } } while (0)
  for (uint32_t guard = 0; guard <= bk.in_len && !err; guard++) {
    h.load(in, pos, lane);
    const uint32_t last = h.get(in, pos, 1, lane), type = h.get(in, pos, 2, lane);
    if (type == 0) {   // stored: byte-aligned LEN / NLEN, then raw bytes, copied by all lanes straight from the input
      pos = (pos + 7u) & ~7u;
      const uint32_t len = h.get(in, pos, 16, lane), nlen = h.get(in, pos, 16, lane);
      if ((len ^ 0xffffu) != nlen || w + len > cap) { err = 2; break; }
      const uint32_t at = pos >> 3;
      if ((unsigned long long)at + len > bk.in_len) { err = 1; break; }
      WINDOW_OUT(); LDS_SYNC(); flush_ring<RINGSZ>(L.ring, o, flushed, w, lane);
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
        if (build(L.lens, 19, L.dist, DIST_BITS, L.dcount, L.dsym, lane) != 0) { err = 3; break; }   // the code-length code, in the distance table's storage; must be complete
        __builtin_amdgcn_wave_barrier();
        int idx = 0;
        while (idx < nlen + ndist) {
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
        const int e1 = build(L.lens, nlen_codes, L.lit, LIT_BITS, L.lcount, L.lsym, lane);
        if (e1 != 0) { err = 3; break; }   // an incomplete literal/length code is never valid (the host decoder's and zlib's rule)
        const int e2 = build(L.lens + 288, ndist_codes, L.dist, DIST_BITS, L.dcount, L.dsym, lane);
        uint32_t used2 = 0; for (int l = 1; l <= 15; l++) used2 += sgpr(L.dcount[l]);
        if (e2 < 0 || (e2 > 0 && !(used2 == 1u && sgpr(L.dcount[1]) == 1u))) { err = 3; break; }   // incomplete distance code: only a single one-bit code
      }
      // tokens until end of block (§3.2.5), 64 bit positions per pass
      bool eob = false;
      while (!eob && !err) {
        if (pos > in_bits + 64u) { err = 1; break; }   // ran off the input (zeros follow it in the window)
        in.ensure(pos, lane);
        const MkpTok t = mkp_tok_decode(mkp_tok_window(L.inw, pos + (uint32_t)lane), L.lit, L.dist);
        uint32_t i = 0; bool special = false;
        do {
          const uint32_t f0 = sgpr(fill);   // (uniform by construction; said so, the walk's arithmetic stays scalar)
          const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)t.a, (int)i), b = (uint32_t)__builtin_amdgcn_readlane((int)t.b, (int)i);
          const uint32_t ol = a >> 8; const bool lit = (a & MKP_TA_LIT) != 0u;
          if (!((a & MKP_TA_WIN) && w + ol <= cap && (lit || b <= w))) { special = true; break; }
          const bool out = f0 + ol > 64u || (!lit && b < f0 + ol);   // no room — or the match reads bytes that are still in the window (f0 > 0 either way)
          if (out) WINDOW_BYTES(f0);
          const uint32_t f1 = out ? 0u : f0;
          const uint32_t rel = (uint32_t)lane - f1;
          const uint32_t nsv = lit ? b : (FARM && b > NEAR) ? (FARBIT | (w - b + rel)) : ((w - b + rel) & M);
          if (rel < ol) sv = nsv;
          fill = f1 + ol; w += ol; i += a & 63u;
        } while (i < 64u);
        pos += i;
        if (special) {   // the token at pos, on its own
          const OneTok k = one_token(in, L, pos);
          if (k.err) { err = k.err; break; }
          pos += k.bits;
          if (k.kind == MKP_TK_EOB) eob = true;
          else if (k.kind == MKP_TK_LIT) {
            if (w >= cap) { err = 6; break; }
            if (fill == 64u) WINDOW_OUT();
            if ((uint32_t)lane == fill) sv = MKP_SV_LITERAL | k.val;
            fill++; w++;
          } else {
            const uint32_t len = k.val, dist = k.dist;
            if (dist > w) { err = 5; break; }
            if (w + len > cap) { err = 6; break; }
            const uint32_t src0 = w - dist;
            WINDOW_OUT();
            if (FARM && dist > NEAR) {   // (dist >= len here: the flushed output is the source)
              for (uint32_t k2 = (uint32_t)lane; k2 < len; k2 += 64u) L.ring[(w + k2) & M] = (uint8_t)far_byte(o + src0 + k2);
            } else if (dist >= len) {
              for (uint32_t k2 = (uint32_t)lane; k2 < len; k2 += 64u) L.ring[(w + k2) & M] = L.ring[(src0 + k2) & M];
            } else if (dist == 1u) {
              const uint8_t v = L.ring[src0 & M];
              for (uint32_t k2 = (uint32_t)lane; k2 < len; k2 += 64u) L.ring[(w + k2) & M] = v;
            } else {
              for (uint32_t k2 = (uint32_t)lane; k2 < len; k2 += 64u) L.ring[(w + k2) & M] = L.ring[(src0 + k2 % dist) & M];
            }
            w += len;
          }
        }
        // a 16 KiB half of the ring is complete: it goes out in one coalesced sweep, long before the write position comes round to it again
        if ((w & ~(FLQ - 1u)) > flushed) { const uint32_t upto = w & ~(FLQ - 1u); WINDOW_OUT(); LDS_SYNC(); flush_ring<RINGSZ>(L.ring, o, flushed, upto, lane); flushed = upto;
          if (FARM) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
      }
    } else { err = 2; break; }
    if (err || last) break;
  }
  WINDOW_OUT(); LDS_SYNC(); flush_ring<RINGSZ>(L.ring, o, flushed, w, lane);
  if (!err && w != cap) err = 6;
  if (!err && pos > in_bits) err = 1;
  if (lane == 0) status[bi] = err;
#undef WINDOW_OUT
#undef WINDOW_BYTES
}

extern "C" __global__ void __launch_bounds__(64)
mkp_inflate_wave2(const uint8_t* __restrict__ in_bytes, const MkpBgzfBlock* __restrict__ blocks, uint32_t n_blocks, uint8_t* __restrict__ out, uint32_t* __restrict__ status) {
  inflate_wave_spec<32768u, false>(in_bytes, blocks, n_blocks, out, status);
}
extern "C" __global__ void __launch_bounds__(64)
mkp_inflate_wave3(const uint8_t* __restrict__ in_bytes, const MkpBgzfBlock* __restrict__ blocks, uint32_t n_blocks, uint8_t* __restrict__ out, uint32_t* __restrict__ status) {
  inflate_wave_spec<8192u, true>(in_bytes, blocks, n_blocks, out, status);
}

extern "C" hipError_t mkp_launch_inflate_wave3(hipStream_t st, const uint8_t* in, const void* blocks, uint32_t n_blocks, uint8_t* out, uint32_t* status) {
  if (!n_blocks) return hipSuccess;
  hipLaunchKernelGGL(mkp_inflate_wave3, dim3(n_blocks), dim3(64), 0, st, in, (const MkpBgzfBlock*)blocks, n_blocks, out, status);
  return hipGetLastError();
}

extern "C" hipError_t mkp_launch_inflate_wave2(hipStream_t st, const uint8_t* in, const void* blocks, uint32_t n_blocks, uint8_t* out, uint32_t* status) {
  if (!n_blocks) return hipSuccess;
  hipLaunchKernelGGL(mkp_inflate_wave2, dim3(n_blocks), dim3(64), 0, st, in, (const MkpBgzfBlock*)blocks, n_blocks, out, status);
  return hipGetLastError();
}
