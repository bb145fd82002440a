// BGZF block inflate, one WAVE per block ("BAM ingest on device"; RFC 1951, SAM spec §4.1).
//
// mkp_inflate_blocks (mkp_inflate.hip) gives every block to one thread: 30 GB/s over a whole file's 54 000 blocks, but ~100 ms for any
// one block — every table probe and every output byte is a global-memory round trip of a single lane — so a shard window of 1 000 blocks
// costs what the whole file costs.  Here a block belongs to a wave and the latency per block is what is optimised:
//   * the decode state (bit buffer, positions) is wave-uniform — scalar registers, scalar ALU;
//   * the compressed bytes sit in registers, one dword per lane (256 B per chunk, the next chunk already in flight), and reach the bit
//     buffer through v_readlane: no memory on the refill path;
//   * the code tables are in LDS: an 11-bit (literal/length) and a 9-bit (distance) direct table built by all lanes, plus the
//     canonical count / symbol lists for the rare longer codes (decoded length by length, RFC 1951 §3.2.2);
//   * the last 32 KiB of output live in an LDS ring, so an LZ77 match (89 % of a BAM's bytes, mean length 10) is ONE step of the whole
//     wave — lane k copies byte k, overlapping matches through k mod dist — instead of a byte loop with a round trip per byte;
//   * the output leaves the ring 16 KiB at a time, 16 bytes per lane: no global store on the symbol path (loads and stores share vmcnt on
//     gfx9: with a store per symbol every wait for the input prefetch also drained the stores, ~0.6 us per symbol).
// LDS: 32 KiB ring + 6.6 KiB tables per wave, one wave per workgroup, four workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mkp_inflate_wave_common.hpp"

struct MkpBgzfBlock { unsigned long long in_off; unsigned long long out_off; uint32_t in_len; uint32_t out_len; };

namespace {

struct WaveLds {
  uint8_t ring[RING];
  uint16_t lit[1u << LIT_BITS];     // [0:3] code length (0 = longer than LIT_BITS or no such code), [4:12] symbol
  uint16_t dist[1u << DIST_BITS];   // [0:3] code length, [4:8] symbol
  uint16_t lcount[16], dcount[16];  // canonical fallback: codes per length ...
  uint16_t lsym[288], dsym[32];     // ... and symbols in canonical order
  uint8_t lens[320];                // code lengths of the block being set up
};

// the compressed bytes of one block, a dword per lane, behind a 64-bit wave-uniform bit buffer
struct Reader {
  const uint8_t* p; uint32_t n;       // block input
  uint32_t w0, w1;                    // lane's dword of the current / the next 256-byte chunk
  uint32_t chunk, widx;               // byte offset of the current chunk, next dword of it to take (uniform)
  unsigned long long buf; uint32_t cnt;   // uniform
  __device__ __forceinline__ uint32_t load_word(uint32_t off) const {   // lane's dword at byte `off`; zero past the end
    if (off + 4u <= n) { uint32_t v; __builtin_memcpy(&v, p + off, 4); return v; }
    uint32_t v = 0; for (uint32_t k = 0; k < 4u; k++) if (off + k < n) v |= (uint32_t)p[off + k] << (8u * k);
    return v;
  }
  __device__ __forceinline__ void seek(uint32_t byte_off, int lane) {
    chunk = byte_off & ~255u;
    w0 = load_word(chunk + 4u * (uint32_t)lane); w1 = load_word(chunk + 256u + 4u * (uint32_t)lane);
    widx = (byte_off & 255u) >> 2;
    const uint32_t sh = 8u * (byte_off & 3u);
    buf = (unsigned long long)((uint32_t)__builtin_amdgcn_readlane((int)w0, (int)widx) >> sh); cnt = 32u - sh; widx++;
    if (widx == 64u) next_chunk(lane);
  }
  __device__ __forceinline__ void next_chunk(int lane) { chunk += 256u; w0 = w1; w1 = load_word(chunk + 256u + 4u * (uint32_t)lane); widx = 0; }
  __device__ __forceinline__ void need32(int lane) {   // at least 32 bits in the buffer
    if (cnt < 32u) {
      buf |= (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)w0, (int)widx) << cnt; cnt += 32u; widx++;
      if (widx == 64u) next_chunk(lane);
    }
  }
  __device__ __forceinline__ uint32_t get(uint32_t k, int lane) {   // k <= 16
    need32(lane);
    const uint32_t v = (uint32_t)buf & ((1u << k) - 1u); buf >>= k; cnt -= k; return v;
  }
  __device__ __forceinline__ unsigned long long bits_taken() const { return 8ull * ((unsigned long long)chunk + 4ull * widx) - cnt; }
};

// canonical decode of a code longer than the direct table (RFC 1951 §3.2.2, as in mkp_inflate.hip); -1 = no such code
__device__ __forceinline__ int slow_sym(Reader& r, const uint16_t* count, const uint16_t* syms, int lane) {
  r.need32(lane);
  uint32_t bits = (uint32_t)r.buf; int code = 0, first = 0, index = 0;
  for (int len = 1; len <= 15; len++) {
    code |= (int)(bits & 1u); bits >>= 1;
    const int c = (int)sgpr(count[len]);
    if (code - c < first) { r.buf >>= len; r.cnt -= (uint32_t)len; return (int)sgpr(syms[index + (code - first)]); }
    index += c; first += c; first <<= 1; code <<= 1;
  }
  return -1;
}

}  // namespace

// status[i]: 0 ok, 1 input exhausted, 2 bad block type / stored length, 3 bad code lengths, 4 bad symbol, 5 distance too far, 6 output size mismatch
extern "C" __global__ void __launch_bounds__(64)
mkp_inflate_wave(const uint8_t* __restrict__ in, const MkpBgzfBlock* __restrict__ blocks, uint32_t n_blocks, uint8_t* __restrict__ out, uint32_t* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) WaveLds L;
  const uint32_t bi = blockIdx.x;
  if (bi >= n_blocks) return;
  const int lane = (int)threadIdx.x;
  const MkpBgzfBlock bk = blocks[bi];
  uint8_t* __restrict__ o = out + bk.out_off;
  const uint32_t cap = bk.out_len;
  Reader r; r.p = in + bk.in_off; r.n = bk.in_len; r.seek(0, lane);
  uint32_t w = 0, err = 0, flushed = 0;   // uniform; output bytes [flushed, w) are in the ring only
  for (uint32_t guard = 0; guard <= bk.in_len && !err; guard++) {
    const uint32_t last = r.get(1, lane), type = r.get(2, lane);
    if (type == 0) {   // stored: byte-aligned LEN / NLEN, then raw bytes, copied by all lanes straight from the input
      const uint32_t drop = r.cnt & 7u; r.buf >>= drop; r.cnt -= drop;
      const uint32_t len = r.get(16, lane), nlen = r.get(16, lane);
      if ((len ^ 0xffffu) != nlen || w + len > cap) { err = 2; break; }
      const unsigned long long taken = r.bits_taken(); const uint32_t at = (uint32_t)(taken >> 3);
      if ((unsigned long long)at + len > bk.in_len) { err = 1; break; }
      LDS_SYNC(); flush(L.ring, o, flushed, w, lane);   // what the ring still owes, then the raw bytes go to both places directly (a stored block may be longer than the ring)
      for (uint32_t k = (uint32_t)lane; k < len; k += 64u) { const uint8_t v = r.p[at + k]; o[w + k] = v; L.ring[(w + k) & (RING - 1u)] = v; }
      w += len; flushed = w; r.seek(at + len, lane);
    } else if (type == 1 || type == 2) {
      int nlen_codes = 288, ndist_codes = 30;
      if (type == 1) {   // fixed codes (§3.2.6)
        for (int s = lane; s < 288; s += 64) L.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
        if (lane < 32) L.lens[288 + lane] = 5;   // 32 five-bit codes, 30 and 31 never valid (checked when one turns up): a complete code
        ndist_codes = 32;
      } else {           // dynamic codes (§3.2.7)
        const int nlen = (int)r.get(5, lane) + 257, ndist = (int)r.get(5, lane) + 1, ncode = (int)r.get(4, lane) + 4;
        if (nlen > 286 || ndist > 30) { err = 3; break; }
        if (lane < 19) L.lens[lane] = 0;   // (lens[0, 19): the code-length code, until it is built)
        LDS_SYNC();
        for (int idx = 0; idx < ncode; idx++) { const uint32_t v = r.get(3, lane); if (lane == 0) L.lens[cl_order(idx)] = (uint8_t)v; }
        LDS_SYNC();
        // the code-length code goes through the distance table's storage (7-bit codes fit its 9-bit direct table); it must be complete
        if (build(L.lens, 19, L.dist, DIST_BITS, L.dcount, L.dsym, lane) != 0) { err = 3; break; }
        __builtin_amdgcn_wave_barrier();
        int idx = 0;   // literal/length and distance code lengths, one run (the code-length code's own lengths in lens[0, 19) are no longer needed)
        while (idx < nlen + ndist) {
          r.need32(lane);
          const uint32_t e = sgpr(L.dist[(uint32_t)r.buf & ((1u << DIST_BITS) - 1u)]);
          if (!(e & 15u)) { err = 4; break; }
          r.buf >>= (e & 15u); r.cnt -= (e & 15u);
          const int sym = (int)(e >> 4);
          if (sym < 16) { if (lane == 0) L.lens[idx] = (uint8_t)sym; idx++; }
          else {
            int len = 0, rep;
            if (sym == 16) { if (idx == 0) { err = 3; break; } LDS_SYNC();
              len = (int)sgpr(L.lens[idx - 1]); rep = 3 + (int)r.get(2, lane); }
            else if (sym == 17) rep = 3 + (int)r.get(3, lane);
            else rep = 11 + (int)r.get(7, lane);
            if (idx + rep > nlen + ndist) { err = 3; break; }
            for (int k = lane; k < rep; k += 64) L.lens[idx + k] = (uint8_t)len;
            idx += rep;
          }
        }
        if (err) break;
        LDS_SYNC();
        // the distance lengths follow the literal/length ones: move them to lens[288, 288 + ndist)
        const uint8_t mine = lane < ndist ? L.lens[nlen + lane] : 0;
        LDS_SYNC();
        if (lane < ndist) L.lens[288 + lane] = mine;
        nlen_codes = nlen; ndist_codes = ndist;
        LDS_SYNC();
        if (sgpr(L.lens[256]) == 0u) { err = 3; break; }   // no end-of-block code
      }
      LDS_SYNC();
      {
        const int e1 = build(L.lens, nlen_codes, L.lit, LIT_BITS, L.lcount, L.lsym, lane);
        uint32_t used1 = 0; for (int l = 1; l <= 15; l++) used1 += sgpr(L.lcount[l]);
        (void)used1; if (e1 != 0) { err = 3; break; }   // an incomplete literal/length code is never valid (the host decoder's and zlib's rule)
        const int e2 = build(L.lens + 288, ndist_codes, L.dist, DIST_BITS, L.dcount, L.dsym, lane);
        uint32_t used2 = 0; for (int l = 1; l <= 15; l++) used2 += sgpr(L.dcount[l]);
        if (e2 < 0 || (e2 > 0 && !(used2 == 1u && sgpr(L.dcount[1]) == 1u))) { err = 3; break; }   // incomplete distance code: only a single one-bit code
      }
      // literal / length + distance symbols until end of block (§3.2.5)
      for (uint32_t g2 = 0; g2 <= cap + 1u; g2++) {   // every symbol but the last emits at least one byte
        r.need32(lane);
        uint32_t e = sgpr(L.lit[(uint32_t)r.buf & ((1u << LIT_BITS) - 1u)]);
        int sym;
        if (e & 15u) { r.buf >>= (e & 15u); r.cnt -= (e & 15u); sym = (int)(e >> 4); }
        else { sym = slow_sym(r, L.lcount, L.lsym, lane); if (sym < 0) { err = 4; break; } }
        if (sym < 256) {
          if (w >= cap) { err = 6; break; }
          if (lane == 0) L.ring[w & (RING - 1u)] = (uint8_t)sym;
          w++;
        } else if (sym == 256) break;
        else {
          const int ls = sym - 257;
          if (ls >= 29) { err = 4; break; }
          const uint32_t len = len_base(ls) + r.get(len_extra(ls), lane);
          r.need32(lane);
          const uint32_t d = sgpr(L.dist[(uint32_t)r.buf & ((1u << DIST_BITS) - 1u)]);
          int ds;
          if (d & 15u) { r.buf >>= (d & 15u); r.cnt -= (d & 15u); ds = (int)(d >> 4); }
          else { ds = slow_sym(r, L.dcount, L.dsym, lane); if (ds < 0) { err = 4; break; } }
          if (ds >= 30) { err = 4; break; }
          const uint32_t dx = dist_extra(ds);
          uint32_t dist = dist_base(ds);
          if (dx) { r.need32(lane); dist += (uint32_t)r.buf & ((1u << dx) - 1u); r.buf >>= dx; r.cnt -= dx; }
          if (dist > w) { err = 5; break; }
          if (w + len > cap) { err = 6; break; }
          // LZ77 copy by the whole wave: byte k of the match comes from byte (k mod dist) of the `dist` bytes before it — all of them written
          // before this match, so every lane's source is final; earlier LDS writes of this wave are ordered before these reads
          LDS_SYNC();
          const uint32_t src0 = w - dist;
          if (dist >= len) {
            for (uint32_t k = (uint32_t)lane; k < len; k += 64u) L.ring[(w + k) & (RING - 1u)] = L.ring[(src0 + k) & (RING - 1u)];
          } else if (dist == 1u) {
            const uint8_t v = L.ring[src0 & (RING - 1u)];
            for (uint32_t k = (uint32_t)lane; k < len; k += 64u) L.ring[(w + k) & (RING - 1u)] = v;
          } else {
            for (uint32_t k = (uint32_t)lane; k < len; k += 64u) L.ring[(w + k) & (RING - 1u)] = L.ring[(src0 + k % dist) & (RING - 1u)];
          }
          w += len;
        }
        // a 16 KiB half of the ring is complete: it goes out in one coalesced sweep, long before the write position comes round to it again
        if ((w & ~(RING / 2u - 1u)) > flushed) { const uint32_t upto = w & ~(RING / 2u - 1u); LDS_SYNC(); flush(L.ring, o, flushed, upto, lane); flushed = upto; }
      }
    } else { err = 2; break; }
    if (err || last) break;
  }
  LDS_SYNC(); flush(L.ring, o, flushed, w, lane);
  if (!err && w != cap) err = 6;
  if (!err && r.bits_taken() > 8ull * bk.in_len) err = 1;
  if (lane == 0) status[bi] = err;
}

extern "C" hipError_t mkp_launch_inflate_wave(hipStream_t st, const uint8_t* in, const MkpBgzfBlock* blocks, uint32_t n_blocks, uint8_t* out, uint32_t* status) {
  if (!n_blocks) return hipSuccess;
  hipLaunchKernelGGL(mkp_inflate_wave, dim3(n_blocks), dim3(64), 0, st, in, blocks, n_blocks, out, status);
  return hipGetLastError();
}
