// BAM ingest on the device, kernels.  The per-item work is in mkp_ingest_dev.hpp (one thread, one chain segment or one
// record; also compiled for the host by the test harness); here are the launches around it, the two exclusive scans that turn counts
// into offsets, and the CRC-32 of the inflated BGZF blocks (htslib checks it on every block the reference reads).
//
// Order on the ingest stream, per shard window (host side: mkp_ingest_host.cpp):
//   inflate (mkp_inflate_wave4.hip) -> mkp_crc32_blocks -> mkp_ingest_count -> mkp_ingest_scan_segs            [host reads n_all]
//   -> mkp_ingest_write -> mkp_ingest_parse -> mkp_ingest_scan_sizes                                       [host reads the array sizes]
//   -> mkp_ingest_pack                                                                                     [host reads headers + digests]
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mkp_ingest_dev.hpp"

struct MkpBgzfBlock { unsigned long long in_off; unsigned long long out_off; uint32_t in_len; uint32_t out_len; };


// CRC-32 of every inflated block against the word behind its payload.  One wave per BGZF block: the block's bytes split into 64 slices, every
// lane runs a slicing-by-4 table CRC over its slice — one dword per step: four independent LDS probes instead of four dependent ones; the
// slice may start at any byte, so dwords are loaded aligned, 16 bytes at a time, and brought into place by v_alignbyte — and the slices
// are joined left to right — crc(A || B) = crc(A) * x^(8 |B|) + crc(B) in the raw (unconditioned) register domain — in six butterfly steps.
// Lane 0 takes the odd-sized first slice so that every right-hand operand of a join has the same length per level; x^(8 S 2^L) mod P for
// every slice length S a block can have comes from a table built at compile time (round 4 spent a third of the kernel raising x to that
// power bit by bit).  status[i] |= 0x100 on a mismatch.
namespace {
constexpr uint32_t cx_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int i = 0; i < 32; i++) { if (a & (0x80000000u >> i)) p ^= b; b = (b & 1u) ? (b >> 1) ^ MKP_CRC_POLY : b >> 1; }
  return p;
}
struct XpowTab { uint32_t v[6][257]; };   // v[L][k] = x^(8 * 4k * 2^L) mod P
constexpr XpowTab make_xpow() {
  XpowTab t{};
  uint32_t step = 0x00000001u;   // x^31 ... the reflected representation has bit 31 = x^0: x^32 = P's low terms; built below from x^8 by squaring
  uint32_t x8 = 0x00800000u, x16 = cx_mulmod(x8, x8); step = cx_mulmod(x16, x16);   // x^32
  for (int L = 0; L < 6; L++) {
    t.v[L][0] = 0x80000000u;   // 1
    for (int k = 1; k <= 256; k++) t.v[L][k] = cx_mulmod(t.v[L][k - 1], step);
    step = cx_mulmod(step, step);
  }
  return t;
}
__device__ const XpowTab kXpow = make_xpow();

// raw CRC register after n bytes from p (any alignment), starting from c; slicing-by-4 tables T[4][256] in LDS
__device__ __forceinline__ uint32_t crc_slice(const uint8_t* __restrict__ p, uint32_t n, uint32_t c, const uint32_t* __restrict__ T) {
  const uintptr_t A = (uintptr_t)p; const uint32_t off = (uint32_t)(A & 3u);
  const uint32_t* __restrict__ q = (const uint32_t*)(A & ~(uintptr_t)3);
  const uint32_t nd = n >> 2;   // whole dwords of the slice
  uint4 w; __builtin_memcpy(&w, q, 16);   // (reads up to 19 bytes past the slice: the inflated window has 64 bytes of slack)
  uint32_t k = 0;
  for (; k + 4u <= nd; k += 4u) {
    uint4 nx; __builtin_memcpy(&nx, q + k + 4u, 16);
    const uint32_t d0 = __builtin_amdgcn_alignbyte(w.y, w.x, off), d1 = __builtin_amdgcn_alignbyte(w.z, w.y, off),
        d2 = __builtin_amdgcn_alignbyte(w.w, w.z, off), d3 = __builtin_amdgcn_alignbyte(nx.x, w.w, off);
    c ^= d0; c = T[768u + (c & 0xffu)] ^ T[512u + ((c >> 8) & 0xffu)] ^ T[256u + ((c >> 16) & 0xffu)] ^ T[c >> 24];
    c ^= d1; c = T[768u + (c & 0xffu)] ^ T[512u + ((c >> 8) & 0xffu)] ^ T[256u + ((c >> 16) & 0xffu)] ^ T[c >> 24];
    c ^= d2; c = T[768u + (c & 0xffu)] ^ T[512u + ((c >> 8) & 0xffu)] ^ T[256u + ((c >> 16) & 0xffu)] ^ T[c >> 24];
    c ^= d3; c = T[768u + (c & 0xffu)] ^ T[512u + ((c >> 8) & 0xffu)] ^ T[256u + ((c >> 16) & 0xffu)] ^ T[c >> 24];
    w = nx;
  }
  // up to three dwords and three bytes left, all inside w and the dword behind it
  const uint32_t e = q[k + 4u];
  const uint32_t r[4] = {__builtin_amdgcn_alignbyte(w.y, w.x, off), __builtin_amdgcn_alignbyte(w.z, w.y, off),
      __builtin_amdgcn_alignbyte(w.w, w.z, off), __builtin_amdgcn_alignbyte(e, w.w, off)};
  uint32_t j = 0;
  for (; k < nd; k++, j++) { const uint32_t d = j == 0 ? r[0] : j == 1 ? r[1] : r[2]; c ^= d;
    c = T[768u + (c & 0xffu)] ^ T[512u + ((c >> 8) & 0xffu)] ^ T[256u + ((c >> 16) & 0xffu)] ^ T[c >> 24]; }
  uint32_t tail = j == 0 ? r[0] : j == 1 ? r[1] : j == 2 ? r[2] : r[3];
  for (uint32_t b = 0; b < (n & 3u); b++) { c = T[(c ^ tail) & 0xffu] ^ (c >> 8); tail >>= 8; }
  return c;
}
}  // namespace

extern "C" __global__ void __launch_bounds__(256)
mkp_crc32_blocks(const uint8_t* __restrict__ zin, const MkpBgzfBlock* __restrict__ blocks, uint32_t n_blocks, const uint8_t* __restrict__ raw,
    uint32_t* __restrict__ status) {
  __shared__ uint32_t T[1024];   // T[0..255]: the byte table; T[256 k + i] = the CRC register after byte i followed by k zero bytes
  { uint32_t c = threadIdx.x; for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ MKP_CRC_POLY : c >> 1; T[threadIdx.x] = c; }
  __syncthreads();
  for (uint32_t k = 1; k < 4u; k++) { const uint32_t v = T[256u * (k - 1u) + threadIdx.x]; T[256u * k + threadIdx.x] = (v >> 8) ^ T[v & 0xffu];
    __syncthreads(); }
  const uint32_t bi = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (bi >= n_blocks) return;
  const MkpBgzfBlock bk = blocks[bi];
  const uint32_t len = bk.out_len;
  const uint32_t S = (len / 64u) & ~3u;                 // slice of lanes 1..63 (<= 1024)
  const uint32_t first = len - 63u * S;                 // lane 0
  // the conditioning (initial all-ones) belongs to the first slice only
  uint32_t c = crc_slice(raw + bk.out_off + (lane ? first + (lane - 1u) * S : 0u), lane ? S : first, lane ? 0u : 0xffffffffu, T);
  // join: at level L lanes with bit L clear hold a left operand whose right neighbour covers S << L bytes
  for (uint32_t L = 0; L < 6u; L++) {
    const uint32_t sh = kXpow.v[L][S >> 2];
    const uint32_t other = (uint32_t)__shfl_xor((int)c, 1 << L);
    const bool left = ((lane >> L) & 1u) == 0u;
    // both lanes of a pair now hold the pair's CRC; only lanes with the low L+1 bits clear matter from here on
    c = gf2_mulmod(left ? c : other, sh) ^ (left ? other : c);
  }
  if (lane == 0) {
    uint32_t want; __builtin_memcpy(&want, zin + bk.in_off + bk.in_len, 4);
    if ((c ^ 0xffffffffu) != want) atomicOr(&status[bi], 0x100u);
  }
}

// Exclusive scan of a[0, n) in place by one 1024-thread workgroup; returns the total to every thread.  Tiles of 4096 elements, four
// consecutive ones per thread (coalesced), a shuffle scan per wave, the 16 wave totals through LDS, a running 64-bit carry.  (Round 4 gave
// every thread one contiguous chunk — 64 different cache lines per load instruction — and let thread 0 add the 1024 partial sums.)
__device__ __forceinline__ unsigned long long block_scan_inplace(uint32_t* __restrict__ a, uint32_t n) {
  __shared__ unsigned long long wtot[16];
  __shared__ unsigned long long tile_total;
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  unsigned long long carry = 0;
  for (uint32_t base = 0; base < n; base += 4096u) {
    const uint32_t i0 = base + 4u * t;
    uint32_t v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) v[k] = i0 + k < n ? a[i0 + k] : 0u;
    const unsigned long long mine = (unsigned long long)v[0] + v[1] + v[2] + v[3];
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(incl, d); if ((int)lane >= d) incl += o; }
    if (lane == 63u) wtot[wv] = incl;
    __syncthreads();
    if (t == 0) { unsigned long long run = 0; for (uint32_t k = 0; k < 16u; k++) { const unsigned long long x = wtot[k]; wtot[k] = run; run += x;
      } tile_total = run; }
    __syncthreads();
    unsigned long long run = carry + wtot[wv] + incl - mine;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) { if (i0 + k < n) a[i0 + k] = (uint32_t)run; run += v[k]; }
    carry += tile_total;
    __syncthreads();
  }
  return carry;
}

// BGZF block table of the uploaded window: one thread per chain (ingest_walk_blocks), counts -> offsets -> {offset, header, payload, ISIZE}
extern "C" __global__ void __launch_bounds__(256)
mkp_bgzf_chain_count(const uint8_t* __restrict__ z, const MkpZChain* __restrict__ chains, uint32_t n, uint32_t* __restrict__ cnt, uint32_t* err) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cnt[i] = ingest_walk_blocks(z, chains[i], nullptr, err);
}
extern "C" __global__ void __launch_bounds__(1024)
mkp_bgzf_chain_scan(uint32_t* __restrict__ cnt, uint32_t n, uint32_t* err) {
  const unsigned long long total = block_scan_inplace(cnt, n);
  if (threadIdx.x == 0) { if (total > 0xfffffff0ull) { atomicOr(err, MKP_ZE_BAD); cnt[n] = 0; } else cnt[n] = (uint32_t)total; }
}
extern "C" __global__ void __launch_bounds__(256)
mkp_bgzf_chain_write(const uint8_t* __restrict__ z, const MkpZChain* __restrict__ chains, uint32_t n, const uint32_t* __restrict__ base, uint32_t cap,
    MkpZBlk* __restrict__ out, uint32_t* err) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (base[i + 1] > cap) { atomicOr(err, MKP_ZE_BAD); return; }
  ingest_walk_blocks(z, chains[i], out + base[i], err);
}

// the inflate's block table of one upload stage from its MkpZBlk entries: payload offsets and sizes, and each block's place in the inflated
// window — an exclusive scan of ISIZE behind *raw_cursor, which moves on by the stage's total.  One workgroup.  A block that claims more than
// 64 KiB, or a window that would not fit raw_cap, raises the error bits (the host then falls back to an exact allocation).
extern "C" __global__ void __launch_bounds__(1024)
mkp_bgzf_layout(const MkpZBlk* __restrict__ zb, uint32_t n, unsigned long long* raw_cursor, unsigned long long raw_cap,
    MkpBgzfBlock* __restrict__ out, uint32_t* err) {
  __shared__ unsigned long long wtot[16];
  __shared__ unsigned long long tile_total;
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  const unsigned long long base0 = *raw_cursor;
  unsigned long long carry = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += 1024u) {
    const uint32_t i = b0 + t;
    MkpZBlk z; z.coff = 0; z.hdr = 0; z.clen = 0; z.isize = 0; z.pad = 0; if (i < n) z = zb[i];
    if (z.isize > 65536u) { atomicOr(err, MKP_ZE_ISIZE); z.isize = 0; }
    unsigned long long incl = z.isize;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(incl, d); if ((int)lane >= d) incl += o; }
    if (lane == 63u) wtot[wv] = incl;
    __syncthreads();
    if (t == 0) { unsigned long long run = 0; for (uint32_t k = 0; k < 16u; k++) { const unsigned long long x = wtot[k]; wtot[k] = run; run += x;
      } tile_total = run; }
    __syncthreads();
    if (i < n) {
      MkpBgzfBlock b; b.in_off = z.coff + z.hdr; b.out_off = base0 + carry + wtot[wv] + incl - z.isize; b.in_len = z.clen; b.out_len = z.isize;
      // a block that would end behind the window's capacity gets no room: the inflate and the CRC of this stage write and read nothing for it
      // (both are bounded by out_len); the host sees MKP_ZE_RAWCAP below and inflates the whole window again into an exact allocation
      if (b.out_off + z.isize + 64ull > raw_cap) { b.out_off = 0; b.out_len = 0; }
      out[i] = b;
    }
    carry += tile_total;
    __syncthreads();
  }
  if (t == 0) { if (base0 + carry + 64ull > raw_cap) atomicOr(err, MKP_ZE_RAWCAP); *raw_cursor = base0 + carry; }
}

extern "C" __global__ void __launch_bounds__(256)
mkp_ingest_count(const uint8_t* __restrict__ raw, MkpIngestParams P, const MkpSeg* __restrict__ segs, uint32_t* __restrict__ seg_cnt,
    MkpIngestTotals* tot) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_seg) return;
  seg_cnt[i] = ingest_walk_segment(raw, P.raw_len, segs[i], nullptr, &tot->err);
}

// exclusive scan of seg_cnt[0, n) in place (+ the total behind it), one workgroup; tot->n_all = records of the window
extern "C" __global__ void __launch_bounds__(1024)
mkp_ingest_scan_segs(uint32_t* __restrict__ seg_cnt, uint32_t n, MkpIngestTotals* tot) {
  const unsigned long long total = block_scan_inplace(seg_cnt, n);
  if (threadIdx.x == 0) { if (total > 0xfffffff0ull) { atomicOr(&tot->err, MKP_IE_TABLE); tot->n_all = 0; } else tot->n_all = (uint32_t)total;
    seg_cnt[n] = (uint32_t)total; }
}

extern "C" __global__ void __launch_bounds__(256)
mkp_ingest_write(const uint8_t* __restrict__ raw, MkpIngestParams P, const MkpSeg* __restrict__ segs, const uint32_t* __restrict__ seg_base,
    unsigned long long* __restrict__ rec_off, MkpIngestTotals* tot) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_seg) return;
  if ((unsigned long long)seg_base[i + 1] > (unsigned long long)P.rec_cap) { atomicOr(&tot->err, MKP_IE_TABLE); return; }
  ingest_walk_segment(raw, P.raw_len, segs[i], rec_off + seg_base[i], &tot->err);
}

// per record: checks, region test, aux walk; sizes of the packed ones into sz[6][rec_cap] (kept, CIGAR words, chunk pairs, SEQ bytes, ML bytes,
// sampler-only)
extern "C" __global__ void __launch_bounds__(256)
mkp_ingest_parse(const uint8_t* __restrict__ raw, MkpIngestParams P, const int32_t* __restrict__ parts,
    const unsigned long long* __restrict__ rec_off, MkpRecInfo* __restrict__ info,
                 uint32_t* __restrict__ sz, int32_t* __restrict__ extra, MkpIngestTotals* tot) {
  const uint32_t n = min(tot->n_all, P.rec_cap);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    MkpRecInfo R; ingest_parse_record(raw, rec_off[i], P, parts, &R, &tot->err);
    info[i] = R;
    const bool k = R.kind == 1, pk = R.kind == 1 || R.kind == 3;   // kept by the pileup; packed (the sampler-only records go behind the kept ones)
    sz[i] = k ? 1u : 0u;
    sz[(size_t)P.rec_cap + i] = pk ? ingest_cigar_words(R.n_cigar) : 0u;
    sz[2 * (size_t)P.rec_cap + i] = pk ? ingest_chunk_pairs(R.n_cigar) : 0u;
    sz[3 * (size_t)P.rec_cap + i] = pk ? ingest_seq_bytes(R.l_seq) : 0u;
    sz[4 * (size_t)P.rec_cap + i] = pk ? R.ml_n : 0u;
    sz[5 * (size_t)P.rec_cap + i] = R.kind == 3 ? 1u : 0u;
    if (R.kind == 2) { const uint32_t at = atomicAdd(&tot->n_extra, 1u); extra[2 * (size_t)at] = R.pos;
      const long long e = (long long)R.pos + (R.reflen > 0 ? R.reflen : 1);
        extra[2 * (size_t)at + 1] = (int32_t)(e > 0x7fffffffll ? 0x7fffffffll : e); }
  }
}

// exclusive scans of the six size arrays in place, one workgroup per array; totals into tot
extern "C" __global__ void __launch_bounds__(1024)
mkp_ingest_scan_sizes(uint32_t* __restrict__ sz, uint32_t rec_cap, MkpIngestTotals* tot) {
  const uint32_t n = min(tot->n_all, rec_cap), q = blockIdx.x;
  const unsigned long long total = block_scan_inplace(sz + (size_t)q * rec_cap, n);
  if (threadIdx.x == 0) {
    if (total > 0xfffffff0ull) atomicOr(&tot->err, MKP_IE_4G);
    if (q == 0) tot->n_kept = (uint32_t)total; else if (q == 1) tot->cigar_words = total; else if (q == 2) tot->chunk_pairs = total;
      else if (q == 3) tot->seq_bytes = total;
      else if (q == 4) tot->ml_bytes = total;
      else tot->n_sample_only = (uint32_t)total;
  }
}

// the serial half of the packing, one thread per record: chunk prefixes, name hashes, the MM tokeniser, header and digest
extern "C" __global__ void __launch_bounds__(256)
mkp_ingest_pack(const uint8_t* __restrict__ raw, uint32_t rec_cap, const MkpRecInfo* __restrict__ info, const uint32_t* __restrict__ sz,
                MkpReadHdr* __restrict__ hdr, uint32_t* __restrict__ chunk_pfx, MkpTagRef* __restrict__ tagref,
                uint32_t* __restrict__ ranks, MkpRecDigest* __restrict__ dig, MkpIngestTotals* tot) {
  const uint32_t n = min(tot->n_all, rec_cap);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const MkpRecInfo R = info[i];
    if (R.kind != 1 && R.kind != 3) continue;
    // headers: the kept records in file order, then the sampler-only ones
    const uint32_t j = R.kind == 1 ? sz[i] : tot->n_kept + sz[5 * (size_t)rec_cap + i];
    ingest_pack_record(raw, R, i, j, sz[(size_t)rec_cap + i], sz[2 * (size_t)rec_cap + i], sz[3 * (size_t)rec_cap + i], sz[4 * (size_t)rec_cap + i],
                       hdr, chunk_pfx, tagref, ranks, dig, tot);
  }
}

// the bulk half, one wave per record: CIGAR words, SEQ and ML bytes, a dword per lane and step (round 4 moved them byte by byte inside
// the thread above: the longest read's 25 000 SEQ bytes were the kernel's 10 ms)
extern "C" __global__ void __launch_bounds__(256)
mkp_ingest_copy(const uint8_t* __restrict__ raw, uint32_t rec_cap, const MkpRecInfo* __restrict__ info, const uint32_t* __restrict__ sz,
                uint32_t* __restrict__ cigar, uint8_t* __restrict__ seq, uint8_t* __restrict__ ml, const MkpIngestTotals* tot) {
  const uint32_t n = min(tot->n_all, rec_cap), lane = threadIdx.x & 63u, waves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += waves) {
    const MkpRecInfo R = info[i];
    if (R.kind != 1 && R.kind != 3) continue;
    ingest_copy_record(raw, R, sz[(size_t)rec_cap + i], sz[3 * (size_t)rec_cap + i], sz[4 * (size_t)rec_cap + i], cigar, seq, ml, lane, 64u);
  }
}

// u32 counters -> u64 (the threshold histograms count in 32 bits per GPU; their sum over the ranks of a node may not fit)
extern "C" __global__ void __launch_bounds__(256)
mkp_widen_u32_u64(const uint32_t* __restrict__ in, unsigned long long* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = in[i];
}

extern "C" {
hipError_t mkp_launch_widen(hipStream_t st, const uint32_t* in, unsigned long long* out, uint32_t n) {
  if (n) hipLaunchKernelGGL(mkp_widen_u32_u64, dim3((n + 255u) / 256u), dim3(256), 0, st, in, out, n);
  return hipGetLastError();
}
hipError_t mkp_launch_bgzf_chain_count(hipStream_t st, const uint8_t* z, const MkpZChain* chains, uint32_t n, uint32_t* cnt, uint32_t* err) {
  if (n) hipLaunchKernelGGL(mkp_bgzf_chain_count, dim3((n + 255u) / 256u), dim3(256), 0, st, z, chains, n, cnt, err);
  hipLaunchKernelGGL(mkp_bgzf_chain_scan, dim3(1), dim3(1024), 0, st, cnt, n, err);
  return hipGetLastError();
}
hipError_t mkp_launch_bgzf_chain_write(hipStream_t st, const uint8_t* z, const MkpZChain* chains, uint32_t n, const uint32_t* base, uint32_t cap,
    MkpZBlk* out, uint32_t* err) {
  if (n) hipLaunchKernelGGL(mkp_bgzf_chain_write, dim3((n + 255u) / 256u), dim3(256), 0, st, z, chains, n, base, cap, out, err);
  return hipGetLastError();
}
hipError_t mkp_launch_bgzf_layout(hipStream_t st, const MkpZBlk* zb, uint32_t n, unsigned long long* raw_cursor, unsigned long long raw_cap,
    void* out, uint32_t* err) {
  hipLaunchKernelGGL(mkp_bgzf_layout, dim3(1), dim3(1024), 0, st, zb, n, raw_cursor, raw_cap, (MkpBgzfBlock*)out, err);
  return hipGetLastError();
}
hipError_t mkp_launch_crc32(hipStream_t st, const uint8_t* zin, const void* blocks, uint32_t n_blocks, const uint8_t* raw, uint32_t* status) {
  if (!n_blocks) return hipSuccess;
  hipLaunchKernelGGL(mkp_crc32_blocks, dim3((n_blocks + 3u) / 4u), dim3(256), 0, st, zin, (const MkpBgzfBlock*)blocks, n_blocks, raw, status);
  return hipGetLastError();
}
hipError_t mkp_launch_ingest_count(hipStream_t st, const uint8_t* raw, const MkpIngestParams* P, const MkpSeg* segs, uint32_t* seg_cnt,
    MkpIngestTotals* tot) {
  if (P->n_seg) hipLaunchKernelGGL(mkp_ingest_count, dim3((P->n_seg + 255u) / 256u), dim3(256), 0, st, raw, *P, segs, seg_cnt, tot);
  hipLaunchKernelGGL(mkp_ingest_scan_segs, dim3(1), dim3(1024), 0, st, seg_cnt, P->n_seg, tot);
  return hipGetLastError();
}
hipError_t mkp_launch_ingest_parse(hipStream_t st, const uint8_t* raw, const MkpIngestParams* P, const int32_t* parts, const MkpSeg* segs,
    const uint32_t* seg_base, unsigned long long* rec_off,
                                   MkpRecInfo* info, uint32_t* sz, int32_t* extra, MkpIngestTotals* tot) {
  if (P->n_seg) hipLaunchKernelGGL(mkp_ingest_write, dim3((P->n_seg + 255u) / 256u), dim3(256), 0, st, raw, *P, segs, seg_base, rec_off, tot);
  const uint32_t grid = P->rec_cap ? (uint32_t)((P->rec_cap + 255u) / 256u < 8192u ? (P->rec_cap + 255u) / 256u : 8192u) : 1u;
  hipLaunchKernelGGL(mkp_ingest_parse, dim3(grid), dim3(256), 0, st, raw, *P, parts, rec_off, info, sz, extra, tot);
  hipLaunchKernelGGL(mkp_ingest_scan_sizes, dim3(6), dim3(1024), 0, st, sz, P->rec_cap, tot);
  return hipGetLastError();
}
hipError_t mkp_launch_ingest_pack(hipStream_t st, const uint8_t* raw, uint32_t rec_cap, const MkpRecInfo* info, const uint32_t* sz, MkpReadHdr* hdr,
    uint32_t* cigar,
                                  uint32_t* chunk_pfx, uint8_t* seq, MkpTagRef* tagref, uint32_t* ranks, uint8_t* ml, MkpRecDigest* dig,
                                      MkpIngestTotals* tot) {
  const uint32_t grid = rec_cap ? (uint32_t)((rec_cap + 255u) / 256u < 8192u ? (rec_cap + 255u) / 256u : 8192u) : 1u;
  hipLaunchKernelGGL(mkp_ingest_pack, dim3(grid), dim3(256), 0, st, raw, rec_cap, info, sz, hdr, chunk_pfx, tagref, ranks, dig, tot);
  const uint32_t cgrid = rec_cap ? (uint32_t)((rec_cap + 3u) / 4u < 16384u ? (rec_cap + 3u) / 4u : 16384u) : 1u;
  hipLaunchKernelGGL(mkp_ingest_copy, dim3(cgrid), dim3(256), 0, st, raw, rec_cap, info, sz, cigar, seq, ml, tot);
  return hipGetLastError();
}
}
