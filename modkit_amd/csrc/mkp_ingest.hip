// BAM ingest on the device, kernels.  The per-item work is in mkp_ingest_dev.hpp (one thread, one chain segment or one
// record; also compiled for the host by the test harness); here are the launches around it, the two exclusive scans that turn counts
// into offsets, and the CRC-32 of the inflated BGZF blocks (htslib checks it on every block the reference reads).
//
// Order on the ingest stream, per shard window (host side: mkp_ingest_host.cpp):
//   inflate (mkp_inflate*.hip) -> mkp_crc32_blocks -> mkp_ingest_count -> mkp_ingest_scan_segs            [host reads n_all]
//   -> mkp_ingest_write -> mkp_ingest_parse -> mkp_ingest_scan_sizes                                       [host reads the array sizes]
//   -> mkp_ingest_pack                                                                                     [host reads headers + digests]
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mkp_ingest_dev.hpp"

struct MkpBgzfBlock { unsigned long long in_off; unsigned long long out_off; uint32_t in_len; uint32_t out_len; };


// One wave per BGZF block: the block's bytes split into 64 slices, every lane runs the byte-wise table CRC over its slice (table in LDS),
// and the slices are joined left to right — crc(A || B) = crc(A) * x^(8 |B|) + crc(B) in the raw (unconditioned) register domain — in six
// butterfly steps.  Lane 0 takes the odd-sized first slice so that every right-hand operand of a join has the same length per level.
// status[i] |= 0x100 on a mismatch with the CRC word stored behind the block's payload.
extern "C" __global__ void __launch_bounds__(256)
mkp_crc32_blocks(const uint8_t* __restrict__ zin, const MkpBgzfBlock* __restrict__ blocks, uint32_t n_blocks, const uint8_t* __restrict__ raw, uint32_t* __restrict__ status) {
  __shared__ uint32_t tab[256];
  { uint32_t c = threadIdx.x; for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ MKP_CRC_POLY : c >> 1; tab[threadIdx.x] = c; }
  __syncthreads();
  const uint32_t bi = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (bi >= n_blocks) return;
  const MkpBgzfBlock bk = blocks[bi];
  const uint32_t len = bk.out_len;
  const uint32_t S = (len / 64u) & ~3u;                 // slice of lanes 1..63
  const uint32_t first = len - 63u * S;                 // lane 0
  const uint8_t* p = raw + bk.out_off + (lane ? first + (lane - 1u) * S : 0u);
  const uint32_t n = lane ? S : first;
  uint32_t c = lane ? 0u : 0xffffffffu;                 // the conditioning (initial all-ones) belongs to the first slice only
  for (uint32_t k = 0; k < n; k++) c = tab[(c ^ p[k]) & 0xffu] ^ (c >> 8);
  // join: at level L lanes with bit L clear hold a left operand whose right neighbour covers S << L bytes
  uint32_t sh = gf2_xpow8n(S);
  for (uint32_t L = 0; L < 6u; L++) {
    const uint32_t other = (uint32_t)__shfl_xor((int)c, 1 << L);
    const bool left = ((lane >> L) & 1u) == 0u;
    const uint32_t joined = gf2_mulmod(left ? c : other, sh) ^ (left ? other : c);
    c = joined;   // both lanes of a pair now hold the pair's CRC; only lanes with the low L+1 bits clear matter from here on
    sh = gf2_mulmod(sh, sh);
  }
  if (lane == 0) {
    uint32_t want; __builtin_memcpy(&want, zin + bk.in_off + bk.in_len, 4);
    if ((c ^ 0xffffffffu) != want) atomicOr(&status[bi], 0x100u);
  }
}

extern "C" __global__ void __launch_bounds__(256)
mkp_ingest_count(const uint8_t* __restrict__ raw, MkpIngestParams P, const MkpSeg* __restrict__ segs, uint32_t* __restrict__ seg_cnt, MkpIngestTotals* tot) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_seg) return;
  seg_cnt[i] = ingest_walk_segment(raw, P.raw_len, segs[i], nullptr, &tot->err);
}

// exclusive scan of seg_cnt[0, n) in place (+ the total behind it), one workgroup; tot->n_all = records of the window
extern "C" __global__ void __launch_bounds__(1024)
mkp_ingest_scan_segs(uint32_t* __restrict__ seg_cnt, uint32_t n, MkpIngestTotals* tot) {
  __shared__ unsigned long long part[1025];
  const uint32_t t = threadIdx.x, chunk = (n + 1023u) / 1024u, lo = min(n, t * chunk), hi = min(n, lo + chunk);
  unsigned long long s = 0; for (uint32_t i = lo; i < hi; i++) s += seg_cnt[i];
  part[t] = s; __syncthreads();
  if (t == 0) { unsigned long long run = 0; for (uint32_t k = 0; k < 1024u; k++) { const unsigned long long v = part[k]; part[k] = run; run += v; } part[1024] = run; }
  __syncthreads();
  unsigned long long run = part[t];
  for (uint32_t i = lo; i < hi; i++) { const uint32_t v = seg_cnt[i]; seg_cnt[i] = (uint32_t)run; run += v; }
  if (t == 0) { const unsigned long long total = part[1024]; if (total > 0xfffffff0ull) { atomicOr(&tot->err, MKP_IE_TABLE); tot->n_all = 0; } else tot->n_all = (uint32_t)total; seg_cnt[n] = (uint32_t)total; }
}

extern "C" __global__ void __launch_bounds__(256)
mkp_ingest_write(const uint8_t* __restrict__ raw, MkpIngestParams P, const MkpSeg* __restrict__ segs, const uint32_t* __restrict__ seg_base, unsigned long long* __restrict__ rec_off, MkpIngestTotals* tot) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_seg) return;
  if ((unsigned long long)seg_base[i + 1] > (unsigned long long)P.rec_cap) { atomicOr(&tot->err, MKP_IE_TABLE); return; }
  ingest_walk_segment(raw, P.raw_len, segs[i], rec_off + seg_base[i], &tot->err);
}

// per record: checks, region test, aux walk; sizes of the packed ones into sz[6][rec_cap] (kept, CIGAR words, chunk pairs, SEQ bytes, ML bytes, sampler-only)
extern "C" __global__ void __launch_bounds__(256)
mkp_ingest_parse(const uint8_t* __restrict__ raw, MkpIngestParams P, const int32_t* __restrict__ parts, const unsigned long long* __restrict__ rec_off, MkpRecInfo* __restrict__ info,
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
      const long long e = (long long)R.pos + (R.reflen > 0 ? R.reflen : 1); extra[2 * (size_t)at + 1] = (int32_t)(e > 0x7fffffffll ? 0x7fffffffll : e); }
  }
}

// exclusive scans of the six size arrays in place, one workgroup; totals into tot
extern "C" __global__ void __launch_bounds__(1024)
mkp_ingest_scan_sizes(uint32_t* __restrict__ sz, uint32_t rec_cap, MkpIngestTotals* tot) {
  __shared__ unsigned long long part[1025];
  const uint32_t n = min(tot->n_all, rec_cap);
  const uint32_t t = threadIdx.x, chunk = (n + 1023u) / 1024u, lo = min(n, t * chunk), hi = min(n, lo + chunk);
  for (uint32_t q = 0; q < 6u; q++) {
    uint32_t* a = sz + (size_t)q * rec_cap;
    unsigned long long s = 0; for (uint32_t i = lo; i < hi; i++) s += a[i];
    __syncthreads();
    part[t] = s; __syncthreads();
    if (t == 0) { unsigned long long run = 0; for (uint32_t k = 0; k < 1024u; k++) { const unsigned long long v = part[k]; part[k] = run; run += v; } part[1024] = run; }
    __syncthreads();
    unsigned long long run = part[t];
    for (uint32_t i = lo; i < hi; i++) { const uint32_t v = a[i]; a[i] = (uint32_t)run; run += v; }
    if (t == 0) {
      const unsigned long long total = part[1024];
      if (total > 0xfffffff0ull) atomicOr(&tot->err, MKP_IE_4G);
      if (q == 0) tot->n_kept = (uint32_t)total; else if (q == 1) tot->cigar_words = total; else if (q == 2) tot->chunk_pairs = total; else if (q == 3) tot->seq_bytes = total; else if (q == 4) tot->ml_bytes = total; else tot->n_sample_only = (uint32_t)total;
    }
  }
}

extern "C" __global__ void __launch_bounds__(256)
mkp_ingest_pack(const uint8_t* __restrict__ raw, uint32_t rec_cap, const MkpRecInfo* __restrict__ info, const uint32_t* __restrict__ sz,
                MkpReadHdr* __restrict__ hdr, uint32_t* __restrict__ cigar, uint32_t* __restrict__ chunk_pfx, uint8_t* __restrict__ seq, MkpTagRef* __restrict__ tagref,
                uint32_t* __restrict__ ranks, uint8_t* __restrict__ ml, MkpRecDigest* __restrict__ dig, MkpIngestTotals* tot) {
  const uint32_t n = min(tot->n_all, rec_cap);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const MkpRecInfo R = info[i];
    if (R.kind != 1 && R.kind != 3) continue;
    const uint32_t j = R.kind == 1 ? sz[i] : tot->n_kept + sz[5 * (size_t)rec_cap + i];   // headers: the kept records in file order, then the sampler-only ones
    ingest_pack_record(raw, R, i, j, sz[(size_t)rec_cap + i], sz[2 * (size_t)rec_cap + i], sz[3 * (size_t)rec_cap + i], sz[4 * (size_t)rec_cap + i],
                       hdr, cigar, chunk_pfx, seq, tagref, ranks, ml, dig, tot);
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
hipError_t mkp_launch_crc32(hipStream_t st, const uint8_t* zin, const void* blocks, uint32_t n_blocks, const uint8_t* raw, uint32_t* status) {
  if (!n_blocks) return hipSuccess;
  hipLaunchKernelGGL(mkp_crc32_blocks, dim3((n_blocks + 3u) / 4u), dim3(256), 0, st, zin, (const MkpBgzfBlock*)blocks, n_blocks, raw, status);
  return hipGetLastError();
}
hipError_t mkp_launch_ingest_count(hipStream_t st, const uint8_t* raw, const MkpIngestParams* P, const MkpSeg* segs, uint32_t* seg_cnt, MkpIngestTotals* tot) {
  if (P->n_seg) hipLaunchKernelGGL(mkp_ingest_count, dim3((P->n_seg + 255u) / 256u), dim3(256), 0, st, raw, *P, segs, seg_cnt, tot);
  hipLaunchKernelGGL(mkp_ingest_scan_segs, dim3(1), dim3(1024), 0, st, seg_cnt, P->n_seg, tot);
  return hipGetLastError();
}
hipError_t mkp_launch_ingest_parse(hipStream_t st, const uint8_t* raw, const MkpIngestParams* P, const int32_t* parts, const MkpSeg* segs, const uint32_t* seg_base, unsigned long long* rec_off,
                                   MkpRecInfo* info, uint32_t* sz, int32_t* extra, MkpIngestTotals* tot) {
  if (P->n_seg) hipLaunchKernelGGL(mkp_ingest_write, dim3((P->n_seg + 255u) / 256u), dim3(256), 0, st, raw, *P, segs, seg_base, rec_off, tot);
  const uint32_t grid = P->rec_cap ? (uint32_t)((P->rec_cap + 255u) / 256u < 8192u ? (P->rec_cap + 255u) / 256u : 8192u) : 1u;
  hipLaunchKernelGGL(mkp_ingest_parse, dim3(grid), dim3(256), 0, st, raw, *P, parts, rec_off, info, sz, extra, tot);
  hipLaunchKernelGGL(mkp_ingest_scan_sizes, dim3(1), dim3(1024), 0, st, sz, P->rec_cap, tot);
  return hipGetLastError();
}
hipError_t mkp_launch_ingest_pack(hipStream_t st, const uint8_t* raw, uint32_t rec_cap, const MkpRecInfo* info, const uint32_t* sz, MkpReadHdr* hdr, uint32_t* cigar,
                                  uint32_t* chunk_pfx, uint8_t* seq, MkpTagRef* tagref, uint32_t* ranks, uint8_t* ml, MkpRecDigest* dig, MkpIngestTotals* tot) {
  const uint32_t grid = rec_cap ? (uint32_t)((rec_cap + 255u) / 256u < 8192u ? (rec_cap + 255u) / 256u : 8192u) : 1u;
  hipLaunchKernelGGL(mkp_ingest_pack, dim3(grid), dim3(256), 0, st, raw, rec_cap, info, sz, hdr, cigar, chunk_pfx, seq, tagref, ranks, ml, dig, tot);
  return hipGetLastError();
}
}
