// BAM ingest on the device: what comes after the BGZF inflate — record boundaries, the per-record checks and the
// region test of htslib's indexed fetch (src/pileup/mod.rs:732-759), the aux walk for MM / ML / MN (src/mod_bam.rs:1388-1470), the
// MM tokeniser (MmTagInfo::parse, src/mod_bam.rs:909-1000) and the packing into the shard arrays of mkp_device.h — so that the
// inflated bytes never leave HBM.  Host counterpart with the same semantics: BamSource::read_chunks / index_record (mkp_bam.hpp) and
// Packer::add / tokenise (mkp_pack.hpp); tests/test_ingest_emul.py runs both over the same BAMs and compares record by record.
//
// Every function here is the work of ONE thread on ONE item (a chain segment, or a record): no cross-lane operations, plain loads
// through __builtin_memcpy (BAM fields are unaligned).  The inflated window is ~25 KB per record; per record the walk touches its core,
// its CIGAR and its aux block once — a few hundred microseconds of device time per shard next to the inflate's tens of
// milliseconds — so the simple shape is kept, and the same source compiles for the host (MKP_INGEST_HOST_SHIM) where the test
// harness drives it thread by thread.
#pragma once
#include <stdint.h>

#include "mkp_device.h"

#ifdef MKP_INGEST_HOST_SHIM
#define MKP_IDEV static inline
#define MKP_ATOMIC_OR(p, v) (*(p) |= (v))
#define MKP_ATOMIC_ADD32(p, v) ([&]() { uint32_t o__ = *(p); *(p) += (v); return o__; }())
#define MKP_ATOMIC_ADD64(p, v) ([&]() { unsigned long long o__ = *(p); *(p) += (v); return o__; }())
#else
#define MKP_IDEV __device__ __forceinline__
#define MKP_ATOMIC_OR(p, v) atomicOr((p), (v))
#define MKP_ATOMIC_ADD32(p, v) atomicAdd((p), (v))
#define MKP_ATOMIC_ADD64(p, v) atomicAdd((p), (unsigned long long)(v))
#endif

// error bits of one ingest (host: DevIngest::check turns them into the Error the host path would throw)
#define MKP_IE_CORRUPT 1u        // "corrupt BAM record" (block_size < 32, fields longer than the record, ids / positions out of range)
#define MKP_IE_TRUNCATED 2u      // a record runs past the inflated window
#define MKP_IE_CHAIN 4u          // a record chain does not land on the next entry point (index does not match the file)
#define MKP_IE_TABLE 8u          // more records than the table holds
#define MKP_IE_QLEN 16u          // "CIGAR query length does not match SEQ length"
#define MKP_IE_SPAN 32u          // a read or its alignment spans 2^26 bases or more
#define MKP_IE_NONASCII 64u      // non-ASCII mod code
#define MKP_IE_CODES 128u        // more than 4 mod codes in one MM tag
#define MKP_IE_TAGS 256u         // more than 8 MM tags in one read
#define MKP_IE_4G 512u           // packed shard exceeds 4 GiB of one array

// records starting in [start, stop - 3); exact: the chain must land on `stop`
struct MkpSeg { unsigned long long start, stop; uint32_t exact, pad; };

struct MkpIngestParams {
  unsigned long long raw_len;   // bytes of the inflated window
  int32_t tid, beg, end;        // region test of the fetch: records of `tid` with pos < end and endpos > beg
  int32_t n_ref;
  uint32_t n_seg, rec_cap;
  // > 1: the fetch is the union of n_parts windows (ascending, disjoint {beg, end} pairs, passed next to the params); beg / end is their hull
  uint32_t n_parts, pad;
};

// one record of the window after mkp_ingest_parse
struct MkpRecInfo {             // 48 B
  unsigned long long core;      // offset of the 32-byte core in the window (block_size sits 4 bytes before it)
  int32_t pos, reflen;
  uint32_t l_seq, bs;           // block_size
  // kind: 0 dropped, 1 kept, 2 span only (supplementary: max-depth guard), 3 sampler-only
  uint16_t n_cigar, flag; uint8_t l_qname, kind, pad0, pad1;
  uint32_t mm, ml, mn;          // offsets of the aux values' TYPE bytes from the core (0 = absent): MM|Mm, ML|Ml, MN
  uint32_t ml_n;                // elements of the ML array when it is B:C
};

// totals of one ingest, device -> host
struct MkpIngestTotals {
  uint32_t err, n_all, n_kept, n_extra;
  unsigned long long cigar_words, chunk_pairs, seq_bytes, ml_bytes;   // capacities of the packed arrays (exclusive-scan totals)
  unsigned long long n_calls, n_ml_used;
  uint32_t n_sample_only, pad;   // records of the region only the threshold sampler takes (QC-fail, no CIGAR): packed behind the kept ones
};

// ---- CRC-32 joins (mkp_crc32_blocks): GF(2) polynomial arithmetic mod the gzip polynomial, bit-reflected operands (bit 31 = x^0)
#define MKP_CRC_POLY 0xedb88320u
MKP_IDEV uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int i = 0; i < 32; i++) { if (a & (0x80000000u >> i)) p ^= b; b = (b & 1u) ? (b >> 1) ^ MKP_CRC_POLY : b >> 1; }
  return p;
}
MKP_IDEV uint32_t gf2_xpow8n(uint32_t n) {   // x^(8 n) mod P
  uint32_t r = 0x80000000u, sq = 0x00800000u;   // 1; x^8
  while (n) { if (n & 1u) r = gf2_mulmod(r, sq); sq = gf2_mulmod(sq, sq); n >>= 1; }
  return r;
}

MKP_IDEV uint32_t ld_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
MKP_IDEV int32_t ld_i32(const uint8_t* p) { int32_t v; __builtin_memcpy(&v, p, 4); return v; }
MKP_IDEV uint16_t ld_u16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }

// ---- BGZF block table (BamSource::ingest_walk_chain_host on the uploaded bytes).  One thread walks one chain: the blocks from a block
// start the index knows to the next one.  Offsets are positions in the uploaded buffer `z` (the window's file ranges, each 64-byte aligned).
// Pass 1 (out == nullptr) counts, pass 2 writes {offset, header bytes, payload bytes, ISIZE}.
// stop: the next known block start (~0: none); ce / ue: block offset and in-block offset of the chunk's end
struct MkpZChain { unsigned long long start, stop, range_end, ce; uint32_t ue, pad; };
struct MkpZBlk { unsigned long long coff; uint32_t hdr, clen, isize, pad; };
#define MKP_ZE_BAD 1u      // not BGZF, or a block without a usable BC field
#define MKP_ZE_CHAIN 2u    // the chain of block sizes misses the block start the index names
#define MKP_ZE_ISIZE 4u    // a block's trailer claims more than 64 KiB (mkp_bgzf_layout)
#define MKP_ZE_RAWCAP 8u   // the inflated window outgrew its buffer: blocks from there on got no room (mkp_bgzf_layout)
MKP_IDEV uint32_t ingest_walk_blocks(const uint8_t* z, const MkpZChain ch, MkpZBlk* out, uint32_t* err) {
  unsigned long long c = ch.start; uint32_t n = 0;
  for (;;) {
    if (c >= ch.stop || c > ch.ce || (c == ch.ce && ch.ue == 0) || c + 18 > ch.range_end) break;
    // (the host reads 600 bytes of header at most)
    const uint8_t* hb = z + c; const unsigned long long hn = ch.range_end - c < 600ull ? ch.range_end - c : 600ull;
    if (hb[0] != 31 || hb[1] != 139 || !(hb[3] & 4)) { MKP_ATOMIC_OR(err, MKP_ZE_BAD); return n; }
    const uint32_t xlen = ld_u16(hb + 10); unsigned long long x = 12; const unsigned long long xe = 12ull + xlen; uint32_t bsize = 0;
      bool found = false;
    if (xe > hn) break;
    while (x + 4 <= xe) { const uint32_t sl = ld_u16(hb + x + 2); if (hb[x] == 'B' && hb[x + 1] == 'C' && sl == 2 && x + 6 <= xe) {
        bsize = (uint32_t)ld_u16(hb + x + 4) + 1u; found = true; } x += 4ull + sl; }
    if (!found || bsize < xlen + 20u) { MKP_ATOMIC_OR(err, MKP_ZE_BAD); return n; }
    const uint32_t hdr = 12u + xlen, clen = bsize - xlen - 20u;
    const unsigned long long next = c + hdr + clen + 8ull;
    if (next > ch.range_end) break;
    if (out) { MkpZBlk b; b.coff = c; b.hdr = hdr; b.clen = clen; b.isize = ld_u32(z + next - 4); b.pad = 0; out[n] = b; }
    n++; c = next;
  }
  if (ch.stop != ~0ull && c != ch.stop && !(c > ch.ce || (c == ch.ce && ch.ue == 0))) MKP_ATOMIC_OR(err, MKP_ZE_CHAIN);
  return n;
}

// ---- record chains.  One thread walks one segment: entry points are record starts the host knows (chunk starts and the BAI's 16 kb
// linear index), `block_size` links the records in between.  Pass 1 (out == nullptr) counts, pass 2 writes the starts.
MKP_IDEV uint32_t ingest_walk_segment(const uint8_t* raw, unsigned long long raw_len, const MkpSeg sg, unsigned long long* out, uint32_t* err) {
  unsigned long long o = sg.start; uint32_t n = 0;
  while (o + 4 <= sg.stop) {
    if (o + 4 > raw_len) break;
    const int32_t bs = ld_i32(raw + o);
    if (bs < 32) { MKP_ATOMIC_OR(err, MKP_IE_CORRUPT); return n; }
    if (o + 4 + (unsigned long long)bs > raw_len) { MKP_ATOMIC_OR(err, MKP_IE_TRUNCATED); return n; }
    if (out) out[n] = o;
    n++; o += 4 + (unsigned long long)bs;
  }
  if (sg.exact && o != sg.stop) MKP_ATOMIC_OR(err, MKP_IE_CHAIN);
  return n;
}

// ---- one record: field checks and reference span (index_record, mkp_bam.hpp), region test and flag mask (BamSource::read_chunks,
// Packer::keep), aux walk (Packer::aux_find_all: first occurrence of each tag reached before any malformed field)
MKP_IDEV bool ingest_overlaps_parts(const int32_t* parts, uint32_t n, long long pos, long long end) {   // overlaps_parts (mkp_bam.hpp)
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if ((long long)parts[2 * mid + 1] > pos) hi = mid; else lo = mid + 1; }
  return lo < n && (long long)parts[2 * lo] < end;
}
MKP_IDEV void ingest_parse_record(const uint8_t* raw, unsigned long long o, const MkpIngestParams& P, const int32_t* parts, MkpRecInfo* out,
    uint32_t* err) {
  MkpRecInfo R; R.core = o + 4; R.kind = 0; R.pad0 = R.pad1 = 0; R.mm = R.ml = R.mn = 0; R.ml_n = 0;
  const uint8_t* c = raw + o + 4;
  const int32_t bs = ld_i32(raw + o); R.bs = (uint32_t)bs;
  const int32_t tid = ld_i32(c); R.pos = ld_i32(c + 4);
  R.l_qname = c[8]; R.n_cigar = ld_u16(c + 12); R.flag = ld_u16(c + 14);
  const int32_t lseq = ld_i32(c + 16); R.l_seq = (uint32_t)lseq; R.reflen = 0;
  const unsigned long long fixed = 32ull + R.l_qname + 4ull * R.n_cigar + ((unsigned long long)(lseq < 0 ? 0
      : lseq) + 1) / 2 + (unsigned long long)(lseq < 0 ? 0 : lseq);
  if (lseq < 0 || fixed > (unsigned long long)bs || tid < -1 || tid >= P.n_ref || R.pos < -1 || R.pos >= 0x7ffffff0) {
    MKP_ATOMIC_OR(err, MKP_IE_CORRUPT); *out = R; return; }
  const uint8_t* cg = c + 32 + R.l_qname; long long rl = 0;
  // M D N = X consume the reference
  for (uint32_t k = 0; k < R.n_cigar; k++) { const uint32_t w = ld_u32(cg + 4 * k); if ((0x18du >> (w & 15u)) & 1u) rl += w >> 4; }
  if ((long long)R.pos + rl > 0x7ffffff0ll) { MKP_ATOMIC_OR(err, MKP_IE_CORRUPT); *out = R; return; }
  R.reflen = (int32_t)rl;
  const long long endpos = (long long)R.pos + (rl > 0 ? rl : 1);
  const bool in_region = tid == P.tid && (long long)R.pos < (long long)P.end && endpos > (long long)P.beg
      && (P.n_parts < 2 || ingest_overlaps_parts(parts, P.n_parts, R.pos, endpos));
  if (!in_region) { *out = R; return; }
  const bool masked = (R.flag & (4u | 256u | 512u | 1024u)) != 0;
  if (!masked && (R.flag & 2048u) && R.n_cigar) { R.kind = 2; *out = R; return; }
  if (masked || (R.flag & 2048u) || lseq <= 0 || R.n_cigar == 0) {
    // QC-fail or CIGAR-less: not in the pileup, but a candidate of the threshold sampler (reads_sampler: only secondary / duplicate /
    // supplementary records are dropped there) — packed like a kept record, behind them
    if (!(R.flag & (4u | 256u | 1024u | 2048u)) && lseq > 0) R.kind = 3; else { *out = R; return; }
  } else R.kind = 1;
  // aux walk
  const uint32_t aux0 = (uint32_t)fixed, aux_n = (uint32_t)bs - aux0;   // offsets from the core
  const uint8_t* a = c + aux0;
  uint32_t at[5] = {0, 0, 0, 0, 0};   // MM Mm ML Ml MN
  uint32_t q = 0; int missing = 5;
  while (q + 3 <= aux_n && missing) {
    const uint8_t ty = a[q + 2]; const uint32_t v = q + 3; unsigned long long len;
    switch (ty) {
      case 'A': case 'c': case 'C': len = 1; break;
      case 's': case 'S': len = 2; break;
      case 'i': case 'I': case 'f': len = 4; break;
      case 'd': len = 8; break;
      // (no terminator: runs past the end, malformed below)
      case 'Z': case 'H': { uint32_t k = v; while (k < aux_n && a[k]) k++; len = (unsigned long long)(k - v) + 1; break; }
      case 'B': { if (v + 5 > aux_n) { q = aux_n; len = 0; missing = -1; break; } const uint8_t st = a[v]; const uint32_t cnt = ld_u32(a + v + 1);
                  const uint32_t es = (st == 'c' || st == 'C') ? 1u : (st == 's' || st == 'S') ? 2u : 4u; len = 5ull + (unsigned long long)es * cnt;
                    break; }
      default: missing = -1; len = 0; break;
    }
    if (missing < 0) break;
    if ((unsigned long long)v + len > aux_n) break;
    if (a[q] == 'M') {
      const uint8_t t1 = a[q + 1];
      const int k = t1 == 'M' ? 0 : t1 == 'm' ? 1 : t1 == 'L' ? 2 : t1 == 'l' ? 3 : t1 == 'N' ? 4 : -1;
      if (k >= 0 && !at[k]) { at[k] = aux0 + q + 2; missing--; }
    }
    q = v + (uint32_t)len;
  }
  R.mm = at[0] ? at[0] : at[1]; R.ml = at[2] ? at[2] : at[3]; R.mn = at[4];   // new style wins, each looked up independently (util.rs:174-188)
  if (R.ml && c[R.ml] == 'B' && c[R.ml + 1] == 'C') R.ml_n = ld_u32(c + R.ml + 2);
  *out = R;
}

// sizes a kept record takes in the packed arrays (scanned into offsets before mkp_ingest_pack)
MKP_IDEV uint32_t ingest_seq_bytes(uint32_t l_seq) { return (((l_seq + 1u) / 2u) + 3u) & ~3u; }
// (a record without a CIGAR is packed with one soft clip over its bases, as Packer::add does)
MKP_IDEV uint32_t ingest_chunk_pairs(uint32_t n_cigar) { return n_cigar ? (n_cigar + 63u) / 64u : 1u; }
MKP_IDEV uint32_t ingest_cigar_words(uint32_t n_cigar) { return n_cigar ? n_cigar : 1u; }

// what the packer's tokeniser leaves for one record
struct MkpTokOut { uint32_t n_tags; uint32_t n_calls; uint32_t ml_used; unsigned long long cap; unsigned long long key_hash; uint32_t sum2; };

// eight text bytes from q on, zero past `lim` (never reads past it)
MKP_IDEV unsigned long long ingest_ld8(const uint8_t* q, const uint8_t* lim) {
  if (q + 8 <= lim) { unsigned long long v; __builtin_memcpy(&v, q, 8); return v; }
  unsigned long long v = 0; for (int k = 0; k < 8 && q + k < lim; k++) v |= (unsigned long long)q[k] << (8 * k);
  return v;
}
MKP_IDEV bool ingest_ws(uint8_t ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r' || ch == '\f' || ch == '\v'; }
MKP_IDEV void fnv_byte(unsigned long long* h, uint8_t b) { *h ^= b; *h *= 1099511628211ull; }
MKP_IDEV void fnv_decimal(unsigned long long* h, uint32_t v) {   // the digits std::to_string(v) would append
  uint8_t d[10]; int n = 0; do { d[n++] = (uint8_t)('0' + v % 10u); v /= 10u; } while (v);
  while (n) fnv_byte(h, d[--n]);
}

// Packer::tokenise (mkp_pack.hpp) for one record: MM header structure -> key hash, delta lists -> cumulative ranks (the ML bytes are moved
// by ingest_copy_record).  ranks: the record's slice (capacity ml_n); tagref: MKP_MAX_TAGS entries; rank_base / ml_base: their offsets in the shard.
// false = the read only contributes coverage (tag error).  `err` collects the conditions the host packer throws on.
MKP_IDEV bool ingest_tokenise(const uint8_t* c, const MkpRecInfo& R, uint32_t* ranks, MkpTagRef* tagref, uint32_t rank_base, uint32_t ml_base,
    MkpTokOut* out, uint32_t* err) {
  out->n_tags = 0; out->n_calls = 0; out->ml_used = 0; out->cap = 0; out->key_hash = 1469598103934665603ull; out->sum2 = 0;
  if (!R.mm || !R.ml) return false;
  if (c[R.mm] != 'Z') return false;
  if (!(c[R.ml] == 'B' && c[R.ml + 1] == 'C')) return false;
  const uint32_t ml_n = R.ml_n;
  if (R.mn) {
    long long v; const uint8_t* m = c + R.mn;
    switch (m[0]) {
      case 'c': v = (int8_t)m[1]; break; case 'C': v = m[1]; break;
      case 's': v = (int16_t)ld_u16(m + 1); break; case 'S': v = ld_u16(m + 1); break;
      case 'i': v = ld_i32(m + 1); break; case 'I': v = ld_u32(m + 1); break;
      default: return false;
    }
    if ((unsigned long long)v != (unsigned long long)R.l_seq) return false;   // check_mn_tag_correct (mod_bam.rs:1431-1449)
  } else if (R.flag & (256u | 1024u | 2048u)) return false;
  const uint8_t* s = c + R.mm + 1;
  const uint8_t* mlp = c + R.ml;
  unsigned long long pointer = 0, calls = 0; bool implicit_strand[2] = {false, false};
  uint32_t n_hdr = 0, n_rank = 0; uint32_t nc[2] = {0, 0};
  while (*s) {
    const uint8_t* e = s; while (*e && *e != ';') e++;
    if (e > s) {
      // ---- header (MmTagInfo::parse, mod_bam.rs:909-983)
      const uint8_t* p = s; const uint8_t* he = s; while (he < e && *he != ',') he++;
      uint32_t fb; bool neg; uint32_t mode = 2; uint32_t codes[MKP_KMAX + 1]; uint32_t n_codes = 0;
      if (p >= he) return false;
      switch (*p) { case 'A': fb = 0; break; case 'C': fb = 1; break; case 'G': fb = 2; break; case 'T': case 'U': fb = 3; break; case 'N': fb = 4;
        break; default: return false; }
      p++; if (p >= he) return false;
      if (*p == '+') neg = false; else if (*p == '-') neg = true; else return false;
      p++; bool chebi = false; uint32_t offset = 2;
      if (p < he && *p >= '0' && *p <= '9') { unsigned long long v = 0; while (p < he && *p >= '0' && *p <= '9') {
          v = v * 10 + (unsigned long long)(*p - '0'); if (v > 0x7fffffffull) return false; p++; offset++; }
        codes[n_codes++] = 0x80000000u | (uint32_t)v; chebi = true; }
      for (; p < he; p++) {
        if (*p == '?' || *p == '.') { mode = *p == '?' ? 0u : 1u; offset++; }
        else if (*p >= '0' && *p <= '9') return false;
        else { if (chebi) return false; if (*p >= 0x80) { MKP_ATOMIC_OR(err, MKP_IE_NONASCII); return false; }
               if (n_codes <= MKP_KMAX) { codes[n_codes] = *p; } n_codes++; if (n_codes > MKP_KMAX + 1) { n_codes = MKP_KMAX + 1; } offset++; }
      }
      if (n_codes > MKP_KMAX) { MKP_ATOMIC_OR(err, MKP_IE_CODES); return false; }
      // ---- delta list -> cumulative ranks (to_positions_specific / to_positions, mod_bam.rs:697-767)
      const uint32_t first_rank = n_rank; uint32_t tn = 0;
      if (offset + 1 <= (uint32_t)(e - s)) {
        // (the hot loop of the tokeniser — a 50 kb read carries ~30 KB of digits and commas: the text is read eight bytes at a time into a
        // register and taken apart there; a byte load per character made the longest read's serial chain the whole kernel)
        const uint8_t* d = s + offset + 1; bool first = true; unsigned long long acc = 0;
        const uint8_t* wb = d; unsigned long long wv = ingest_ld8(d, e);   // window: the eight bytes at wb
#define MKP_CH(q) ((uint8_t)(((q) >= wb && (q) < wb + 8) ? (wv >> (8 * (int)((q) - wb))) : ((wb = (q)), (wv = ingest_ld8((q), e)))))
        for (;;) {
          const uint8_t* save = d;
          if (!first) { if (d >= e || MKP_CH(d) != ',') break; d++; }
          while (d < e && ingest_ws(MKP_CH(d))) d++;
          if (!(d < e && MKP_CH(d) >= '0' && MKP_CH(d) <= '9')) { if (first) return false; d = save; break; }
          unsigned long long v = 0; while (d < e) { const uint8_t ch = MKP_CH(d); if (ch < '0' || ch > '9') break;
            v = v * 10 + (unsigned long long)(ch - '0'); if (v > 0xffffffffull) return false; d++; }
          while (d < e && ingest_ws(MKP_CH(d))) d++;
          acc = first ? v : acc + v + 1;   // sum(d + 1) - 1
          if (acc >= (fb == 4 ? (unsigned long long)R.l_seq : 0xffffffffull)) { if (fb == 4 || acc >= 0xffffffffull) return false; }
          // the host packer stores the whole list and then finds the ML array too short; the answer is the same as soon as it is known
          if (n_codes == 0) return false;                                                  // stride 0 with calls (the reference's chunks(0) would panic)
          if (pointer + ((unsigned long long)tn + 1) * n_codes > ml_n) return false;       // "ML array too short" (mod_bam.rs:1222-1228)
          ranks[n_rank++] = (uint32_t)acc; tn++; first = false;
        }
#undef MKP_CH
      }
      const unsigned long long need = pointer + (unsigned long long)tn * n_codes;
      if (n_hdr < MKP_MAX_TAGS) {
        MkpTagRef tr; tr.rank_off = rank_base + first_rank; tr.n = tn; tr.ml_off = ml_base + (uint32_t)pointer; tr.pad = 0;
        if (n_hdr > 0) {   // same delta list as the tag before (`C+h?,d..;C+m?,d..` as basecallers write them)
          const MkpTagRef& pv = tagref[n_hdr - 1];
          bool same = pv.n == tn; const uint32_t* a = ranks + (pv.rank_off - rank_base); const uint32_t* b = ranks + first_rank;
          for (uint32_t k = 0; same && k < tn; k++) same = a[k] == b[k];
          tr.pad = same ? 1u : 0u;
        }
        tagref[n_hdr] = tr;
        if (n_hdr < 2) nc[n_hdr] = n_codes;
      }
      pointer = need; calls += tn;
      if (mode != 0 && fb != 4) implicit_strand[neg ? 1 : 0] = true;
      fnv_byte(&out->key_hash, (uint8_t)"ACGTN"[fb]); fnv_byte(&out->key_hash, neg ? '-' : '+'); fnv_byte(&out->key_hash, (uint8_t)('0' + mode));
      for (uint32_t k = 0; k < n_codes; k++) { fnv_decimal(&out->key_hash, codes[k]); fnv_byte(&out->key_hash, '/'); }
      fnv_byte(&out->key_hash, ';');
      n_hdr++;
    }
    s = *e ? e + 1 : e;
  }
  if (n_hdr == 0) return false;   // no tags -> ModBaseInfo::is_empty -> NoModifiedBaseInformation
  if (n_hdr > MKP_MAX_TAGS) { MKP_ATOMIC_OR(err, MKP_IE_TAGS); return false; }
  // two tags over one delta list: combine_checked's "> 1.01" test (mod_bam.rs:629-656) as an integer test on the ML bytes — every
  // term is (2q + 1) / 512, so the f32 sum is exact and the test is "the numerators reach 518"
  if (n_hdr == 2 && tagref[1].pad) {
    const uint32_t n = tagref[0].n; const uint8_t* m0 = mlp + 6; const uint8_t* m1 = mlp + 6 + (tagref[1].ml_off - ml_base); bool bad = false;
    for (uint32_t j = 0; j < n && !bad; j++) { uint32_t num = 0; for (uint32_t i = 0; i < nc[0]; i++) num += 2u * m0[j * nc[0] + i] + 1u;
      for (uint32_t i = 0; i < nc[1]; i++) num += 2u * m1[j * nc[1] + i] + 1u;
      bad = num >= 518u; }
    out->sum2 = bad ? 1u : 0u;
  }
  out->n_tags = n_hdr; out->n_calls = (uint32_t)calls; out->ml_used = (uint32_t)pointer;
  out->cap = calls + (unsigned long long)R.l_seq * ((implicit_strand[0] ? 1u : 0u) + (implicit_strand[1] ? 1u : 0u));
  return true;
}

// per-record digest the host plans with (next to the record's MkpReadHdr and tag table)
// name_hash2: a second, independent hash of the read name (128 bits identify a name in the sampler's sets); win_idx: the record's place in the window
// (file order across kept and sampler-only records)
struct MkpRecDigest { unsigned long long name_hash, key_hash, name_hash2, win_idx; };

// Packer::add for one kept record, the serial half (one thread): chunk prefixes of the CIGAR, name hashes, the tags; writes the header with
// the offsets the scan gave.  The bulk copies are ingest_copy_record's.
MKP_IDEV void ingest_pack_record(const uint8_t* raw, const MkpRecInfo& R, uint32_t win_idx, uint32_t j, uint32_t cigar_off, uint32_t chunk_off,
    uint32_t seq_off, uint32_t ml_off,
                                 MkpReadHdr* hdr, uint32_t* chunk_pfx, MkpTagRef* tagref, uint32_t* ranks, MkpRecDigest* dig, MkpIngestTotals* tot) {
  const uint8_t* c = raw + R.core;
  const uint8_t* cg = c + 32 + R.l_qname;
  MkpReadHdr h;
  h.cigar_off = cigar_off; h.chunk_off = chunk_off; h.seq_off = seq_off; h.tag_off = j * MKP_MAX_TAGS;
  long long reflen = 0, qlen = 0;
  for (uint32_t k = 0; k < R.n_cigar; k++) {
    const uint32_t w = ld_u32(cg + 4 * k), op = w & 15u;
    if ((k & 63u) == 0) { chunk_pfx[2 * (chunk_off + (k >> 6))] = (uint32_t)qlen; chunk_pfx[2 * (chunk_off + (k >> 6)) + 1] = (uint32_t)reflen; }
    if ((0x18du >> op) & 1u) reflen += w >> 4;
    if ((0x193u >> op) & 1u) qlen += w >> 4;   // M I S = X consume the query
  }
  // sampler-only record without alignment ops
  if (R.n_cigar == 0) { chunk_pfx[2 * chunk_off] = 0; chunk_pfx[2 * chunk_off + 1] = 0; qlen = R.l_seq; }
  if (qlen != (long long)R.l_seq) MKP_ATOMIC_OR(&tot->err, MKP_IE_QLEN);
  if (qlen >= (1 << 26) || reflen >= (1 << 26)) MKP_ATOMIC_OR(&tot->err, MKP_IE_SPAN);
  h.ref_start = R.pos; h.ref_end = R.pos + (int32_t)reflen; h.l_seq = R.l_seq; h.n_cigar = ingest_cigar_words(R.n_cigar);
  h.flags = (R.flag & 16u) ? MKP_RF_REVERSE : 0u;
  { unsigned long long hh = 1469598103934665603ull, h2 = 0x9e3779b97f4a7c15ull; for (int i = 0; i + 1 < (int)R.l_qname; i++) { hh ^= c[32 + i];
      hh *= 1099511628211ull; h2 = (h2 ^ c[32 + i]) * 0xff51afd7ed558ccdull; h2 ^= h2 >> 29; }
    dig[j].name_hash = hh; dig[j].name_hash2 = h2; dig[j].win_idx = win_idx; }
  MkpTokOut t;
  for (uint32_t k = 0; k < MKP_MAX_TAGS; k++) { MkpTagRef z; z.rank_off = 0; z.n = 0; z.ml_off = 0; z.pad = 0; tagref[h.tag_off + k] = z; }
  const bool ok = ingest_tokenise(c, R, ranks + ml_off, tagref + h.tag_off, ml_off, ml_off, &t, &tot->err);
  if (!ok) { h.flags |= MKP_RF_BAD; t.n_tags = 0; t.cap = 0; t.n_calls = 0; t.ml_used = 0; t.sum2 = 0; t.key_hash = 0; }
  if (t.cap > 0xfffffff0ull) { MKP_ATOMIC_OR(&tot->err, MKP_IE_4G); t.cap = 0; }
  h.n_tags = (uint16_t)t.n_tags; h.layout = 0;
  h.event_off = 0; h.event_cap = (uint32_t)t.cap;
  // pad: bit 0 = the two tags' probabilities of some call add up to more than 1.01 (the planner moves it into flags)
  h.gs0 = 0; h.n_sl = 0; h.cov_off = 0; h.pad = t.sum2;
  dig[j].key_hash = t.key_hash;
  hdr[j] = h;
  if (t.n_calls) MKP_ATOMIC_ADD64(&tot->n_calls, (unsigned long long)t.n_calls);
  if (t.ml_used) MKP_ATOMIC_ADD64(&tot->n_ml_used, (unsigned long long)t.ml_used);
}

// Packer::add for one kept record, the bulk half: CIGAR words, SEQ bytes (zero-padded to a dword), the ML array — by `nlanes` lanes that
// share the record (a wave on the device; the test harness calls it lane after lane).  No lane reads what another wrote.  The whole B:C
// array is moved (the record's slice has room for it; the tags only ever point at the bytes their calls use).
MKP_IDEV void ingest_copy_record(const uint8_t* raw, const MkpRecInfo& R, uint32_t cigar_off, uint32_t seq_off, uint32_t ml_off, uint32_t* cigar,
    uint8_t* seq, uint8_t* ml,
                                 uint32_t lane, uint32_t nlanes) {
  const uint8_t* c = raw + R.core;
  const uint8_t* cg = c + 32 + R.l_qname;
  const uint8_t* sq = cg + 4 * (uint32_t)R.n_cigar;
  for (uint32_t k = lane; k < R.n_cigar; k += nlanes) cigar[cigar_off + k] = ld_u32(cg + 4 * k);
  if (R.n_cigar == 0 && lane == 0) cigar[cigar_off] = (R.l_seq << 4) | 4u;   // one soft clip over the bases, as Packer::add does
  const uint32_t nb = (R.l_seq + 1u) / 2u, nd = ingest_seq_bytes(R.l_seq) / 4u;
  uint32_t* sd = (uint32_t*)(seq + seq_off);   // (seq_off is a multiple of 4: every record's room is)
  for (uint32_t k = lane; k < nd; k += nlanes) {
    uint32_t v;
    if (4u * k + 4u <= nb) v = ld_u32(sq + 4u * k);
    else { v = 0; for (uint32_t b = 0; b < 4u; b++) if (4u * k + b < nb) v |= (uint32_t)sq[4u * k + b] << (8u * b); }
    sd[k] = v;
  }
  if (R.ml && c[R.ml] == 'B' && c[R.ml + 1] == 'C') {
    const uint8_t* mlp = c + R.ml + 6; uint8_t* md = ml + ml_off; const uint32_t n = R.ml_n;
    // dwords where source and destination allow (the destination decides; the source is read unaligned)
    const uint32_t head = (uint32_t)((4u - ((uintptr_t)md & 3u)) & 3u) < n ? (uint32_t)((4u - ((uintptr_t)md & 3u)) & 3u) : n;
    for (uint32_t k = lane; k < head; k += nlanes) md[k] = mlp[k];
    const uint32_t nw = (n - head) / 4u;
    for (uint32_t k = lane; k < nw; k += nlanes) { const uint32_t v = ld_u32(mlp + head + 4u * k); __builtin_memcpy(md + head + 4u * k, &v, 4); }
    for (uint32_t k = head + 4u * nw + lane; k < n; k += nlanes) md[k] = mlp[k];
  }
}
