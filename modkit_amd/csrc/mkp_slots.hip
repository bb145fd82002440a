// gfx950 (CDNA4, wave64) kernels of the SLOT PIPELINE: the modkit-pileup hot path for runs with focus positions (--cpg, --motif,
// --include-bed — FocusPositions::{Motif, MotifCombineStrands, Regions}, interval_chunks.rs:32-59).  A "slot" is a focus position;
// the host numbers the window's slots in genome order (slot_pos[g] = position of global slot g) and gives every read the slot
// range [gs0, gs0 + n_sl) of its reference span.  One device pass:
//
//   mkp_decode_slots[_long]  one wave per read, reads whose MM tags form one explicit-mode ('?') group with one shared delta list
//                         (`C+m?`, `C+hm?`, `C+h?;C+m?` as basecallers write them), no edge filter.  The walk is driven by the
//                         read's SLOTS, not by its calls: the SEQ is swept once into LDS (a flag bit per base "is the fundamental
//                         base" + a running count per 32 bases), the delta list is marked into a bitmap over the base's
//                         occurrences, and then, 64 slots per step: CIGAR (128-op register window, reference -> query), the base,
//                         its occurrence number (2 LDS reads), whether that occurrence is listed and as which call (2 LDS reads),
//                         ML -> f32 probabilities -> MultipleThresholdModCaller::call, and ONE FEATURE BYTE per slot goes to the
//                         read's run of the feature stream.  Calls on non-focus positions are never located (their rows would be
//                         dropped, pileup/mod.rs:570-604); SEQ and CIGAR are read once per pass.
//   mkp_cover_reads       every other read (implicit-mode / multi-group / duplex / `N` tags / failed tags, or any read when an edge
//                         filter is set): the decode kernels of mkp_kernels.hip leave position-sorted call events; this kernel walks
//                         the read's slots (coverage features) and merges the events into the stream.
//   mkp_pileup_stream     accumulate + emit: one workgroup per tile of <= 1024 slots; the tile's reads' feature bytes are a
//                         position-implicit stream (byte k of a read = global slot gs0 + k): four bytes per lane, one LDS atomic per
//                         feature on 16-bit-packed strand tallies; observed codes as difference arrays; rows straight from LDS
//                         (mkp_dev_rows.hpp).  This is the packed-event-stream histogram of BASELINE.json's north_star.
//   (mkp_scan_tiles + mkp_gather_rows of mkp_kernels.hip order the row runs.)
//
// Semantics follow /root/reference/src (cited inline).  f32 arithmetic is the reference's (contraction off, IEEE division).
#include <cstdlib>

#include "mkp_dev_common.hpp"
#include "mkp_dev_rows.hpp"

#define SL_WB MKP_SLOT_WB      // stored bases per base window of the fused decoder
#define SL_FW (SL_WB / 32u)    // words of the window's flag bitmaps

// per-wave LDS of mkp_decode_slots*: F = "base is the fundamental base" (bit = nibble index inside the dword, dword k of a word at
// bits 8k..), P = occurrences before the word (inside the window), B = "occurrence is listed" over the window's occurrences,
// WP = listed occurrences before the B word
// ck_* = the read's caller constants per code (integer pass threshold, offset and stride of its ML bytes, counter of Modified(code)):
// uniform, but the kernel is short of
// scalar registers — every lane reads them back as vectors once per slot batch
struct SlotLds { int32_t ck_thr[4]; uint32_t ck_off[4]; uint32_t ck_str[4]; uint32_t ck_cid[4]; uint32_t F[SL_FW]; uint32_t B[SL_FW + 2]; uint16_t P[SL_FW]; uint16_t WP[SL_FW + 2]; };

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
// flag word of a SEQ dword: bit 4n set where nibble n equals the BAM code in `pat`
__device__ __forceinline__ uint32_t nib_eq(uint32_t x, uint32_t pat) { uint32_t t = x ^ pat; t |= t >> 1; t |= t >> 2; return ~t & 0x11111111u; }
// the eight flags (bits 4n) gathered into bits n
__device__ __forceinline__ uint32_t gather8(uint32_t t) { return (((t | (t >> 3)) & 0x03030303u) * 0x01041040u) >> 24; }
__device__ __forceinline__ uint32_t feat(uint32_t cid, uint32_t tally) { return cid | (tally << 5); }

#define SLOT_PARAMS(PRM) const MkpReadHdr* __restrict__ hdrs, uint32_t n_reads, const uint32_t* __restrict__ read_ids, const uint32_t* __restrict__ cigar, \
    const uint8_t* __restrict__ seqs, const MkpTagRef* __restrict__ tagref, const uint32_t* __restrict__ ranks, const uint8_t* __restrict__ ml, \
    const MkpLayout* __restrict__ layouts, const MkpFusedDesc* __restrict__ fdesc, PRM prm, const uint32_t* __restrict__ slot_pos, uint8_t* __restrict__ cov, MkpVisit* __restrict__ visits, \
    MkpEvent* __restrict__ events, MkpReadOut* __restrict__ readout, uint32_t* __restrict__ dev_err
#define SLOT_PASS hdrs, n_reads, read_ids, cigar, seqs, tagref, ranks, ml, layouts, fdesc, prm, slot_pos, cov, visits, events, readout, dev_err

// The CIGAR as the slot walk sees it: 256 ops per window (four per lane), reference -> query.  re = inclusive reference end of the
// lane's four ops (window-relative), m1..m3 = where its second..fourth op start, pk[j] = ((query start - (reference start -
// ref_start)) << 2) | kind (0 match: M = X, 1 deletion, 2 ref-skip / nothing) of op j.
struct RefWin {
  uint32_t c0, q_run, Rtot, Qtot, re, m1, m2, m3, pk[4]; int32_t r_run; uint4 pref; bool loaded;
};
// loads as `uniform base + 32-bit byte offset`: the address is one VALU instruction (global saddr form), not 64-bit arithmetic
template <class T> __device__ __forceinline__ T ldo(const void* __restrict__ base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off); }
// four CIGAR words per lane; the address is clamped to the read's last op (the buffer has slack behind it), ops past the end read as 0H
__device__ __forceinline__ uint4 cigar_quad(const uint32_t* __restrict__ cg, uint32_t n_cigar, uint32_t c) {
  const uint32_t k = c + 4u * (uint32_t)lane_id();
  uint4 r = ldo<uint4>(cg, 4u * min(k, n_cigar - 1u));
  if (__any(k + 4u > n_cigar)) { if (k >= n_cigar) r.x = 5u; if (k + 1u >= n_cigar) r.y = 5u; if (k + 2u >= n_cigar) r.z = 5u;
    if (k + 3u >= n_cigar) r.w = 5u;
    }
  return r;
}
// per op code (MIDNSHP=X): bit 0 consumes query, bit 1 consumes reference, bits 2-3 kind
#define MKP_CIGAR_LUT 0x888888833889a693ull
__device__ __forceinline__ void refwin_load(RefWin& w, const uint32_t* __restrict__ cg, uint32_t n_cigar, int32_t ref_start) {
  const uint4 v = w.pref;
  w.pref = cigar_quad(cg, n_cigar, w.c0 + 256u);   // requested one window ahead
  const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
  uint32_t ql[4], rl[4], kind[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t f = (uint32_t)(MKP_CIGAR_LUT >> ((wd[j] & 15u) << 2)), len = wd[j] >> 4;
    ql[j] = len & (0u - (f & 1u)); rl[j] = len & (0u - ((f >> 1) & 1u)); kind[j] = (f >> 2) & 3u;
  }
  const uint32_t qsum = ql[0] + ql[1] + ql[2] + ql[3], rsum = rl[0] + rl[1] + rl[2] + rl[3];
  uint32_t qe;
  if (!__any((v.x | v.y | v.z | v.w) >= (128u << 4))) {   // ops shorter than 128: both running sums fit 16 bits, one scan serves both
    const uint32_t sc = wave_incl_scan(qsum | (rsum << 16));
    qe = sc & 0xffffu; w.re = sc >> 16;
  } else { qe = wave_incl_scan(qsum); w.re = wave_incl_scan(rsum); }
  w.Qtot = (uint32_t)__builtin_amdgcn_readlane((int)qe, 63); w.Rtot = (uint32_t)__builtin_amdgcn_readlane((int)w.re, 63);
  uint32_t qs = w.q_run + qe - qsum, rs = w.re - rsum;   // query offset / window-relative reference offset of the lane's first op
  const int32_t rbase = w.r_run - ref_start;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    w.pk[j] = ((uint32_t)((int32_t)qs - (rbase + (int32_t)rs)) << 2) | kind[j];
    qs += ql[j]; rs += rl[j];
    if (j == 0) w.m1 = rs; else if (j == 1) w.m2 = rs; else if (j == 2) w.m3 = rs;
  }
  w.loaded = true;
}
// (kind, query index) of the reference positions p (ascending over the lanes and from call to call; `valid` lanes lie inside the
// read's reference span).  htslib pileup columns: M = X -> the base, D -> is_del, N -> is_refskip (pileup/mod.rs:783-851).
__device__ __forceinline__ void refwin_map(RefWin& w, const uint32_t* __restrict__ cg, uint32_t n_cigar, int32_t ref_start, bool valid, int32_t p,
    uint32_t* kind, uint32_t* q) {
  bool pending = valid; *kind = 2u; *q = 0u;
  for (;;) {
    if (!__any(pending)) break;
    const bool inw = pending && w.loaded && (uint32_t)(p - w.r_run) < w.Rtot;
    if (!__any(inw)) {
      if (w.loaded) { w.c0 += 256u; w.q_run += w.Qtot; w.r_run += (int32_t)w.Rtot; }
      if (w.c0 >= n_cigar) break;   // (cannot happen for positions inside the span)
      refwin_load(w, cg, n_cigar, ref_start);
      continue;
    }
    const uint32_t rel = inw ? (uint32_t)(p - w.r_run) : 0u;
    // the lane whose four ops hold `rel`: the number of lanes whose inclusive end is <= rel (six probes; the probe address is carried)
    uint32_t probe = 31u << 2;
#pragma unroll
    for (int hstep = 16; hstep >= 1; hstep >>= 1) {
      const uint32_t vv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)probe, (int)w.re);
      probe = vv <= rel ? probe + 4u * (uint32_t)hstep : probe - 4u * (uint32_t)hstep;
    }
    { const uint32_t vv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)probe, (int)w.re); probe = vv <= rel ? probe + 4u : probe; }
    const int oa = (int)(probe & 255u);
    const uint32_t o_m1 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.m1), o_m2 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.m2),
        o_m3 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.m3);
    const uint32_t o0 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.pk[0]), o1 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.pk[1]);
    const uint32_t o2 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.pk[2]), o3 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.pk[3]);
    const uint32_t pk = rel < o_m1 ? o0 : rel < o_m2 ? o1 : rel < o_m3 ? o2 : o3;
    if (inw) { *kind = pk & 3u; *q = (uint32_t)((p - ref_start) + ((int32_t)pk >> 2)); pending = false; }
  }
}

// The fused decoders' form of the same window (mkp_decode_slots*: VALU-issue bound, 42 % of their vector instructions were this mapping).
// The window's running values live in scalar registers (read back through readfirstlane after every update: left to itself the compiler
// carries them as vectors and advances them with v_cndmask), "nothing loaded yet" is a window of no reference bases in front of op 0 —
// no flag, no conditional advance — and the slot step maps its lanes in a pass of straight-line code, with the loop only behind it for
// the steps that straddle windows (the one loop of refwin_map kept every window register twice and copied 14 of them per iteration).
struct RefWinS { uint32_t c0, q_run, Rtot, Qtot; int32_t r_run; uint32_t re, m1, m2, m3, pk[4]; uint4 pref; };
__device__ __forceinline__ void refwin_s_init(RefWinS& w, const uint32_t* __restrict__ cg, uint32_t n_cigar, int32_t ref_start) {
  w.c0 = 0u - 256u; w.q_run = 0; w.r_run = ref_start; w.Rtot = 0; w.Qtot = 0; w.re = w.m1 = w.m2 = w.m3 = 0;
    w.pk[0] = w.pk[1] = w.pk[2] = w.pk[3] = 0;
  w.pref = cigar_quad(cg, n_cigar, 0);
}
// the next 256 ops; false behind the last op
__device__ __forceinline__ bool refwin_s_next(RefWinS& w, const uint32_t* __restrict__ cg, uint32_t n_cigar, int32_t ref_start) {
  w.c0 = rfl(w.c0 + 256u); w.q_run = rfl(w.q_run + w.Qtot); w.r_run = (int32_t)rfl((uint32_t)w.r_run + w.Rtot);
  if (w.c0 >= n_cigar) { w.Rtot = 0; w.Qtot = 0; return false; }
  const uint4 v = w.pref;
  w.pref = cigar_quad(cg, n_cigar, w.c0 + 256u);   // requested one window ahead
  const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
  uint32_t ql[4], rl[4], kind[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t f = (uint32_t)(MKP_CIGAR_LUT >> ((wd[j] & 15u) << 2)), len = wd[j] >> 4;
    ql[j] = len & (0u - (f & 1u)); rl[j] = len & (0u - ((f >> 1) & 1u)); kind[j] = (f >> 2) & 3u;
  }
  const uint32_t qsum = ql[0] + ql[1] + ql[2] + ql[3], rsum = rl[0] + rl[1] + rl[2] + rl[3];
  uint32_t qe;
  if (!__any((v.x | v.y | v.z | v.w) >= (128u << 4))) {   // ops shorter than 128: both running sums fit 16 bits, one scan serves both
    const uint32_t sc = wave_incl_scan(qsum | (rsum << 16));
    qe = sc & 0xffffu; w.re = sc >> 16;
  } else { qe = wave_incl_scan(qsum); w.re = wave_incl_scan(rsum); }
  w.Qtot = (uint32_t)__builtin_amdgcn_readlane((int)qe, 63); w.Rtot = (uint32_t)__builtin_amdgcn_readlane((int)w.re, 63);
  // d = query offset - (reference offset - ref_start) of the lane's op, carried op to op; rs = window-relative reference offset
  uint32_t rs = w.re - rsum;
  uint32_t d = (w.q_run - (uint32_t)(w.r_run - ref_start)) + (qe - qsum) - rs;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    w.pk[j] = (d << 2) | kind[j];
    d += ql[j] - rl[j]; rs += rl[j];
    if (j == 0) w.m1 = rs; else if (j == 1) w.m2 = rs; else if (j == 2) w.m3 = rs;
  }
  return true;
}
// one pass over the lanes whose position lies in the loaded window
__device__ __forceinline__ void refwin_s_pass(const RefWinS& w, int32_t ref_start, bool& pending, int32_t p, uint32_t& kind, uint32_t& q) {
  const uint32_t rel0 = (uint32_t)(p - w.r_run);
  const bool inw = pending && rel0 < w.Rtot;
  if (!__any(inw)) return;
  const uint32_t rel = inw ? rel0 : 0u;
  uint32_t probe = 31u << 2;
#pragma unroll
  for (int hstep = 16; hstep >= 1; hstep >>= 1) {
    const uint32_t vv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)probe, (int)w.re);
    probe = vv <= rel ? probe + 4u * (uint32_t)hstep : probe - 4u * (uint32_t)hstep;
  }
  { const uint32_t vv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)probe, (int)w.re); probe = vv <= rel ? probe + 4u : probe; }
  const int oa = (int)(probe & 255u);
  const uint32_t o_m1 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.m1), o_m2 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.m2),
      o_m3 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.m3);
  const uint32_t o0 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.pk[0]), o1 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.pk[1]);
  const uint32_t o2 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.pk[2]), o3 = (uint32_t)__builtin_amdgcn_ds_bpermute(oa, (int)w.pk[3]);
  const uint32_t pk = rel < o_m1 ? o0 : rel < o_m2 ? o1 : rel < o_m3 ? o2 : o3;
  if (inw) { kind = pk & 3u; q = (uint32_t)((p - ref_start) + ((int32_t)pk >> 2)); pending = false; }
}
__device__ __forceinline__ void refwin_s_map(RefWinS& w, const uint32_t* __restrict__ cg, uint32_t n_cigar, int32_t ref_start, bool valid, int32_t p,
    uint32_t* kind_out, uint32_t* q_out) {
  bool pending = valid; uint32_t kind = 2u, q = 0u;
  refwin_s_pass(w, ref_start, pending, p, kind, q);
  while (__any(pending)) {
    if (!refwin_s_next(w, cg, n_cigar, ref_start)) break;   // (cannot happen for positions inside the span)
    refwin_s_pass(w, ref_start, pending, p, kind, q);
  }
  *kind_out = kind; *q_out = q;
}

// coverage feature of a slot: NoCall(base) on the alignment strand (the base complemented on '-': get_forward_read_base,
// pileup/mod.rs:612-624), Delete, nothing on a ref-skip, and no feature for a base that is not A/C/G/T (864-874)
__device__ __forceinline__ uint32_t cover_feature(uint32_t kind, uint32_t nib, uint32_t aln) {
  if (kind == 1u) return feat(MKP_C_DEL, aln);
  if (kind != 0u) return MKP_FB_NONE;
  const int sb = nib2base(nib);
  if (sb < 0) return MKP_FB_BLANK;
  return feat((uint32_t)MKP_C_NC + (aln ? 3u - (uint32_t)sb : (uint32_t)sb), aln);
}

// ----------------------------------------------------------------------------------------------------------------------
// mkp_decode_slots: MM/ML decode + threshold caller + coverage, slot-driven (see the head of the file).
// Reference: DeltaListConverter (mod_bam.rs:667-733), get_base_mod_probs (1213-1295), combine_checked (629-656),
// into_collapsed (530-627), MultipleThresholdModCaller::call (threshold_mod_caller.rs:28-63), ReadCache::add_record
// (read_cache.rs:111-211), get_aligned_pairs_forward (util.rs:122-145), process_region's alignment loop (pileup/mod.rs:783-939).
// One wave runs one read start to end, so what bounds the kernel is the chain of dependent memory round trips per read, not
// instruction issue: everything that depends only on the header (first slots, first CIGAR window, the first 8192 bases) is
// requested before the layout tables are looked at and the sweep issues four 16-byte loads per lane at a time.  Measured (SQ
// counters, C3): the kernel is VALU-issue bound (~80 % of the SIMD cycles), so the per-base and per-slot instruction counts matter:
// three VALU per SEQ dword for the flags, one popcount per 32 bases, the caller resolved to a fixed walk per read.
// MULTI = false: reads of at most SL_WB bases — one base window, swept before the slot loop, and a straight-line slot step;
// MULTI = true: longer reads — the windows are swept as the slots reach them.  (Two instances: the window-advance code inside the
// slot loop costs the short-read kernel registers it never uses, and uniform values that do not fit the SGPR file are spilled to
// VGPR lanes — VALU instructions in a VALU-bound kernel.)
#define FUSED_PARAMS(PRM) const MkpWork* __restrict__ work, uint32_t n_reads, const uint32_t* __restrict__ cigar, const uint8_t* __restrict__ seqs, \
    const uint32_t* __restrict__ ranks, const uint8_t* __restrict__ ml, const MkpFusedDesc* __restrict__ fdesc, PRM prm, const uint32_t* __restrict__ slot_pos, \
    uint8_t* __restrict__ cov, MkpVisit* __restrict__ visits, MkpReadOut* __restrict__ readout
#define FUSED_PASS work, n_reads, cigar, seqs, ranks, ml, fdesc, prm, slot_pos, cov, visits, readout
template <bool MULTI>
__device__ __forceinline__ void decode_slots_body(FUSED_PARAMS(const MkpRunParams&), SlotLds* __restrict__ lds_all) {
  const int lane = lane_id();
  const uint32_t wib = rfl(threadIdx.x >> 6);
  const uint32_t widx = rfl(blockIdx.x * (blockDim.x >> 6)) + wib;
  if (widx >= n_reads) return;
  const MkpWork h = work[widx];
  SlotLds& W = lds_all[wib];
  const bool rev = (h.flags & MKP_RF_REVERSE) != 0;
  const uint32_t aln = rev ? 1u : 0u;
  const uint32_t L = h.l_seq, nd = (L + 7u) >> 3;
  const uint8_t* __restrict__ seqb = seqs + h.seq_off;   // reads start 4-byte aligned, zero-padded to a dword
  const uint32_t* __restrict__ cg = cigar + h.cigar_off;
  bool have_calls = !(h.flags & MKP_RF_BAD) && h.n_tags != 0;
  // combine_checked's sum test (mod_bam.rs:629-656) — two tags on one base: the probabilities of a call add up to more than 1.01 —
  // is made by the host planner over the ML bytes (the f32 sums are exact multiples of 1/512: an integer comparison)
  const bool err_sum = (h.flags & MKP_RF_SUMERR) != 0;
  const uint32_t n_sl = h.n_sl;
  const uint32_t* __restrict__ spos = slot_pos + h.gs0;

  // ---- requests that depend on the header alone (per-read bases are uniform: scalar base + 32-bit lane offset)
  // 16 bytes of SEQ at dword d.  The address is clamped to the read's last dword (the SEQ buffer has slack behind the last read)
  // and dwords past the read come back as zero: no divergent tail path.
  auto load4 = [&](uint32_t d) {
    uint4 x = ldo<uint4>(seqb, 4u * min(d, nd - 1u));
    if (__any(d + 4u > nd)) { if (d >= nd) x.x = 0u; if (d + 1u >= nd) x.y = 0u; if (d + 2u >= nd) x.z = 0u; if (d + 3u >= nd) x.w = 0u; }
    return x;
  };
  uint32_t p_next = (uint32_t)lane < n_sl ? ldo<uint32_t>(spos, 4u * (uint32_t)lane) : 0u;
  RefWinS rw; refwin_s_init(rw, cg, h.n_cigar, h.ref_start);
  uint4 xpre[4];   // stored bases [0, 8192): a 16-byte vector per lane and 2048 bases
#pragma unroll
  for (int j = 0; j < 4; j++) xpre[j] = (have_calls && 2048u * (uint32_t)j < L) ? load4(4u * (64u * (uint32_t)j + (uint32_t)lane)) : make_uint4(0u,
      0u, 0u, 0u);
  const uint32_t pad_nib = (L & 1u) ? ((uint32_t)seqb[L >> 1] & 15u) : 0u;   // the low nibble of the last byte is not a base when L is odd

  // ---- the read's one (mod strand, base) group.  The caller's walk over a call's map in iteration order is resolved once per
  // layout by the host (MkpFusedDesc: one scalar load): where the ML byte of the i-th code sits (tag + index: offset and stride
  // per call), its pass threshold and the counter of Modified(code).  --ignore / --preset traditional (ReDistribute) keep the
  // general tables.
  const bool collapse = prm.numeric_mode == 2;
  uint32_t fmisc = 0, f_col = 0;
  int32_t i_can = 0;   // Canonical's threshold in units of 2^-11 (the exact integer form of the caller, below); the codes' are in W.ck_thr
  // where the ML byte of the i-th code of call j sits: ml[W.ck_off[i] + j * W.ck_str[i]] (tag + index inside the tag: offset and stride)
  uint32_t mlx_o = 0, mlx_s = 0;
  uint32_t t_n = 0;
  uint32_t e_pre = 0, e_last = 0;   // the first 64 entries of the rank list as the first window consumes it, and its last entry
  const uint32_t* __restrict__ rk = ranks;
  if (have_calls) {
    t_n = h.n_calls;
    rk = ranks + h.rank_off;
    if (t_n) {
      const uint32_t i = rev ? t_n - 64u + (uint32_t)lane : (uint32_t)lane;
      e_pre = ((int32_t)i >= 0 && i < t_n) ? ldo<uint32_t>(rk, 4u * i) : (rev ? 0u : 0xffffffffu);
      e_last = rk[t_n - 1u];
    }
    const MkpFusedDesc fd = fdesc[h.layout];
    fmisc = fd.misc; i_can = fd.i_can;
    if (lane < MKP_KMAX) {
      const uint32_t src = (fd.it_src >> (4 * lane)) & 15u, tg = src & 1u;
      W.ck_off[lane] = (tg ? h.ml_off1 : h.ml_off0) + (src >> 1); W.ck_str[lane] = (fd.nc >> (8u * tg)) & 0xffu;
      W.ck_thr[lane] = lane == 0 ? fd.i_thr[0] : lane == 1 ? fd.i_thr[1] : lane == 2 ? fd.i_thr[2] : fd.i_thr[3];
      W.ck_cid[lane] = (fd.it_cid >> (8 * lane)) & 0xffu;
    }
    if (collapse) {
      f_col = fd.col;
      const uint32_t src = (fd.col >> 1) & 15u, tg = src & 1u;
      mlx_o = (tg ? h.ml_off1 : h.ml_off0) + (src >> 1); mlx_s = (fd.nc >> (8u * tg)) & 0xffu;
    }
    if (t_n == 0) have_calls = false;   // a tag without calls: the record has no modified-base information
  }
  const uint32_t sg0u = (fmisc >> 2) & 1u, n_post = (fmisc >> 3) & 7u;
  const bool int_caller = (fmisc >> (collapse ? 7 : 6)) & 1u;     // the exact integer form of the caller applies (fused_desc)
  const uint32_t red_shift = (f_col >> 5) & 3u;                   // log2 of the number of codes a collapsed code's probability is shared among
  const uint32_t xs = rev ? 3u - (fmisc & 3u) : (fmisc & 3u);     // the stored base the tags count
  bool err = have_calls && err_sum;
  const uint32_t pat = 0x11111111u << xs;
  const bool pad_hit = (L & 1u) && pad_nib == (1u << xs);

  // ---- base windows: stored bases [w0, w0 + SL_WB) swept into F / P.
  // F word = 32 bases = 4 SEQ dwords: the flag of nibble n of dword k sits at bit 4n + 3 - k (base 2j of a dword is nibble 2j+1).
  uint32_t w0 = 0, cntW = 0, cum = 0, tot = 0, t_cur = 0, t_base = 0;
  bool bw_loaded = false, first_mark = true;
  auto flags4 = [&](const uint4& x) {   // "nibble != base" lands on bit 3 of the nibble after two shift-ors; the four dwords interleave
    uint32_t n0 = x.x ^ pat, n1 = x.y ^ pat, n2 = x.z ^ pat, n3 = x.w ^ pat;
    n0 |= n0 << 1; n1 |= n1 << 1; n2 |= n2 << 1; n3 |= n3 << 1;
    n0 |= n0 << 2; n1 |= n1 << 2; n2 |= n2 << 2; n3 |= n3 << 2;
    const uint32_t a = (n0 & 0x88888888u) | ((n1 >> 1) & ~0x88888888u), b = ((n2 >> 2) & 0x22222222u) | ((n3 >> 3) & ~0x22222222u);
    return ~((a & 0xccccccccu) | (b & ~0xccccccccu));
  };
  auto sweep = [&]() {   // F, P, cntW of the window at w0; four vectors per lane (8192 bases) in flight at a time
    const uint32_t nwords = min(SL_FW, (L - w0 + 31u) >> 5);
    const uint32_t dbase = w0 >> 3;
    uint32_t carry = 0;
    if (MULTI) wave_lds_fence();   // the previous window's readers are done (same wave)
    for (uint32_t i0 = 0; i0 < nwords; i0 += 256) {
      uint4* x = xpre;   // (the vectors requested at the top serve the first 8192 bases)
      if (!(w0 == 0 && i0 == 0)) {
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = (i0 + 64u * (uint32_t)j < nwords) ? load4(dbase + 4u * (i0 + 64u * (uint32_t)j + (uint32_t)lane))
            : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      // two vectors per prefix sum: a word holds at most 32 occurrences, 64 words 2 048 — both running counts fit 16 bits
      for (int jp = 0; jp < 4; jp += 2) {
        if (i0 + 64u * (uint32_t)jp < nwords) {
          const bool has1 = i0 + 64u * (uint32_t)(jp + 1) < nwords;   // (uniform)
          const uint32_t wi0 = i0 + 64u * (uint32_t)jp + (uint32_t)lane, wi1 = wi0 + 64u;
          // (a window of 416 words ends inside a vector: the words behind it belong to the next window)
          const uint32_t Fw0 = flags4(x[jp]), c0 = wi0 < nwords ? (uint32_t)__popc(Fw0) : 0u;
          uint32_t Fw1 = 0, c1 = 0;
          if (has1) { Fw1 = flags4(x[jp + 1]); c1 = wi1 < nwords ? (uint32_t)__popc(Fw1) : 0u; }
          const uint32_t sc = wave_incl_scan(c0 | (c1 << 16));
          const uint32_t tot01 = (uint32_t)__builtin_amdgcn_readlane((int)sc, 63), tot0 = tot01 & 0xffffu;
          if (wi0 < nwords) { W.F[wi0] = Fw0; W.P[wi0] = (uint16_t)(carry + (sc & 0xffffu) - c0); }
          if (has1 && wi1 < nwords) { W.F[wi1] = Fw1; W.P[wi1] = (uint16_t)(carry + tot0 + (sc >> 16) - c1); }
          carry += tot0 + (tot01 >> 16);
        }
      }
    }
    wave_lds_fence();
    if (pad_hit && L - w0 < (nwords << 5)) {   // the pad nibble matched: take its flag back (it lies behind every base)
      const uint32_t qr = L - w0;
      if (lane == 0) W.F[qr >> 5] &= ~(1u << (4u * ((qr & 7u) ^ 1u) + 3u - ((qr >> 3) & 3u)));
      carry -= 1u;
      wave_lds_fence();
    }
    cntW = carry; bw_loaded = true;
  };
  auto mark = [&]() {   // B, WP of the window: the listed ranks among its occurrences (stored-order ordinals cum .. cum + cntW)
    const uint32_t nbw = (cntW + 31u) >> 5;
    for (uint32_t k = (uint32_t)lane; k < nbw + 1u; k += 64) W.B[k] = 0;
    wave_lds_fence();
    t_base = t_cur;
    if (!rev) {
      const uint32_t whi = cum + cntW;
      for (;;) {
        const uint32_t i = t_cur + (uint32_t)lane; const bool valid = i < t_n;
        const uint32_t e = first_mark ? e_pre : (valid ? ldo<uint32_t>(rk, 4u * i) : 0xffffffffu);
        first_mark = false;
        const bool hit = valid && e < whi;
        const uint32_t o = e - cum;
        if (hit) atomicOr(&W.B[o >> 5], 1u << (o & 31u));
        const uint32_t nh = (uint32_t)__popcll(__ballot(hit));
        t_cur += nh;
        if (nh < 64u) break;
      }
    } else {   // forward rank of a stored ordinal o: tot - 1 - o; the list is consumed from its end
      const uint32_t wlo = tot - cum - cntW;
      for (;;) {
        const uint32_t i = t_cur - 64u + (uint32_t)lane; const bool valid = (int32_t)i >= 0 && i < t_cur;
        const uint32_t e = first_mark ? e_pre : (valid ? ldo<uint32_t>(rk, 4u * i) : 0u);
        first_mark = false;
        const bool hit = valid && e >= wlo;
        const uint32_t o = (tot - 1u - e) - cum;
        if (hit) atomicOr(&W.B[o >> 5], 1u << (o & 31u));
        const uint32_t nh = (uint32_t)__popcll(__ballot(hit));
        t_cur -= nh;
        if (nh < 64u) break;
      }
    }
    wave_lds_fence();
    uint32_t carry = 0;
    for (uint32_t k0 = 0; k0 < nbw; k0 += 64) {
      const uint32_t k = k0 + (uint32_t)lane;
      const uint32_t c = k < nbw ? (uint32_t)__popc(W.B[k]) : 0u;
      const uint32_t inc = wave_incl_scan(c);
      if (k < nbw) W.WP[k] = (uint16_t)(carry + inc - c);
      carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    wave_lds_fence();
  };
#ifdef MKP_DEBUG
  if (prm.debug_skip & 256u) have_calls = false;   // ablation: no sweep, no calls
#endif
  if (have_calls && !err) {
    if (MULTI) {   // several windows: the total is needed up front (reverse reads; the list's last entry)
      uint32_t acc = 0;
      for (uint32_t d0 = 0; d0 < nd; d0 += 1024) {
        uint4 x[4];
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = d0 == 0 ? xpre[j] : load4(d0 + 256u * (uint32_t)j + 4u * (uint32_t)lane);
#pragma unroll
        for (int j = 0; j < 4; j++) acc += (uint32_t)__popc(flags4(x[j]));
      }
      tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(acc), 63) - (pad_hit ? 1u : 0u);
    } else { sweep(); tot = cntW; }
    // a delta list must not run past the last occurrence of its base (mod_bam.rs:705-727)
    if (rfl(e_last) >= tot) err = true;
    else { t_cur = rev ? t_n : 0u; if (!MULTI) mark(); }
  }
  if (err) have_calls = false;   // the record only contributes coverage (skip_set, read_cache.rs:272-277)

  // ---- the read's slots, 64 per step
  bool gaps = false; uint32_t n_callfeat = 0;
#ifdef MKP_DEBUG
  const uint32_t n_sl_walk = (prm.debug_skip & 512u) ? 0u : n_sl;   // ablation: no slot loop
#else
  const uint32_t n_sl_walk = n_sl;
#endif
  for (uint32_t s0 = 0; s0 < n_sl_walk; s0 += 64u) {
    const uint32_t i = s0 + (uint32_t)lane;
    const bool valid = i < n_sl;
    const int32_t p = (int32_t)p_next;
    { const uint32_t in = i + 64u; p_next = in < n_sl ? ldo<uint32_t>(spos, 4u * in) : 0u; }
    uint32_t kind, q;
#ifdef MKP_DEBUG
    if (prm.debug_skip & 64u) { kind = 0u; q = min((uint32_t)(p - h.ref_start), L - 1u); } else   // ablation: no CIGAR mapping
#endif
    refwin_s_map(rw, cg, h.n_cigar, h.ref_start, valid, p, &kind, &q);
    const bool is_match = valid && kind == 0u && q < L;
    const uint32_t byte = is_match ? (uint32_t)ldo<uint8_t>(seqb, q >> 1) : 0u;
    uint32_t call_fb = 0xffffffffu;
#ifdef MKP_DEBUG
    if (have_calls && !(prm.debug_skip & 128u)) {   // ablation: no rank lookups, no calls
#else
    if (have_calls) {
#endif
      bool pend = is_match;
      for (;;) {
        bool inw = pend;
        if (MULTI) {
          if (!__any(pend)) break;
          inw = pend && bw_loaded && (q - w0) < SL_WB;
          if (!__any(inw)) {   // the next base window (never skipped: the occurrence counts run on)
            if (bw_loaded) { w0 += SL_WB; cum += cntW; }
            if (w0 >= L) break;
            sweep(); mark();
            continue;
          }
        }
        const uint32_t qr = inw ? q - w0 : 0u, wv = qr >> 5;
        const uint32_t Fw = W.F[wv], Pw = W.P[wv];
        const uint32_t r = qr & 7u, kd = (qr >> 3) & 3u, sh = 3u - kd;
        const bool cand = inw && ((Fw >> (4u * (r ^ 1u) + sh)) & 1u);
        // occurrences before q inside the word: the dwords below (bit offsets above 3 - kd in every nibble), then the nibbles of
        // the bases before q in its own dword (base 2j sits in nibble 2j+1)
        const uint32_t m_dw = ~((0x11111111u << (4u - kd)) - 0x11111111u);
        const uint32_t re4 = (r & 6u) << 2;
        const uint32_t m_in = (((1u << re4) - 1u) & 0x11111111u) | ((r & 1u) << (4u * r));
        const uint32_t ord = Pw + (uint32_t)__popc(Fw & (m_dw | (m_in << sh)));               // ordinal inside the window, stored order
        const uint32_t Bw = cand ? W.B[ord >> 5] : 0u;
        const bool listed = cand && ((Bw >> (ord & 31u)) & 1u);
        if (__any(listed)) {
          const uint32_t jrel = (uint32_t)W.WP[listed ? (ord >> 5) : 0u] + (uint32_t)__popc(Bw & ((1u << (ord & 31u)) - 1u));
          const uint32_t jx = listed ? (rev ? (t_base - 1u - jrel) : (t_base + jrel)) : 0u;
          // MultipleThresholdModCaller::call (threshold_mod_caller.rs:28-63) on the map of this call, entries in the map's
          // iteration order: pass threshold, Iterator::max keeps the last maximum, canonical pushed last.  ReDistribute runs
          // (--ignore, --preset traditional) first add the collapsed code's share to every entry (mod_bam.rs:558-600).
          // All ML bytes of the call are requested before any is looked at (round 5 waited for each in turn: two to five dependent
          // memory round trips per 64 slots, the longest chain of the wave's life).
          const uint32_t jl = listed ? jx : 0u;
          const uint4 off4 = *reinterpret_cast<const uint4*>(W.ck_off), str4 = *reinterpret_cast<const uint4*>(W.ck_str);
          const uint32_t ml_o[MKP_KMAX] = {off4.x, off4.y, off4.z, off4.w}, ml_s[MKP_KMAX] = {str4.x, str4.y, str4.z, str4.w};
          uint32_t mlb[MKP_KMAX] = {0u, 0u, 0u, 0u}, mlx = 0;
          auto mlq = [&](int k) { return (uint32_t)ldo<uint8_t>(ml, __umul24(jl, ml_s[k]) + ml_o[k]); };
          switch (n_post) {   // (uniform)
            case 1: mlb[0] = mlq(0); break;
            case 2: mlb[0] = mlq(0); mlb[1] = mlq(1); break;
            case 3: mlb[0] = mlq(0); mlb[1] = mlq(1); mlb[2] = mlq(2); break;
            case 4: mlb[0] = mlq(0); mlb[1] = mlq(1); mlb[2] = mlq(2); mlb[3] = mlq(3); break;
            default: break;
          }
          if (f_col & 1u) mlx = (uint32_t)ldo<uint8_t>(ml, __umul24(jl, mlx_s) + mlx_o);
          uint32_t cid = MKP_C_FAIL;
          if (int_caller) {
            // The same walk in integers, exactly: q -> (q + 0.5) / 256 is a multiple of 2^-9, the share of a collapsed code (its
            // probability over 1, 2 or 4 codes) a multiple of 2^-11, every sum of up to five of them and 1 - sum are exact in f32 — so
            // the f32 comparisons of the reference are comparisons of integers in units of 2^-11, thresholds rounded up to the next
            // multiple by the host (fused_desc: the least integer T with T / 2048 >= threshold).  One straight-line instance per number
            // of codes (uniform switch): no per-code masks held in scalar registers.
            const int4 thr4 = *reinterpret_cast<const int4*>(W.ck_thr);
            const uint4 cid4 = *reinterpret_cast<const uint4*>(W.ck_cid);
            const int32_t red4 = (f_col & 1u) ? (int32_t)(((2u * mlx + 1u) << 2) >> red_shift) + 4 : 4;
            int32_t sum = 0, best = INT32_MIN;
            auto step = [&](uint32_t b, int32_t thr, uint32_t c) {
              const int32_t v = (int32_t)(b << 3) + red4;
              sum += v;
              const bool take = v >= max(thr, best);   // passes, and no entry before it is larger (the last maximum wins)
              cid = take ? c : cid; best = take ? v : best;
            };
            switch (n_post) {
              case 1: step(mlb[0], thr4.x, cid4.x); break;
              case 2: step(mlb[0], thr4.x, cid4.x); step(mlb[1], thr4.y, cid4.y); break;
              case 3: step(mlb[0], thr4.x, cid4.x); step(mlb[1], thr4.y, cid4.y); step(mlb[2], thr4.z, cid4.z); break;
              case 4: step(mlb[0], thr4.x, cid4.x); step(mlb[1], thr4.y, cid4.y); step(mlb[2], thr4.z, cid4.z); step(mlb[3], thr4.w, cid4.w); break;
              default: break;
            }
            const int32_t pc = 2048 - sum;
            if (pc >= max(i_can, best)) cid = (fmisc >> 8) & 0xffu;
          // (a share over three codes: the f32 walk; its thresholds are read here — this is the rare path — not held in registers by every read)
          } else {
          const MkpFusedDesc& fdr = fdesc[h.layout];
          float f_thr[MKP_KMAX]; const float thr_can = fdr.thr_can;
#pragma unroll
          for (int k = 0; k < MKP_KMAX; k++) f_thr[k] = fdr.it_thr[k];
          float red = 0.f;
          if (f_col & 1u) red = (((float)mlx + 0.5f) / 256.0f) / fdr.n_other;
          float s = 0.f, best_p = 0.f; bool have = false;
#pragma unroll
          for (int k = 0; k < MKP_KMAX; k++) {
            if ((uint32_t)k < n_post) {
              float pr = ((float)mlb[k] + 0.5f) / 256.0f;   // quals_to_probs (mod_bam.rs:808-816)
              if (f_col & 1u) pr = pr + red;
              s = s + pr;
              const bool take = pr >= f_thr[k] && (!have || !(pr < best_p));
              cid = take ? W.ck_cid[k] : cid; best_p = take ? pr : best_p; have = have || take;
            }
          }
          const float pc = 1.0f - s;
          if (pc >= thr_can && (!have || !(pc < best_p))) cid = (fmisc >> 8) & 0xffu;
          }
          if (listed) call_fb = feat(cid, aln ^ sg0u);   // FeatureVector::add_feature's tally (pileup/mod.rs:238-281)
        }
        if (!MULTI) break;
        pend = pend && !inw;
      }
    }
    const uint32_t nib = (q & 1u) ? (byte & 15u) : (byte >> 4);
    uint32_t fb = cover_feature(valid ? kind : 2u, nib, aln);
    if (valid && kind == 0u && q >= L) fb = MKP_FB_NONE;   // (a CIGAR longer than SEQ is refused by the packer)
    // (a scalar count: no per-lane counter, no scan at the end)
    { const bool cf = call_fb != 0xffffffffu; if (cf) fb = call_fb; n_callfeat += (uint32_t)__popcll(__ballot(cf)); }
    if (valid) cov[h.cov_off + i] = (uint8_t)fb;
    gaps = gaps || (valid && fb == MKP_FB_NONE);
  }
  gaps = __any(gaps);
  const bool ok = have_calls;
  const uint32_t tally = aln ^ sg0u, ob_const = fmisc >> 16;
  const uint32_t n_cf = n_callfeat;
  if (lane == 0) {
    // (what the records below need of the work record is read again here rather than held in scalar registers through the slot loop)
    const MkpWork* hp = work + widx; asm volatile("" : "+s"(hp));
    const uint32_t h_gs0 = hp->gs0, h_flags = hp->flags, h_cov_off = hp->cov_off, h_rid = hp->rid;
    MkpVisit v; v.gs0 = h_gs0; v.n_sl = n_sl; v.cov_off = h_cov_off;
    v.flags = (ok ? MKP_VF_OK : 0u) | (rev ? MKP_VF_REV : 0u) | (gaps ? MKP_VF_GAPS : 0u) | ((h_flags >> MKP_RF_KEY_SHIFT) << 8);
    v.obs0 = (ok && tally == 0u) ? ob_const : 0u; v.obs1 = (ok && tally == 1u) ? ob_const : 0u;
    v.over_off = 0; v.n_over = 0;
    visits[h_rid] = v;
    MkpReadOut out; out.n_events = ok ? n_cf : 0u; out.ok = ok ? 1u : 0u; out.obs[0] = v.obs0; out.obs[1] = v.obs1;
    readout[h_rid] = out;
  }
}

// Residency on gfx950 is also bounded by the SIMD's 800 scalar registers: a wave is charged ceil(sgprs / 16) * 16 + 16 of them
// (MI355X_MICROARCH.md, "Residency and cooperative launch"), so the 105 the compiler takes when left alone admit six waves per SIMD whatever
// the LDS and VGPR budgets say (the compiler's own occupancy figure says eight).  The short-read kernel is capped at 80 — eight waves; the
// spilled scalars cost ~26 v_readlane / v_writelane per slot batch (+7 % VALU) against +33 % resident waves: 0.75 -> 0.70 ms on C3
// (A/B on one box: tools/dbg/ab.sh, MKP_DECODE_SGPRS=96 gives seven waves and 0.705).  The long-read kernel (several slot windows per read,
// 106 SGPRs / 79 VGPRs left alone: six waves) is bounded to seven waves — 94 SGPRs, 72 VGPRs, 12 B of scratch: C3 decode 0.694 -> 0.682 ms on
// one box; eight waves (MKP_LONG_WAVES=8) spills enough to lose: 0.725.
#ifndef MKP_DECODE_SGPRS
#define MKP_DECODE_SGPRS 80
#endif
#ifndef MKP_LONG_WAVES
#define MKP_LONG_WAVES 7
#endif
#define MKP_LONG_LB __launch_bounds__(256, MKP_LONG_WAVES)
#define MKP_SLOT_KERNEL(NAME, MULTI, ...) extern "C" __global__ void __VA_ARGS__ NAME(FUSED_PARAMS(MkpRunParams)) { \
    __shared__ __attribute__((aligned(16))) SlotLds lds_all[4]; decode_slots_body<MULTI>(FUSED_PASS, lds_all); }
MKP_SLOT_KERNEL(mkp_decode_slots, false, __attribute__((amdgpu_num_sgpr(MKP_DECODE_SGPRS))) __launch_bounds__(256))
MKP_SLOT_KERNEL(mkp_decode_slots_long, true, MKP_LONG_LB)

// ----------------------------------------------------------------------------------------------------------------------
// mkp_cover_reads: coverage features of the reads the event-producing decode kernels handled, with their call events merged in.
// An event's counter id / tally strand are those the decode kernels computed; the first event on a position (bit 12) replaces the
// NoCall of the base, a second one (pos_call and neg_call on one base, pileup/mod.rs:889-938) goes to the read's overflow list —
// written over the front of its own event slice, which holds only events already consumed.
extern "C" __global__ void __launch_bounds__(256) mkp_cover_reads(SLOT_PARAMS(MkpRunParams)) {
  __shared__ uint32_t stage_all[4][64];
  const int lane = lane_id();
  const uint32_t wib = rfl(threadIdx.x >> 6);
  const uint32_t widx = rfl(blockIdx.x * (blockDim.x >> 6)) + wib;
  if (widx >= n_reads) return;
  const uint32_t rid = rfl(read_ids[widx]);
  const MkpReadHdr h = hdrs[rid];
  const MkpReadOut ro = readout[rid];
  uint32_t* __restrict__ stage = stage_all[wib];
  const bool rev = (h.flags & MKP_RF_REVERSE) != 0;
  const uint32_t aln = rev ? 1u : 0u, L = h.l_seq;
  const uint8_t* __restrict__ seqb = seqs + h.seq_off;
  const uint32_t* __restrict__ cg = cigar + h.cigar_off;
  const bool ok = ro.ok == 1u;
  const uint32_t n_ev = ok ? ro.n_events : 0u;
  MkpEvent* __restrict__ ev = events + h.event_off;
  const uint32_t n_sl = h.n_sl, gs0 = h.gs0;
  uint8_t* __restrict__ covp = cov + h.cov_off;
  RefWin rw; rw.c0 = 0; rw.q_run = 0; rw.r_run = h.ref_start; rw.Rtot = 0; rw.Qtot = 0; rw.re = rw.m1 = rw.m2 = rw.m3 = 0;
    rw.pk[0] = rw.pk[1] = rw.pk[2] = rw.pk[3] = 0; rw.loaded = false;
  rw.pref = cigar_quad(cg, h.n_cigar, 0);
  uint32_t p_next = (uint32_t)lane < n_sl ? slot_pos[gs0 + (uint32_t)lane] : 0u;
  uint32_t ec = 0, n_over = 0; bool gaps = false;
  stage[lane] = 0xffffffffu;
  for (uint32_t s0 = 0; s0 < n_sl; s0 += 64) {
    const uint32_t i = s0 + (uint32_t)lane; const bool valid = i < n_sl;
    const int32_t p = (int32_t)p_next;
    { const uint32_t in = i + 64u; p_next = in < n_sl ? slot_pos[gs0 + in] : 0u; }
    uint32_t kind, q;
    refwin_map(rw, cg, h.n_cigar, h.ref_start, valid, p, &kind, &q);
    const bool is_match = valid && kind == 0u && q < L;
    const uint32_t byte = is_match ? (uint32_t)seqb[q >> 1] : 0u;
    const uint32_t nib = (q & 1u) ? (byte & 15u) : (byte >> 4);
    uint32_t fb = cover_feature(valid ? kind : 2u, nib, aln);
    if (valid && kind == 0u && q >= L) fb = MKP_FB_NONE;
    if (ec < n_ev) {
      // the read's events up to the step's last slot position, 64 at a time; each finds the lane holding its position
      const uint32_t nval = min(64u, n_sl - s0);
      const uint32_t pkey = valid ? (uint32_t)p : 0xffffffffu;
      const uint32_t p_hi = (uint32_t)__builtin_amdgcn_readlane((int)pkey, (int)(nval - 1u));
      wave_lds_fence();
      for (;;) {
        const uint32_t k = ec + (uint32_t)lane; const bool in = k < n_ev;
        MkpEvent e; e.pos = 0xffffffffu; e.info = 0;
        if (in) e = ev[k];
        const bool take = in && e.pos <= p_hi;
        const int tgt = find_sorted(pkey, take ? e.pos : 0u);
        const uint32_t tp = (uint32_t)__shfl((int)pkey, tgt & 63, 64);
        const bool match = take && tgt < 64 && tp == e.pos;
        const uint32_t eb = feat(e.info & 0x1fu, (e.info >> 8) & 1u);
        const bool prim = match && (e.info & (1u << 12)), over = match && !(e.info & (1u << 12));
        if (prim) stage[tgt] = eb;
        const unsigned long long ob = __ballot(over);
        if (over) { MkpEvent o; o.pos = gs0 + s0 + (uint32_t)tgt; o.info = eb; ev[n_over + (uint32_t)__popcll(ob & lanemask_lt())] = o; }
        n_over += (uint32_t)__popcll(ob);
        const uint32_t nt = (uint32_t)__popcll(__ballot(take));
        ec += nt;
        if (nt < 64u) break;
      }
      wave_lds_fence();
      const uint32_t sv = stage[lane];
      if (sv != 0xffffffffu) { fb = sv; stage[lane] = 0xffffffffu; }
    }
    if (valid) covp[i] = (uint8_t)fb;
    gaps = gaps || (valid && fb == MKP_FB_NONE);
  }
  gaps = __any(gaps);
  if (lane == 0) {
    MkpVisit v; v.gs0 = gs0; v.n_sl = n_sl; v.cov_off = h.cov_off;
    v.flags = (ok ? MKP_VF_OK : 0u) | (rev ? MKP_VF_REV : 0u) | (gaps ? MKP_VF_GAPS : 0u) | ((h.flags >> MKP_RF_KEY_SHIFT) << 8);
    v.obs0 = ok ? ro.obs[0] : 0u; v.obs1 = ok ? ro.obs[1] : 0u;
    v.over_off = h.event_off; v.n_over = n_over;
    visits[rid] = v;
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// mkp_pileup_stream: accumulate + emit over the feature stream.  LDS: [counter | observed-code slot][S] packed tallies ('+' tally
// in the low, '-' in the high 16 bits; the host refuses shards with more than 65535 records over one position), the tile's slot
// positions, a word per slot for the emission (focus byte | strand-combining partners) and the row map.  Waves draw the tile's candidate
// reads from an LDS ticket; a visit = the read's MkpVisit (scalar loads), its bytes for the tile's slots (a dword per lane), one LDS
// atomic per feature.  Observed codes: +1 / -1 at the ends of the read's slot range (and at every change between covered and not
// covered when the read holds ref-skips).
//
// Row emission (round 6; rounds 2-5 ran the parameter-driven interpreter of mkp_dev_rows.hpp twice per slot — count, then write — with one
// thread per slot: ~800 VALU instructions per thread, 29 spilled registers, rows of a slot written with stride-2 stores):
//   E1  one thread per SLOT decides which rows exist: a bit per (strand | motif, code group) candidate — existence needs the coverage of
//       the candidate's primary base and its observed-code count, a handful of LDS reads — and counts them;
//   E2  block scan of the counts; wave 0 reserves the tile's run in the row buffer (decoupled look-back over the runs before, in ticket
//       order) while every thread scatters (slot, candidate) words of its rows into the ROW MAP in LDS;
//   E3  one thread per ROW fills its row from the tallies and stores it: every store instruction writes 64 consecutive rows of one column.
// MkpRunParams is resolved once per workgroup into a small table (StreamProg) so that neither pass walks slot lists.
template <bool KEYED, uint32_t VB /* visits drawn per ticket, their records and first stream dwords requested together */>
__device__ __forceinline__ void pileup_stream_body(const MkpVisit* __restrict__ visits, const uint8_t* __restrict__ cov,
    const MkpEvent* __restrict__ events,
                 const MkpSTile* __restrict__ tiles, uint32_t n_tiles, const MkpRunParams* __restrict__ prmp, const uint32_t* __restrict__ slot_pos,
                 const uint8_t* __restrict__ focus, const MkpCombo* __restrict__ combos, uint32_t* __restrict__ rows_base,
                     uint32_t* __restrict__ row_cursor,
                 uint32_t* __restrict__ tile_row_off, uint32_t* __restrict__ dev_err, uint32_t key_arg, uint32_t n_combos, uint32_t n_runs,
                     uint32_t S, uint32_t tal_words) {
  const uint32_t key_filter = KEYED ? (key_arg & 0xffffu) : 0u, key_run = key_arg >> 16;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t next_read, n_real, run_ticket, row_base_s, row_total_s;
  __shared__ uint32_t wave_tot[PILEUP_WAVES];
  __shared__ __attribute__((aligned(16))) uint32_t prm_lds[(sizeof(MkpRunParams) + 3) / 4];
  __shared__ __attribute__((aligned(16))) uint32_t combo_lds[64 * sizeof(MkpCombo) / 4];
  __shared__ StreamProg prog;
  // A workgroup takes the next TILE from an atomic ticket (row_cursor[0]; the host zeroes it before the first pass), not from blockIdx: rows
  // leave in genome order through a look-back over the runs before (mkp_dev_rows.hpp), and a run may only wait for runs whose workgroups
  // are already running — true for tickets whatever the dispatch order, with any number of contexts on the device.
  // The prologue is a chain of memory round trips in front of the first tally — ticket, tile record, visit records, stream dwords.  What does
  // not hang on it starts beside the ticket: the tallies are cleared at once (their size is a kernel argument), the parameter block and the
  // combos come in; the slot positions and focus bytes (needed by the emission only) are requested when the tile is known and land in LDS
  // behind the visits.  ONE barrier stands between a workgroup's start and its first visit (round 5: two, the second behind the positions).
  if (threadIdx.x == 0) { run_ticket = atomicAdd(row_cursor, 1u); next_read = 0; n_real = 0; }
  { const uint32_t nv = tal_words >> 2; uint4* l4 = reinterpret_cast<uint4*>(lds);
    for (uint32_t k = threadIdx.x; k < nv; k += PILEUP_THREADS) l4[k] = make_uint4(0u, 0u, 0u, 0u);
    for (uint32_t k = (nv << 2) + threadIdx.x; k < tal_words; k += PILEUP_THREADS) lds[k] = 0; }
  for (uint32_t kq = threadIdx.x; kq < sizeof(MkpRunParams) / 4; kq += PILEUP_THREADS) prm_lds[kq] = reinterpret_cast<const uint32_t*>(prmp)[kq];
  for (uint32_t kq = threadIdx.x; kq < n_combos * (sizeof(MkpCombo) / 4); kq += PILEUP_THREADS) combo_lds[kq] = reinterpret_cast<const uint32_t*>(combos)[kq];
  __syncthreads();
  const MkpRunParams& prm = *reinterpret_cast<const MkpRunParams*>(prm_lds);
  const MkpCombo* combos_l = reinterpret_cast<const MkpCombo*>(combo_lds);
  const uint32_t n_counters = rfl(prm.n_counters), n_oslots = rfl(prm.n_slots);
  uint32_t* __restrict__ tal = lds;
  uint32_t* __restrict__ obs = lds + n_counters * S;
  int32_t* __restrict__ fpos = reinterpret_cast<int32_t*>(lds + tal_words);
  // per slot: [0:7] focus byte, [8 + 6m ..] partner column of the m-th '+' motif (strand combining)
  uint32_t* __restrict__ aux = lds + tal_words + S;
  uint32_t* __restrict__ rowmap = lds + tal_words + 2u * S;
  const int lane = lane_id();
  const uint32_t wave = rfl(threadIdx.x >> 6);
  const uint32_t run = rfl(run_ticket);                               // row-run index: key pass * tiles + tile (passes run one after the other on the stream)
  const uint32_t tix = run - key_run * n_tiles;
  if (tix >= n_tiles) { if (threadIdx.x == 0) atomicOr(dev_err, ERR_ROW_CAP); return; }   // (cannot happen: one ticket per workgroup)
  const MkpSTile tl = tiles[tix];   // (uniform: scalar loads)
  const uint32_t gh0 = tl.gh0, gh1 = tl.gh1, n_tslots = gh1 - gh0;
  const uint32_t rid_first = tl.first, rid_end = tl.last;
  // this thread's slot: position now, focus byte behind the visits
  int32_t my_pos = 0;
  if (threadIdx.x < n_tslots) my_pos = (int32_t)slot_pos[gh0 + threadIdx.x];
  rowprog_build(prm, n_counters, prog);   // (threads 0..15; read after the visits' barrier)
  const uint32_t talbase = lds_addr(tal), S4 = S * 4u;
  // one visit: the read's bytes for this tile's slots (first dword per lane already in `wcur`), its observed codes, its overflow events
  auto visit = [&](const MkpVisit& v, uint32_t wcur) {
    const uint32_t a = max(v.gs0, gh0), b = min(v.gs0 + v.n_sl, gh1);
    if (a >= b) return;
    if (KEYED && (v.flags >> 8) != key_filter) return;   // --partition-tag: one pass per key
    const uint32_t k_lo = a - v.gs0, k_hi = b - v.gs0;      // the read's bytes for this tile
    const uint32_t col0 = v.gs0 - gh0;                      // column of byte k = col0 + k (mod 2^32)
    const uint8_t* __restrict__ cp = cov + v.cov_off;
    const uint32_t kfirst = (k_lo & ~3u) + 4u * (uint32_t)lane;
    // observed mod codes over the columns the read is in (add_mod_codes_for_record, pileup/mod.rs:831-835)
    if ((v.flags & MKP_VF_OK) && (v.obs0 | v.obs1)) {
      // one lane per (observed-code slot, tally strand): +1 where the read enters the tile's columns, -1 behind its last
      if (!(v.flags & MKP_VF_GAPS)) {
        const uint32_t sl = (uint32_t)lane >> 1, st = (uint32_t)lane & 1u;
        if (sl < n_oslots && (((st ? v.obs1 : v.obs0) >> sl) & 1u)) {
          const uint32_t inc = st ? 0x10000u : 1u, at = __umul24(sl, S) + (a - gh0);
          atomicAdd(&obs[at], inc);
          if (b - gh0 < n_tslots) atomicAdd(&obs[at + (b - a)], 0u - inc);
        }
      } else {   // ref-skips: the read is not in those columns (alignment.is_refskip()) — a change of state per boundary
        for (uint32_t k = k_lo + (uint32_t)lane; k <= k_hi; k += 64) {
          const bool cur = k < k_hi && cp[k] != MKP_FB_NONE, prev = k > k_lo && cp[k - 1u] != MKP_FB_NONE;
          const uint32_t col = col0 + k;
          if (cur != prev && col < n_tslots) for (uint32_t s = 0; s < 2; s++) {
            uint32_t m = s ? v.obs1 : v.obs0;
            const uint32_t inc = s ? 0x10000u : 1u;
            while (m) { const uint32_t sl = (uint32_t)__ffs((int)m) - 1u; m &= m - 1u; atomicAdd(&obs[sl * S + col], cur ? inc : 0u - inc); }
          }
        }
      }
    }
    // bytes outside [k_lo, k_hi) — the ends of the read's first and last dword in this tile — become "no feature": the first dword's (lane 0 of
    // the first round only) once per visit, the last dword's once per round
    uint32_t lom = kfirst < k_lo ? ~(0xffffffffu << (8u * (k_lo - kfirst))) : 0u;
    for (uint32_t k = kfirst;; k += 256u) {
      uint32_t w = wcur | lom;
      lom = 0;
      const uint32_t kn = k + 256u;
      wcur = kn < k_hi ? *reinterpret_cast<const uint32_t*>(cp + kn) : 0xffffffffu;
      if (k + 4u > k_hi) w |= k >= k_hi ? 0xffffffffu : 0xffffffffu << (8u * (k_hi - k));
      const uint32_t lane_base = talbase + 4u * (col0 + k);
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        // feature byte: [0:4] counter, [5] tally strand, >= 0x40 none.  One multiply-add for the row address (the byte's column goes into the
        // instruction's offset field), one for the increment (1 or 1 << 16)
        const uint32_t row = (w >> (8u * j)) & 31u, st = (w >> (8u * j + 5u)) & 1u;
        if (!(w & (0xc0u << (8u * j)))) {
          lds_u32* col = (lds_u32*)(uintptr_t)(lane_base + __umul24(row, S4));
          __hip_atomic_fetch_add(col + j, __umul24(st, 0xffffu) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      if (!__any(kn < k_hi)) break;
    }
    if (v.n_over) {   // second features on one column
      for (uint32_t k = (uint32_t)lane; k < v.n_over; k += 64) {
        const MkpEvent e = events[v.over_off + k];
        if (e.pos >= gh0 && e.pos < gh1) lds_add(talbase + 4u * (e.pos - gh0) + __umul24(e.info & 31u, S4), (e.info & 32u) ? 0x10000u : 1u);
      }
    }
  };
  // The tile's candidate reads [first, last) are a superset (everything that starts before the tile's end and behind the longest read's reach:
  // about half of them end before the tile starts, C3).  Every thread looks at one candidate — slot range only, one 8-byte load — and the ones
  // that reach into the tile are compacted into a list in LDS (the row map's words: the emission is not running yet); the waves then draw
  // from that list, FOUR at a time: the visit records and the first stream dword of each are requested before any of them is used (a visit is
  // a chain ticket -> 32-byte record -> one dword per lane -> LDS atomics).  Round 6 ablation (profiles/r06_stream_ablation.txt): with the
  // emission rewritten the visits are the kernel — 0.077 of 0.133 ms — and every empty candidate used to cost a ticket, a scalar load and its wait.
  for (uint32_t c0 = rid_first; c0 < rid_end; c0 += PILEUP_THREADS) {
#ifdef MKP_DEBUG
    if (prm.debug_skip & 1024u) break;   // ablation: no visits (prologue + scans + emission only)
#endif
    // (the round before is through with the list)
    if (c0 != rid_first) { __syncthreads(); if (threadIdx.x == 0) { next_read = 0; n_real = 0; } __syncthreads(); }
    { const uint32_t cnd = c0 + threadIdx.x; bool reaches = false;
      if (cnd < rid_end) {
        const uint2 gr = *reinterpret_cast<const uint2*>(&visits[cnd]);   // gs0, n_sl
        reaches = max(gr.x, gh0) < min(gr.x + gr.y, gh1);
        if (KEYED && reaches) reaches = (visits[cnd].flags >> 8) == key_filter;   // --partition-tag: one pass per key
      }
      const unsigned long long bal = __ballot(reaches);
      uint32_t w0 = 0; if (lane == 0 && bal) w0 = atomicAdd(&n_real, (uint32_t)__popcll(bal));
      w0 = rfl(w0);
      if (reaches) rowmap[w0 + (uint32_t)__popcll(bal & lanemask_lt())] = cnd; }
    __syncthreads();
    const uint32_t n_here = n_real;
    for (;;) {
      uint32_t base; { uint32_t ticket = 0; if (lane == 0) ticket = atomicAdd(&next_read, VB); base = rfl(ticket); }
      if (base >= n_here) break;
      MkpVisit vv[VB]; uint32_t ww[VB];
#pragma unroll
      for (uint32_t j = 0; j < VB; j++) vv[j] = visits[rfl(rowmap[min(base + j, n_here - 1u)])];   // (uniform: scalar loads)
#pragma unroll
      for (uint32_t j = 0; j < VB; j++) {
        const uint32_t a = max(vv[j].gs0, gh0), b = min(vv[j].gs0 + vv[j].n_sl, gh1);
        const uint32_t k_lo = a - vv[j].gs0, k_hi = b - vv[j].gs0, kfirst = (k_lo & ~3u) + 4u * (uint32_t)lane;
        ww[j] = (a < b && kfirst < k_hi) ? *reinterpret_cast<const uint32_t*>(cov + vv[j].cov_off + kfirst) : 0xffffffffu;
      }
#pragma unroll
      for (uint32_t j = 0; j < VB; j++) if (base + j < n_here) visit(vv[j], ww[j]);
    }
  }
  // the emission's per-slot words: position and focus byte (the byte's load overlaps the barrier and the scans below)
  uint32_t my_fv = 0;
  if (threadIdx.x < n_tslots) { fpos[threadIdx.x] = my_pos; my_fv = prm.has_focus ? (uint32_t)focus[my_pos - prm.win_start] : 3u; }
  __syncthreads();
  // observed-code difference arrays -> counts, in place and still packed
  for (uint32_t a = wave; a < n_oslots; a += PILEUP_WAVES) {
    uint32_t* __restrict__ arr = obs + a * S;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_tslots; b0 += 64) {
      const uint32_t vv = (b0 + lane < n_tslots) ? arr[b0 + lane] : 0u;
      const uint32_t sc = wave_incl_scan(vv);
      if (b0 + lane < n_tslots) arr[b0 + lane] = sc + carry;
      carry += (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
    }
  }
  if (threadIdx.x < n_tslots) aux[threadIdx.x] = my_fv;
  __syncthreads();

  // ---- E1: which rows does this thread's slot yield (FeatureVector::decode, pileup/mod.rs:412-446; combine_strand_features 469-561)
  const StreamProg& P = prog;
  const uint32_t n_groups = P.n_groups;
  const bool combine = prm.combine_strands != 0;
  const uint32_t i = threadIdx.x;
  unsigned long long em = 0; uint32_t cnt = 0, mult0 = 1, mult1 = 1;
#ifdef MKP_DEBUG
  const bool dbg_norows = (prm.debug_skip & 2048u) != 0;   // ablation: no rows (the look-back word is still published so that nothing waits)
#else
  constexpr bool dbg_norows = false;
#endif
  if (!dbg_norows && i < n_tslots && my_pos >= tl.r0 && my_pos < tl.r1 && (my_fv & 3u)) {
    const uint32_t rule = my_fv & 3u, combo = my_fv >> 2;
    if (!combine) {
      if (combo) { const MkpCombo& cb = combos_l[combo]; mult0 = cb.n_pos ? cb.n_pos : 1u; mult1 = cb.n_neg ? cb.n_neg : 1u; }
      for (uint32_t s = 0; s < 2; s++) {
        if (!((rule >> s) & 1u)) continue;
        for (uint32_t g = 0; g < n_groups; g++) if (stream_row_exists(tal, S, n_counters, P, s, i, g)) em |= 1ull << (16u * s + g);
      }
      cnt = (uint32_t)__popc((uint32_t)em & 0xffffu) * mult0 + (uint32_t)__popc((uint32_t)(em >> 16) & 0xffffu) * mult1;
    } else if (combo) {   // only '+' motif positions produce rows
      const MkpCombo& cb = combos_l[combo];
      const bool pos_ok = (rule & 1u) != 0;
      uint32_t part = 0;
      for (uint32_t m = 0; m < cb.n_pos; m++) {
        const int delta = cb.pos_delta[m];
        if (delta == -128) continue;   // not a palindrome / negative_strand_position() == None
        const int idx = cb.pos_ids[m];
        const int32_t qpos = my_pos + delta;
        bool neg_ok = false; uint32_t iq = i;
        if (delta != -127 && qpos >= prm.win_start && qpos < prm.win_end) {   // -127: the mate position is in another interval
          // the partner's column: a focus position at most MKP_HALO away — a few columns up or down
          if (delta > 0) { while (iq + 1u < n_tslots && fpos[iq + 1u] <= qpos) iq++; } else { while (iq > 0u && fpos[iq - 1u] >= qpos) iq--; }
          if (fpos[iq] == qpos) {
            const uint32_t fq = aux[iq] & 0xffu;
            if ((fq & 2u) && (fq >> 2)) { const MkpCombo& cq = combos_l[fq >> 2];
              for (uint32_t k = 0; k < cq.n_neg; k++) neg_ok |= (cq.neg_ids[k] == idx);
              }
          }
        }
        if (neg_ok) part |= ((iq - i + 32u) & 63u) << (8u + 6u * m);
        for (uint32_t g = 0; g < n_groups; g++) {
          if (!((P.info[g] >> 18) & 1u)) continue;   // grouped by code (BTreeMap)
          bool any = false;
          for (uint32_t gj = g; gj < n_groups && (gj == g || !((P.info[gj] >> 18) & 1u)); gj++)
            any = any || (pos_ok && stream_row_exists(tal, S, n_counters, P, 0, i, gj))
                || (neg_ok && stream_row_exists(tal, S, n_counters, P, 1, iq, gj));
          if (any) { em |= 1ull << (16u * m + g); cnt++; }
        }
      }
      if (part) aux[i] = my_fv | part;   // (own word; the other threads only look at the focus byte, which stays)
    }
  }
  // ---- E2: place of every slot's rows inside the tile, the tile's place in the row buffer
  const uint32_t inc2 = wave_incl_scan(cnt);
  if (lane == 63) wave_tot[wave] = inc2;
  __syncthreads();
  uint32_t off = inc2 - cnt, tile_rows = 0;
  for (uint32_t w2 = 0; w2 < PILEUP_WAVES; w2++) { const uint32_t t = wave_tot[w2]; if (w2 < wave) off += t; tile_rows += t; }
  MkpRowsDev rows;
  { const size_t cap = prm.row_capacity; uint32_t* q = rows_base;
    rows.pos = q; rows.info = q + cap; rows.code = q + 2 * cap; rows.n_valid = q + 3 * cap; rows.n_mod = q + 4 * cap; rows.n_can = q + 5 * cap;
      rows.n_other = q + 6 * cap;
    rows.n_del = q + 7 * cap; rows.n_fail = q + 8 * cap; rows.n_diff = q + 9 * cap; rows.n_nocall = q + 10 * cap; }
  for (uint32_t r0 = 0; r0 == 0u || r0 < tile_rows; r0 += MKP_STREAM_ROWMAP_WORDS) {
    if (r0) __syncthreads();   // the round before has read the map
    // (slot, candidate) of every row of this round: [0:9] slot, [10:13] group, [14:15] strand | motif, [16:17] which of the position's motif ids
    if (cnt && off < r0 + MKP_STREAM_ROWMAP_WORDS && off + cnt > r0) {
      uint32_t r = off;
      for (uint32_t sm = 0; sm < 4u; sm++) {
        uint32_t bits = (uint32_t)(em >> (16u * sm)) & 0xffffu;
        const uint32_t mult = combine ? 1u : (sm ? mult1 : mult0);
        while (bits) {
          const uint32_t g = (uint32_t)__ffs((int)bits) - 1u; bits &= bits - 1u;
          for (uint32_t k = 0; k < mult; k++, r++) if (r >= r0
              && r < r0 + MKP_STREAM_ROWMAP_WORDS) rowmap[r - r0] = i | (g << 10) | (sm << 14) | (k << 16);
        }
      }
    }
    if (r0 == 0u && wave == 0u) {   // tile_row_off = the runs' look-back words (two dwords each); row_cursor[1] = total rows, written by the last run
#ifdef MKP_DEBUG
      uint32_t base;
      // ablation: no look-back (rows in completion order)
      if (prm.debug_skip & 4096u) { uint32_t b0 = 0; if (lane == 0) b0 = atomicAdd(row_cursor + 1, tile_rows); base = rfl(b0); }
      else base = lookback_reserve_wave(reinterpret_cast<unsigned long long*>(tile_row_off), run, tile_rows);
#else
      const uint32_t base = lookback_reserve_wave(reinterpret_cast<unsigned long long*>(tile_row_off), run, tile_rows);
#endif
      if (lane == 0) {
#ifdef MKP_DEBUG
        if (!(prm.debug_skip & 4096u))
#endif
        if (run + 1u == n_runs) row_cursor[1] = base + tile_rows;
        uint32_t s = tile_rows;
        if (base + s > prm.row_capacity) { atomicOr(dev_err, ERR_ROW_CAP); s = 0; }
        row_base_s = base; row_total_s = s;
      }
    }
    __syncthreads();
    // ---- E3: one thread per row
    const uint32_t n_here = min(row_total_s, r0 + MKP_STREAM_ROWMAP_WORDS) > r0 ? min(row_total_s, r0 + MKP_STREAM_ROWMAP_WORDS) - r0 : 0u;
    for (uint32_t rr = threadIdx.x; rr < n_here; rr += PILEUP_THREADS) {
      const uint32_t e = rowmap[rr], si = e & 1023u, g = (e >> 10) & 15u, sm = (e >> 14) & 3u, k = (e >> 16) & 3u;
      const uint32_t ax = aux[si], combo = (ax & 0xffu) >> 2;
      RowAcc acc = {0, 0, 0, 0, 0, 0, 0, 0};
      uint32_t strand, motif1;   // motif1 = motif id + 1 (0: none)
      if (!combine) {
        stream_row_add(tal, S, P, sm, si, g, acc);
        strand = sm; motif1 = 0;
        if (combo) { const MkpCombo& cb = combos_l[combo]; const uint32_t n_ids = sm ? cb.n_neg : cb.n_pos;
          if (n_ids) motif1 = (uint32_t)(sm ? cb.neg_ids[k] : cb.pos_ids[k]) + 1u;
          }
      } else {
        const MkpCombo& cb = combos_l[combo];
        const uint32_t pd = (ax >> (8u + 6u * sm)) & 63u, iq = si + pd - 32u;
        const bool pos_ok = (ax & 1u) != 0, neg_ok = pd != 0u;
        for (uint32_t gj = g; gj < n_groups && (gj == g || !((P.info[gj] >> 18) & 1u)); gj++) {
          if (pos_ok && stream_row_exists(tal, S, n_counters, P, 0, si, gj)) stream_row_add(tal, S, P, 0, si, gj, acc);
          if (neg_ok && stream_row_exists(tal, S, n_counters, P, 1, iq, gj)) stream_row_add(tal, S, P, 1, iq, gj, acc);
        }
        strand = 2; motif1 = (uint32_t)cb.pos_ids[sm] + 1u;
      }
      const size_t at = (size_t)row_base_s + r0 + rr;
      rows.pos[at] = (uint32_t)fpos[si]; rows.info[at] = strand | (motif1 << 8) | (key_filter << 16); rows.code[at] = P.code[g];
      rows.n_valid[at] = acc.n_valid; rows.n_mod[at] = acc.n_mod; rows.n_can[at] = acc.n_can; rows.n_other[at] = acc.n_other;
      rows.n_del[at] = acc.n_del; rows.n_fail[at] = acc.n_fail; rows.n_diff[at] = acc.n_diff; rows.n_nocall[at] = acc.n_nocall;
    }
  }
}

#define STREAM_PARAMS const MkpVisit* __restrict__ visits, const uint8_t* __restrict__ cov, const MkpEvent* __restrict__ events, const MkpSTile* __restrict__ tiles, uint32_t n_tiles, \
    const MkpRunParams* __restrict__ prmp, const uint32_t* __restrict__ slot_pos, const uint8_t* __restrict__ focus, const MkpCombo* __restrict__ combos, uint32_t* __restrict__ rows_base, \
    uint32_t* __restrict__ row_cursor, uint32_t* __restrict__ tile_row_off, uint32_t* __restrict__ dev_err, uint32_t key_arg, uint32_t n_combos, uint32_t n_runs, uint32_t S, uint32_t tal_words
#define STREAM_PASS visits, cov, events, tiles, n_tiles, prmp, slot_pos, focus, combos, rows_base, row_cursor, tile_row_off, dev_err, key_arg, n_combos, n_runs, S, tal_words
#ifndef MKP_STREAM_VB
#define MKP_STREAM_VB 4
#endif
extern "C" __global__ void __launch_bounds__(PILEUP_THREADS, 8) mkp_pileup_stream(STREAM_PARAMS) {
  pileup_stream_body<false, MKP_STREAM_VB>(STREAM_PASS); }
extern "C" __global__ void __launch_bounds__(PILEUP_THREADS, 8) mkp_pileup_stream_keyed(STREAM_PARAMS) {
  pileup_stream_body<true, MKP_STREAM_VB>(STREAM_PASS); }

// ----------------------------------------------------------------------------------------------------------------------
// Records sharing a read name inside one interval (MkpDupCons / MkpDupSeg, mkp_device.h).  Rare: a handful of records per shard.
//   mkp_dup_restore  before the decode kernels: a consumer's header points at its OWN event slice again (a re-launch on a resident shard finds
//                    it pointing at the rebuilt list of the pass before)
//   mkp_dup_events   after the decode kernels: one wave per consumer rebuilds its event list segment by segment.  Own segment: its events are
//                    copied.  Foreign segment: the owner's events in the segment's positions — each is a (reference position, counter, mod
//                    strand, base) of the OWNER's call map — are kept where the consumer's own alignment has a base at that position and the
//                    base (read orientation) is the call's base (get_mod_call asks `calls[strand][read_base].get(position)`, read_cache.rs:232-297),
//                    with the tally strand of the consumer's alignment (add_feature, pileup/mod.rs:238-281).
//   mkp_dup_apply    the consumer's header / summary now describe the rebuilt list: cover / accumulate kernels take it as the record's own.
// The observed codes and the status (skip set, read_cache.rs:272-277) are the owner's too; the accumulate kernels apply ONE set per record, so
// owners that disagree across a consumer's segments raise ERR_DUP_MIXED (the host refuses the shard) instead of being approximated.
extern "C" __global__ void __launch_bounds__(64) mkp_dup_restore(MkpReadHdr* __restrict__ hdrs, const MkpDupCons* __restrict__ cons, uint32_t n) {
  const uint32_t k = blockIdx.x * 64u + threadIdx.x;
  if (k < n) hdrs[cons[k].rid].event_off = cons[k].own_off;
}
extern "C" __global__ void __launch_bounds__(64) mkp_dup_apply(MkpReadHdr* __restrict__ hdrs, MkpReadOut* __restrict__ readout,
    const MkpDupCons* __restrict__ cons, uint32_t n) {
  const uint32_t k = blockIdx.x * 64u + threadIdx.x;
  if (k >= n) return;
  const MkpDupCons c = cons[k];
  hdrs[c.rid].event_off = c.eff_off;
  MkpReadOut o; o.n_events = c.out_n; o.ok = c.out_ok; o.obs[0] = c.out_obs0; o.obs[1] = c.out_obs1;
  readout[c.rid] = o;
}
extern "C" __global__ void __launch_bounds__(64) mkp_dup_events(const MkpReadHdr* __restrict__ hdrs, const uint32_t* __restrict__ cigar,
    const uint8_t* __restrict__ seqs, MkpEvent* __restrict__ events,
                                                                const MkpReadOut* __restrict__ readout, MkpDupCons* __restrict__ cons,
                                                                    const MkpDupSeg* __restrict__ segs, uint32_t n, uint32_t* __restrict__ dev_err) {
  const uint32_t ci = blockIdx.x;
  if (ci >= n) return;
  const int lane = lane_id();
  const MkpDupCons c = cons[ci];
  const MkpReadHdr h = hdrs[c.rid];
  const uint32_t aln = (h.flags & MKP_RF_REVERSE) ? 1u : 0u, L = h.l_seq;
  const uint8_t* __restrict__ seqb = seqs + h.seq_off;
  const uint32_t* __restrict__ cg = cigar + h.cigar_off;
  RefWin rw; rw.c0 = 0; rw.q_run = 0; rw.r_run = h.ref_start; rw.Rtot = 0; rw.Qtot = 0; rw.re = rw.m1 = rw.m2 = rw.m3 = 0;
    rw.pk[0] = rw.pk[1] = rw.pk[2] = rw.pk[3] = 0; rw.loaded = false;
  rw.pref = cigar_quad(cg, h.n_cigar, 0);
  MkpEvent* __restrict__ dst = events + c.eff_off;
  uint32_t n_out = 0, st_ok = 0, st_o0 = 0, st_o1 = 0; bool have_st = false, mixed = false, over = false;
  for (uint32_t s = 0; s < c.n_seg; s++) {
    const MkpDupSeg sg = segs[c.seg_off + s];
    const MkpReadOut ro = readout[sg.owner];   // (the owner's own summary: mkp_dup_apply runs behind this kernel)
    const uint32_t ok = ro.ok == 1u ? 1u : 0u, o0 = ok ? ro.obs[0] : 0u, o1 = ok ? ro.obs[1] : 0u;
    if (!have_st) { st_ok = ok; st_o0 = o0; st_o1 = o1; have_st = true; } else if (ok != st_ok || o0 != st_o0 || o1 != st_o1) mixed = true;
    const uint32_t n_src = ok ? ro.n_events : 0u;
    const MkpEvent* __restrict__ src = events + sg.src_off;
    const bool own = sg.owner == c.rid;
    for (uint32_t at = event_lower_bound(src, n_src, sg.p_lo);; at += 64u) {
      const uint32_t k = at + (uint32_t)lane;
      MkpEvent e; e.pos = 0xffffffffu; e.info = 0;
      if (k < n_src) e = src[k];
      const bool in = k < n_src && (int32_t)e.pos < sg.p_hi;
      bool keep = in;
      if (!own) {
        // the consumer's own alignment at the call's reference position: a base (M / = / X), and the call's base in read orientation
        uint32_t kind = 2u, q = 0u;
        const bool inside = in && (int32_t)e.pos >= h.ref_start && (int32_t)e.pos < h.ref_end;
        refwin_map(rw, cg, h.n_cigar, h.ref_start, inside, (int32_t)e.pos, &kind, &q);
        keep = false;
        if (inside && kind == 0u && q < L) {
          const uint32_t byte = (uint32_t)seqb[q >> 1], nib = (q & 1u) ? (byte & 15u) : (byte >> 4);
          const int sb = nib2base(nib);
          const uint32_t rb = sb < 0 ? 4u : (aln ? 3u - (uint32_t)sb : (uint32_t)sb);
          keep = rb == ((e.info >> 9) & 3u);
        }
        // the mod strand of the owner's call (its tally strand seen from its own alignment), tallied from THIS record's alignment strand
        const uint32_t mod_strand = ((e.info >> 8) ^ (e.info >> 11)) & 1u;
        e.info = (e.info & ~((1u << 8) | (1u << 11))) | ((aln ^ mod_strand) << 8) | (aln << 11);
      }
      const unsigned long long bk = __ballot(keep);
      const uint32_t nk = (uint32_t)__popcll(bk);
      if (n_out + nk > c.eff_cap) { over = true; break; }
      if (keep) dst[n_out + (uint32_t)__popcll(bk & lanemask_lt())] = e;
      n_out += nk;
      if (!__all(in)) break;
    }
    if (over) break;
  }
  if (lane == 0) {
    if (mixed) atomicOr(dev_err, ERR_DUP_MIXED);
    if (over) atomicOr(dev_err, ERR_EVENT_CAP);
    MkpDupCons* o = cons + ci;
    o->out_n = st_ok ? n_out : 0u; o->out_ok = st_ok; o->out_obs0 = st_o0; o->out_obs1 = st_o1;
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// host-side launchers (called from mkp_api.cpp)
// work = the fused decoder's reads [longer than one base window | the others], cover_ids = the reads of mkp_cover_reads
extern "C" hipError_t mkp_launch_slots(hipStream_t st, const MkpWork* work, uint32_t n_long, uint32_t n_short, const MkpReadHdr* hdrs,
    const uint32_t* cover_ids, uint32_t n_cover, const uint32_t* cigar,
                                       const uint8_t* seqs, const MkpTagRef* tagref, const uint32_t* ranks, const uint8_t* ml,
                                           const MkpLayout* layouts, const MkpFusedDesc* fdesc, const MkpRunParams* prm,
                                       const uint32_t* slot_pos, uint8_t* cov, MkpVisit* visits, MkpEvent* events, MkpReadOut* readout,
                                           uint32_t* dev_err) {
#define MKP_FUSED_LAUNCH(K, W, N) hipLaunchKernelGGL(K, dim3(((N) + 3u) / 4u), dim3(256), 0, st, W, N, cigar, seqs, ranks, ml, fdesc, *prm, slot_pos, cov, visits, readout)
  if (n_long) MKP_FUSED_LAUNCH(mkp_decode_slots_long, work, n_long);
  if (n_short) MKP_FUSED_LAUNCH(mkp_decode_slots, work + n_long, n_short);
  if (n_cover) hipLaunchKernelGGL(mkp_cover_reads, dim3((n_cover + 3u) / 4u), dim3(256), 0, st, hdrs, n_cover, cover_ids, cigar, seqs, tagref, ranks,
      ml, layouts, fdesc, *prm, slot_pos, cov, visits, events, readout, dev_err);
  return hipGetLastError();
}

extern "C" hipError_t mkp_launch_dup_restore(hipStream_t st, MkpReadHdr* hdrs, const MkpDupCons* cons, uint32_t n) {
  if (n) hipLaunchKernelGGL(mkp_dup_restore, dim3((n + 63u) / 64u), dim3(64), 0, st, hdrs, cons, n);
  return hipGetLastError();
}
extern "C" hipError_t mkp_launch_dup_events(hipStream_t st, MkpReadHdr* hdrs, const uint32_t* cigar, const uint8_t* seqs, MkpEvent* events,
    MkpReadOut* readout, MkpDupCons* cons, const MkpDupSeg* segs,
                                            uint32_t n, uint32_t* dev_err) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(mkp_dup_events, dim3(n), dim3(64), 0, st, hdrs, cigar, seqs, events, readout, cons, segs, n, dev_err);
  hipLaunchKernelGGL(mkp_dup_apply, dim3((n + 63u) / 64u), dim3(64), 0, st, hdrs, readout, cons, n);
  return hipGetLastError();
}

extern "C" hipError_t mkp_stream_set_lds(uint32_t bytes) {
  for (const void* k : {(const void*)mkp_pileup_stream, (const void*)mkp_pileup_stream_keyed}) {
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

extern "C" hipError_t mkp_launch_stream(hipStream_t st, uint32_t lds_bytes, const MkpVisit* visits, const uint8_t* cov, const MkpEvent* events,
    const MkpSTile* tiles, uint32_t n_tiles,
                                        const MkpRunParams* prm_dev, const uint32_t* slot_pos, const uint8_t* focus, const MkpCombo* combos,
                                            const MkpRowsDev* rows, uint32_t* row_cursor,
                                        uint32_t* tile_row_off, uint32_t* tile_row_cnt, uint32_t* dev_err, uint32_t key_filter, uint32_t key_slot,
                                            uint32_t n_combos, uint32_t n_runs, uint32_t slot_cap, uint32_t words_per_slot) {
  if (!n_tiles) return hipSuccess;
  (void)tile_row_cnt;
  const bool keyed = key_filter != MKP_NO_KEY_FILTER;
  const uint32_t key_arg = keyed ? ((key_filter & 0xffffu) | (key_slot << 16)) : 0u;
#define MKP_STREAM_LAUNCH(K) hipLaunchKernelGGL(K, dim3(n_tiles), dim3(PILEUP_THREADS), lds_bytes, st, visits, cov, events, tiles, n_tiles, prm_dev, slot_pos, focus, combos, rows->pos, row_cursor, \
                                                tile_row_off, dev_err, key_arg, n_combos > 64u ? 64u : n_combos, n_runs, slot_cap, words_per_slot * slot_cap)
  // ONE build per kernel: a one-shot shard pass and a re-launch on the resident shard run the same code object
  if (keyed) MKP_STREAM_LAUNCH(mkp_pileup_stream_keyed); else MKP_STREAM_LAUNCH(mkp_pileup_stream);
  return hipGetLastError();
}
