// gfx950 (CDNA4, wave64) kernels of the modkit-pileup hot path, part 1: the event pipeline.  Integer / byte work bound by
// instruction issue, LDS atomics and HBM — no MFMA.  (Part 2, mkp_slots.hip, is the slot pipeline focus runs take; it reuses the
// decode kernels below for the read classes its fused decoder does not cover.)  One device pass here, in order:
//
//   decode, one wave per read, one kernel per read class (host-built id lists, longest reads first):
//     mkp_decode_sparse1/2   one explicit-mode ('?') group, one shared delta list: 4096-base steps that only count (nibble flags,
//                            byte-packed popcounts, DPP scan); the listed ranks are located per call (lane search + SWAR select)
//     mkp_decode_fast1/2     one group, any mode / differing delta lists: 1024-base steps, ordinal bitmap in LDS + 4-bit deposit
//     mkp_decode_reads       everything else (<= 8 tags, `N` tags, duplex, repeated codes): combine_positions_to_probs per group
//     mkp_merge_duplex       duplex reads decoded one group per wave: interleave the two position-sorted event lists
//   all share the consumer: per 64 queued calls ML -> f32 BaseModProbs, edge filter, ReDistribute collapse,
//   MultipleThresholdModCaller::call, CIGAR mapping (register window) and one packed 8-byte event per mapped call.
//   mkp_sample_* = the same walks emitting argmax probabilities / summary classes (threshold estimate, sample-probs, summary).
//
//   mkp_pileup_tiles[_focus|_hemi][_keyed]   accumulate + emit: one 1024-thread workgroup per tile, two per CU; LDS holds the
//                            tile's 16-bit-packed strand tallies; waves draw the tile's reads from an LDS ticket (observed-code
//                            difference arrays, the read's event slice, the depth walk over its CIGAR); rows are produced straight
//                            from LDS (mkp_dev_rows.hpp).  Used for runs without focus positions and for pileup-hemi.
//   mkp_hemi_failed_reads    pileup-hemi: the one NoCall per interval a record with failing tags leaves
//   mkp_scan_tiles, mkp_gather_rows   order the tiles' row runs (exclusive scan of the counts + coalesced copy).
//   mkp_sample_accumulate / _hist1, mkp_summary_accumulate   summaries of the threshold sample (two-level histograms), summary counts.
//
// Semantics follow /root/reference/src (cited inline); arithmetic that must be bit-exact is f32
// with contraction off (-ffp-contract=off) and IEEE division.
#include "mkp_dev_common.hpp"
#include "mkp_dev_rows.hpp"


// One wave per read.  The read is walked 512 bases at a time (one SEQ dword = 8 bases per lane):
//   1. per lane: match masks of the stored bases the read's MM tags count, popcounts, wave prefix sums
//      (DeltaListConverter's cumulative counts, mod_bam.rs:667-684, 8 bases per instruction);
//   2. per tag: the calls whose rank falls into the chunk's rank window are located (prefix search + select)
//      and marked in a per-lane 8-bit "call here" mask held in LDS;
//   3. the union of the masks (plus every occurrence of an implicit-mode base) is enumerated, 64 positions
//      per batch: BaseModProbs are rebuilt in MM order, edge filter, ReDistribute collapse and
//      MultipleThresholdModCaller::call in f32, CIGAR mapping through a 64-op window, and one packed
//      8-byte event per mapped call is appended (ballot-compacted, position order).
// SAMPLE = threshold-sampling pass: emits argmax probabilities instead of call events.
// arr[j] for a wave-uniform j < N (N is tiny: a select chain, no scratch)
template <int N> __device__ __forceinline__ uint32_t selN(const uint32_t* a, uint32_t j) {
  uint32_t r = 0;   // OR of masked elements: a select chain over array elements is folded into an indexed (scratch) load
#pragma unroll
  for (int i = 0; i < N; i++) r |= (j == (uint32_t)i) ? a[i] : 0u;
  return r;
}

// NT = compile-time bound on the read's MM tag count (the entry point dispatches on it): every per-tag loop and
// register array below is sized for the layout actually present instead of NT.
template <bool SAMPLE, int NT>
__device__ __forceinline__ void decode_read_body(const MkpReadHdr* __restrict__ hdrs, uint32_t n_reads, const uint32_t* __restrict__ cigar,
                 const uint8_t* __restrict__ seqs, const MkpTagRef* __restrict__ tagref, const uint32_t* __restrict__ ranks,
                 const uint8_t* __restrict__ ml, const MkpLayout* __restrict__ layouts, const MkpRunParams& prm,
                 MkpEvent* __restrict__ events, MkpReadOut* __restrict__ readout, uint32_t* __restrict__ dev_err,
                 const uint8_t* __restrict__ bedmask, float* __restrict__ sample_vals, const uint32_t* __restrict__ read_ids,
                     uint32_t* __restrict__ lds_layouts, uint32_t (*__restrict__ lds_marks)[64], const uint8_t* __restrict__ pdep4) {
  const int lane = lane_id();
  // wave-uniform values are made provably uniform (readfirstlane) so they live in SGPRs and load through the scalar cache
  const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t widx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6))) + wib;
  if (widx >= n_reads) return;
  const uint32_t rid = (uint32_t)__builtin_amdgcn_readfirstlane((int)read_ids[widx]);
  const MkpReadHdr h = hdrs[rid];
  MkpReadOut out; out.n_events = 0; out.ok = 0; out.obs[0] = 0; out.obs[1] = 0;
  if ((h.flags & MKP_RF_BAD) || h.n_tags == 0) { if (lane == 0) readout[rid] = out; return; }
  // the read's layout (1216 B) goes to LDS once; every table lookup below is an LDS read
  constexpr int NB = NT < 4 ? NT : 4;   // distinct stored bases the tags can count
  uint32_t* __restrict__ lds_lay = lds_layouts + wib * MKP_LAYOUT_DWORDS;
  uint32_t* __restrict__ ordb = &lds_marks[wib * 7][0];                        // [NT][18] ordinal / position bitmaps of the chunk
  uint16_t* __restrict__ slots = reinterpret_cast<uint16_t*>(ordb + 192);                 // 512 compacted {lane, bit} entries
  { const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(&layouts[h.layout]);
    for (int i = lane; i < MKP_LAYOUT_DWORDS; i += 64) lds_lay[i] = src[i]; }
  __builtin_amdgcn_wave_barrier();
  const MkpLayout* lay = reinterpret_cast<const MkpLayout*>(lds_lay);
  const uint32_t* __restrict__ seqw = reinterpret_cast<const uint32_t*>(seqs + h.seq_off);  // reads start 4-byte aligned
  const bool rev = (h.flags & MKP_RF_REVERSE) != 0;
  const uint32_t L = h.l_seq;
  const uint32_t nd = (L + 7u) >> 3;
  const uint32_t aln = rev ? 1u : 0u;
  const int n_tags = (int)h.n_tags;

  // per-tag cursors into the sorted rank lists: a merge join against the read's bases, 512 at a time
  uint32_t t_off[NT], t_n[NT], t_ml[NT], t_cur[NT];
  MkpTagDesc t_desc[NT];
  uint32_t sbase[NB], tslot[NT], nb = 0;   // base slots: the distinct stored bases (A,C,G,T = 0..3) some tag counts
#pragma unroll
  for (int j = 0; j < NB; j++) sbase[j] = 0;
#pragma unroll
  for (int t = 0; t < NT; t++) {
    t_off[t] = 0; t_n[t] = 0; t_ml[t] = 0; t_cur[t] = 0; tslot[t] = 0; t_desc[t] = lay->tags[t];
    if (t < n_tags) {
      const MkpTagRef tr = tagref[h.tag_off + t]; t_off[t] = tr.rank_off; t_n[t] = tr.n; t_ml[t] = tr.ml_off; t_cur[t] = rev ? tr.n : 0u;
      if (t_desc[t].fb != 4) {
        const uint32_t xb = (uint32_t)(rev ? 3 - t_desc[t].fb : t_desc[t].fb);
        uint32_t found = nb;
#pragma unroll
        for (int j = 0; j < NB; j++) if ((uint32_t)j < nb && sbase[j] == xb) found = (uint32_t)j;
        if (found == nb) {
#pragma unroll
          for (int j = 0; j < NB; j++) if ((uint32_t)j == nb) sbase[j] = xb;
          nb++;
        }
        tslot[t] = (uint32_t)__builtin_amdgcn_readfirstlane((int)found);
      }
    }
  }
  nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
  // slots every occurrence of whose base is a call: groups with implicit-mode members (mod_bam.rs:1265-1292)
  uint32_t implslots = 0;
#pragma unroll
  for (int j = 0; j < NB; j++) {
    sbase[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)sbase[j]);
    const uint32_t b = rev ? 3u - sbase[j] : sbase[j];
    const uint32_t m0 = lds_lay[MKP_LAYOUT_GROUP_DW + b * 32], m1 = lds_lay[MKP_LAYOUT_GROUP_DW + (4 + b) * 32];
    if ((uint32_t)j < nb && (MKP_G_IMPL(m0) | MKP_G_IMPL(m1))) implslots |= 1u << j;
  }
  implslots = (uint32_t)__builtin_amdgcn_readfirstlane((int)implslots);
  // reverse reads need the totals up front (forward rank = total - inclusive count in stored order)
  uint32_t tot[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) tot[j] = 0;
  if (rev && nb) {
    uint32_t acc[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) acc[j] = 0;
    for (uint32_t d0 = 0; d0 < nd; d0 += 64) {
      const uint32_t d = d0 + lane;
      const uint32_t xl = linearize(d < nd ? seqw[d] : 0u);
      const int nv = min(max((int)L - (int)(8u * d), 0), 8);
      const uint32_t vmask = (1u << nv) - 1u;
#pragma unroll
      for (int j = 0; j < NB; j++) if ((uint32_t)j < nb) acc[j] += (uint32_t)__popc(match8(xl, (int)sbase[j]) & vmask);
    }
#pragma unroll
    for (int j = 0; j < NB; j++) if ((uint32_t)j < nb) tot[j] = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(acc[j]), 63);
  }
  bool err = false;
  const bool trimmable = !prm.edge_filter || !(L <= prm.edge_start || L <= prm.edge_end);  // read_can_be_trimmed (mod_bam.rs:1668-1671)
  const bool collapse = prm.numeric_mode == 2;

  uint32_t obs0 = 0, obs1 = 0, contrib_lo = 0, contrib_hi = 0;  // contrib: 8 groups x 8 tag bits
  bool any_surviving = false;
  uint32_t n_ev = 0;
  uint32_t cum[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) cum[j] = 0;
  uint32_t x_next = (uint32_t)lane < nd ? seqw[lane] : 0u;   // the next step's SEQ dword is always in flight
  // CIGAR window: 64 ops in registers, advanced as the walk moves along the read
  uint32_t c0 = 0, wq0 = 0, wq1 = 0; int32_t wr0 = h.ref_start;
  uint32_t w_op = 5u, w_qe = 0, w_qs = 0; int32_t w_rs = 0; uint32_t w_rtot = 0;
  bool win_loaded = false;

  for (uint32_t d0 = 0; d0 < nd && !err; d0 += 64) {
    const uint32_t d = d0 + lane;
    const uint32_t xl = linearize(x_next);
    { const uint32_t dn = d + 64u; x_next = dn < nd ? seqw[dn] : 0u; }
    const int nv = min(max((int)L - (int)(8u * d), 0), 8);
    const uint32_t vmask = (1u << nv) - 1u;
    uint32_t m8[NB], incl[NB], cnt[NB];
    uint32_t U = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      m8[j] = 0; incl[j] = 0; cnt[j] = 0;
      if ((uint32_t)j < nb) {
        m8[j] = match8(xl, (int)sbase[j]) & vmask; incl[j] = wave_incl_scan((uint32_t)__popc(m8[j]));
          cnt[j] = (uint32_t)__builtin_amdgcn_readlane((int)incl[j], 63);
        if ((implslots >> j) & 1u) U |= m8[j];
      }
    }
    const uint32_t qa = 8u * d0, qb = min(qa + 512u, L);
    uint32_t cur_before[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) { cur_before[t] = t_cur[t]; if (t < n_tags && lane < 18) ordb[t * 18 + lane] = 0; }
    // ---- 2. mark the calls of every tag that fall into this chunk
#pragma unroll
    for (int t = 0; t < NT; t++) {
      if (t >= n_tags || (prm.debug_skip & 32u)) break;
      const MkpTagDesc dsc = t_desc[t];
      const uint32_t sj = tslot[t];
      uint32_t wlo, whi;   // window of keys this chunk can ask for (windows of successive chunks tile the key space)
      if (dsc.fb == 4) { wlo = rev ? (L - qb) : qa; whi = rev ? (L - qa) : qb; }
      else { const uint32_t c = selN<NB>(cum, sj), n = selN<NB>(cnt, sj); wlo = rev ? (selN<NB>(tot, sj) - c - n) : c; whi = wlo + n; }
      const uint32_t kbase = dsc.fb == 4 ? qa : selN<NB>(cum, sj), ktot = dsc.fb == 4 ? L : selN<NB>(tot, sj);
      for (;;) {
        uint32_t e; bool hit; uint32_t nh;
        if (!rev) { const uint32_t i = t_cur[t] + lane; const bool valid = i < t_n[t]; e = valid ? ranks[t_off[t] + i] : 0xffffffffu;
          hit = valid && e < whi; nh = (uint32_t)__popcll(__ballot(hit)); t_cur[t] += nh; }
        // an entry >= whi is past the last occurrence: never consumed -> error at the end
        else { const uint32_t i = t_cur[t] - 64u + lane; const bool valid = (int32_t)i >= 0 && i < t_cur[t]; e = valid ? ranks[t_off[t] + i] : 0u;
          hit = valid && e >= wlo && e < whi; nh = (uint32_t)__popcll(__ballot(hit)); t_cur[t] -= nh; }
        if (nh == 0) break;
        // chunk-relative stored ordinal of the call's base (specific-base tags) or stored position (`N` tags): one bit in LDS
        const uint32_t ib = (rev ? (ktot - 1u - e) : e) - kbase;
        if (hit) atomicOr(&ordb[t * 18 + (ib >> 5)], 1u << (ib & 31u));
        if (nh < 64) break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t pack_t[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      pack_t[t] = 0;
      if (t < n_tags) {
        uint32_t bm;
        if (t_desc[t].fb == 4) bm = (ordb[t * 18 + (lane >> 2)] >> ((lane & 3) * 8)) & 0xffu & vmask;
        else {  // bits [excl, excl+cnt) of the ordinal bitmap deposited onto the set bits of the lane's match mask
          const uint32_t m = selN<NB>(m8, tslot[t]), c = (uint32_t)__popc(m), ex = selN<NB>(incl, tslot[t]) - c;
          const uint32_t w0 = ordb[t * 18 + (ex >> 5)], w1 = ordb[t * 18 + (ex >> 5) + 1];
          const uint32_t f = __builtin_amdgcn_alignbit(w1, w0, ex & 31u) & ((1u << c) - 1u);
          const uint32_t clo = (uint32_t)__popc(m & 15u);
          bm = (uint32_t)pdep4[((m & 15u) << 4) | (f & 15u)] | ((uint32_t)pdep4[(m & 0xf0u) | ((f >> clo) & 15u)] << 4);
        }
        U |= bm; const uint32_t c2 = (uint32_t)__popc(bm); pack_t[t] = bm | ((wave_incl_scan(c2) - c2) << 8);
      }
    }
    const uint32_t ucnt = (uint32_t)__popc(U);
    const uint32_t uincl = wave_incl_scan(ucnt);
    const uint32_t H = (uint32_t)__builtin_amdgcn_readlane((int)uincl, 63);
    // compaction: every lane writes {lane, bit} of its called positions into the wave's slot list, in read order
    { uint32_t ut = U, sidx = uincl - ucnt;
      while (ut) { slots[sidx++] = (uint16_t)(((uint32_t)lane << 3) | ((uint32_t)__ffs((int)ut) - 1u)); ut &= ut - 1u; } }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- 3. the called positions, 64 per batch, in read order
    for (uint32_t g0 = 0; g0 < H && !err && !(prm.debug_skip & 16u); g0 += 64) {
      const uint32_t g = g0 + lane;
      const bool active = g < H;
      const uint32_t slot = active ? (uint32_t)slots[g] : 0u;
      const int owner = (int)(slot >> 3);
      const uint32_t bit = slot & 7u;
      const uint32_t xo = __shfl(xl, owner, 64);
      const uint32_t q = 8u * (d0 + (uint32_t)owner) + bit;
      const int x = active ? nib2base((xo >> (4u * bit)) & 15u) : -1;
      const uint32_t f = rev ? (L - 1 - q) : q;  // forward (as-sequenced) position
      const int b = x < 0 ? -1 : (rev ? 3 - x : x);
      GState S0, S1;
#pragma unroll
      for (int k = 0; k < MKP_KMAX; k++) { at(S0.pk, k) = 0.f; at(S1.pk, k) = 0.f; }
      S0.H = S0.setmask = S1.H = S1.setmask = 0;
      {
#pragma unroll
      for (int t = 0; t < NT; t++) {
        if (t >= n_tags) break;
        const MkpTagDesc dsc = t_desc[t];
        const uint32_t pt = __shfl(pack_t[t], owner, 64);
        const bool found = active && ((pt >> bit) & 1u);
        if (!found) continue;
        if (x < 0) { err = true; continue; }  // a call listed on a non-ACGT base (DnaBase::try_from, mod_bam.rs:1245)
        const uint32_t idx = (pt >> 8) + (uint32_t)__popc(pt & ((1u << bit) - 1u));  // index among the tag's calls of this chunk, read order
        const uint32_t jx = rev ? (cur_before[t] - 1u - idx) : (cur_before[t] + idx);
        const uint32_t tm = lay->tagmap[t][b];  // [0:3] member index, [4+4i : 8+4i) local code of the tag's i-th code
        const uint32_t mi = tm & 15u;
        // get_base_mod_probs (mod_bam.rs:1242-1263): stride = #codes of the tag
        F4 ts = {0.f, 0.f, 0.f, 0.f};
        uint32_t seen = 0;
        for (int i = 0; i < (int)dsc.n_codes; i++) {
          const float p = ((float)ml[t_ml[t] + jx * dsc.n_codes + i] + 0.5f) / 256.0f;  // quals_to_probs 808-816
          const uint32_t kk = (tm >> (4 + 4 * i)) & 15u;
#pragma unroll
          for (int k = 0; k < MKP_KMAX; k++) if ((uint32_t)k == kk) {
            if (seen & (1u << k)) { if (at(ts, k) + p > 1.01f) err = true; at(ts, k) = at(ts, k) + p; } else { at(ts, k) = p; seen |= 1u << k; }
          }
        }
        if (dsc.neg) { if (merge_tag(S1, ts, seen, mi)) err = true; } else { if (merge_tag(S0, ts, seen, mi)) err = true; }
      }
      }
      // reference position through the CIGAR window (aligned pairs: M/=/X only, util.rs:122-145)
      bool mapped = false; int32_t rpos = 0;
      {
        bool pending = active && x >= 0;
        for (;;) {
          if (!win_loaded || (__any(pending && q >= wq1) && !__any(pending && q < wq1))) {
            // load / advance the window
            if (win_loaded) { c0 += 64; wq0 = wq1; wr0 += (int32_t)w_rtot; }
            if (c0 >= h.n_cigar) break;
            const uint32_t w = (c0 + lane < h.n_cigar) ? cigar[h.cigar_off + c0 + lane] : 5u /*0H*/;
            w_op = w & 15u; const uint32_t len = w >> 4;
            const uint32_t qlen = op_consumes_query(w_op) ? len : 0u, rlen = op_consumes_ref(w_op) ? len : 0u;
            w_qe = wave_incl_scan(qlen); const uint32_t re = wave_incl_scan(rlen);
            w_qs = wq0 + w_qe - qlen; w_rs = wr0 + (int32_t)(re - rlen);
            wq1 = wq0 + (uint32_t)__builtin_amdgcn_readlane((int)w_qe, 63); w_rtot = (uint32_t)__builtin_amdgcn_readlane((int)re, 63);
            win_loaded = true;
            continue;
          }
          const bool ready = pending && q < wq1;
          const int oi = find_op(w_qe, ready ? q - wq0 : 0u) & 63;
          const uint32_t my_qs = __shfl(w_qs, oi, 64), my_op = __shfl(w_op, oi, 64);
          const int32_t my_rs = __shfl(w_rs, oi, 64);
          if (ready) { mapped = op_is_match(my_op); rpos = my_rs + (int32_t)(q - my_qs); pending = false; }
          if (!__any(pending)) break;
        }
      }
      uint32_t ev_info[2]; float sv[2] = {0.f, 0.f}; uint32_t ev_cnt = 0; int32_t ev_pos = 0;
      if (active && x >= 0) {
        const bool edge_keep = !prm.edge_filter ||
            (prm.edge_inverted ? (f < prm.edge_start || f >= L - prm.edge_end) : (f >= prm.edge_start && f < L - prm.edge_end));
        bool dec_done = false;
#pragma unroll
        for (int sg = 0; sg < 2; sg++) {
          const uint32_t* gp = lds_lay + MKP_LAYOUT_GROUP_DW + (sg * 4 + b) * 32;
          const uint32_t gmisc = gp[0];
          const int nm = (int)MKP_G_NMEM(gmisc);
          if (nm == 0) continue;
          F4& spk = sg ? S1.pk : S0.pk;
          const uint32_t SH = sg ? S1.H : S0.H;
          const uint32_t impl = MKP_G_IMPL(gmisc);
          int pat;
          uint32_t member_contrib;
          if (SH) {
            if (impl & ~SH) err = true;  // ExplicitConflictInferred
            pat = (int)SH; member_contrib = SH;
          } else if (impl) {
            pat = MKP_PAT_INFERRED; member_contrib = impl;  // implicit fill (mod_bam.rs:1265-1292)
          } else continue;
          const GroupRegs gr = load_group(gp);
          const uint32_t pv = gp[12 + pat];
          uint32_t tagbits = 0;
#pragma unroll
          for (int mi = 0; mi < MKP_MAX_MEMBERS; mi++) if (member_contrib & (1u << mi)) tagbits |= 1u << ((gr.member_tags >> (4 * mi)) & 15u);
          const int gi = sg * 4 + b;
          if (gi < 4) contrib_lo |= tagbits << (8 * gi); else contrib_hi |= tagbits << (8 * (gi - 4));
          if (!trimmable || !edge_keep) continue;
          if (SAMPLE) {  // SeqPosBaseModProbs::filter_positions (read_ids_to_base_mod_probs.rs:966-1070)
            bool keep = !prm.only_mapped || mapped;
            if (prm.has_focus) keep = keep && mapped && rpos >= prm.win_start && rpos < prm.win_end
                && ((bedmask[rpos - prm.win_start] >> (aln ^ (uint32_t)sg)) & 1u);
            if (!keep) continue;
            any_surviving = true;
            // 2 `summary`: thresholded + argmax class; 3 `extract calls`: + forward position, mod strand, inferred, call_prob
            if (prm.sample_mode >= 2) {
              // the collapse left no code in the map: no profile row (iter_probs is empty)
              if (prm.sample_mode == 3 && ((pv >> 3) & 7u) == 0u) continue;
              uint32_t ob = 0; float am = 0.f; uint32_t inf = summary_info(gr, pv, spk, collapse, &ob, MKP_KMAX, &am);
              if (prm.sample_mode == 3) { inf |= ((uint32_t)sg << 2) | ((pat == MKP_PAT_INFERRED ? 1u : 0u) << 3); ev_pos = (int32_t)f; }
              sv[ev_cnt] = prm.sample_mode == 3 ? am : 0.f; ev_info[ev_cnt++] = inf; obs0 |= ob; continue;
            }
            sv[ev_cnt] = argmax_group(gr, pv, spk, collapse);
            ev_info[ev_cnt++] = MKP_G_TB(gr.misc);
            continue;
          }
          any_surviving = true;
          uint32_t ob = 0;
          const int cls = call_group(gr, pv, spk, collapse, &ob);
          const uint32_t tally = aln ^ (uint32_t)sg;  // read_cache.rs:181-188 / FeatureVector::add_feature
          if (tally) obs1 |= ob; else obs0 |= ob;
          if (mapped) {
            const uint32_t cid = cls == 0 ? (uint32_t)MKP_C_FAIL : cls == 1 ? MKP_G_CIDCAN(gr.misc) : ((gr.cids >> (8 * (cls - 2))) & 0xffu);
            ev_info[ev_cnt++] = cid | (tally << 8) | ((uint32_t)b << 9) | (aln << 11) | (dec_done ? 0u : (1u << 12));
            dec_done = true;
            ev_pos = rpos;
          }
        }
      }
      // ballot-compacted, position-ordered append of this batch's events
      unsigned long long b1 = __ballot(ev_cnt >= 1), b2 = __ballot(ev_cnt >= 2);
      uint32_t step_total = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2);
      if (step_total) {
        unsigned long long mlt = lanemask_lt();
        uint32_t off = n_ev + (uint32_t)__popcll(b1 & mlt) + (uint32_t)__popcll(b2 & mlt);
        if (n_ev + step_total > h.event_cap) { err = true; if (lane == 0) atomicOr(dev_err, ERR_EVENT_CAP); }
        else for (uint32_t e2 = 0; e2 < ev_cnt; e2++) {
          MkpEvent ev; ev.pos = (uint32_t)ev_pos; ev.info = ev_info[e2]; events[h.event_off + off + e2] = ev;
          if (SAMPLE) sample_vals[h.event_off + off + e2] = sv[e2];
        }
        n_ev += step_total;
      }
      err = __any(err);
    }
#pragma unroll
    for (int j = 0; j < NB; j++) cum[j] += cnt[j];
    err = __any(err);
  }
  // a delta list must not run past the last occurrence of its base / the end of the read: every entry must have
  // been consumed by the join (mod_bam.rs:705-727, 750-756)
#pragma unroll
  for (int t = 0; t < NT; t++) if (t < n_tags) { if (rev ? (t_cur[t] != 0u) : (t_cur[t] != t_n[t])) err = true; }
  err = __any(err);
  obs0 = wave_or(obs0); obs1 = wave_or(obs1);
  contrib_lo = wave_or(contrib_lo); contrib_hi = wave_or(contrib_hi);
  any_surviving = __any(any_surviving);
  // InvalidImplicitMode: a group all of whose contributing tags have no mode character (read_cache.rs:122-137)
  if (!prm.force_allow && !SAMPLE) {
    for (int gi = 0; gi < 8; gi++) {
      uint32_t m = ((gi < 4 ? contrib_lo >> (8 * gi) : contrib_hi >> (8 * (gi - 4)))) & 0xffu;
      if (m && (m & ~(uint32_t)lay->default_mask) == 0) err = true;
    }
  }
  if (lane == 0) {
    if (!err && any_surviving) { out.ok = 1; out.n_events = n_ev; out.obs[0] = obs0; out.obs[1] = obs1; }
    readout[rid] = out;
  }
}


// ----------------------------------------------------------------------------------------------
// Decode, FAST layouts (MkpLayout::fast: every tag on the same specific base and mod strand, no code listed twice —
// `C+m?`, `C+hm?`, `C+h?;C+m?`, ...; NT <= 2 tags).  Same semantics as decode_read_body, restructured as a
// producer/consumer inside the wave: the 1024-base steps (16 bases per lane) only *locate* calls and append {stored position, ML index per
// tag} to a queue in LDS; whenever 64 calls are queued one full batch runs the per-call work (ML -> f32, collapse,
// threshold caller, CIGAR mapping, event append).  With CpG data a step finds ~13 calls, so batches run at full
// lane occupancy instead of ~20 %.  The group descriptor, thresholds and code maps are wave-uniform (SGPRs).
#define MKP_QCAP 576   // 63 left over + up to 512 from half a step (32 lanes x 16 bases)
#define MKP_ORD_WORDS 34   // ordinal bitmap of a 1024-base step: 32 words + 2 for the 64-bit window read
template <bool SAMPLE, int NT>
__device__ __forceinline__ void decode_read_fast(const MkpReadHdr* __restrict__ hdrs, uint32_t n_reads, const uint32_t* __restrict__ cigar,
                 const uint8_t* __restrict__ seqs, const MkpTagRef* __restrict__ tagref, const uint32_t* __restrict__ ranks,
                 const uint8_t* __restrict__ ml, const MkpLayout* __restrict__ layouts, const MkpRunParams& prm,
                 MkpEvent* __restrict__ events, MkpReadOut* __restrict__ readout, uint32_t* __restrict__ dev_err,
                 const uint8_t* __restrict__ bedmask, float* __restrict__ sample_vals, uint32_t* __restrict__ lds_layouts,
                 const uint32_t* __restrict__ read_ids, uint32_t* __restrict__ lds_ord, const uint8_t* __restrict__ pdep4,
                     uint32_t* __restrict__ lds_queue) {
  static_assert(NT <= 2, "the call queue holds two ML indices per entry");
  const int lane = lane_id();
  const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t widx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6))) + wib;
  if (widx >= n_reads) return;
  const uint32_t rid = (uint32_t)__builtin_amdgcn_readfirstlane((int)read_ids[widx]);
  const MkpReadHdr h = hdrs[rid];
  MkpReadOut out; out.n_events = 0; out.ok = 0; out.obs[0] = 0; out.obs[1] = 0;
  if ((h.flags & MKP_RF_BAD) || h.n_tags == 0) { if (lane == 0) readout[rid] = out; return; }
  uint32_t* __restrict__ lds_lay = lds_layouts + wib * MKP_LAYOUT_DWORDS;
  uint32_t* __restrict__ ordb = lds_ord + wib * (NT * MKP_ORD_WORDS);   // [NT][MKP_ORD_WORDS] ordinal bitmaps of the step
  // queue, SoA: stored position, ML call index of tag 0 / 1 (~0 = not listed)
  uint32_t* __restrict__ q_pos = lds_queue + wib * ((1 + NT) * MKP_QCAP);
  uint32_t* __restrict__ q_j0 = q_pos + MKP_QCAP;
  uint32_t* __restrict__ q_j1 = q_pos + (NT > 1 ? 2 : 1) * MKP_QCAP;
  { const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(&layouts[h.layout]);
    for (int i = lane; i < MKP_LAYOUT_DWORDS; i += 64) lds_lay[i] = src[i]; }
  __builtin_amdgcn_wave_barrier();
  const MkpLayout* lay = reinterpret_cast<const MkpLayout*>(lds_lay);
  const uint32_t* __restrict__ seqw = reinterpret_cast<const uint32_t*>(seqs + h.seq_off);
  const bool rev = (h.flags & MKP_RF_REVERSE) != 0;
  const uint32_t L = h.l_seq, nd = (L + 7u) >> 3, aln = rev ? 1u : 0u;
  const int n_tags = (int)h.n_tags;
  const int b0 = (int)lay->tags[0].fb & 3, sg0 = (int)lay->tags[0].neg & 1;
  const int xs = rev ? 3 - b0 : b0;                                   // the stored base the tags count
  const uint32_t* gp0 = lds_lay + MKP_LAYOUT_GROUP_DW + (sg0 * 4 + b0) * 32;
  GroupRegs grp0 = load_group(gp0);
  grp0.misc = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp0.misc); grp0.slots = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp0.slots);
  grp0.cids = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp0.cids);
    grp0.member_tags = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp0.member_tags);
#pragma unroll
  for (int kq = 0; kq < MKP_KMAX; kq++) at(grp0.thr,
      kq) = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(at(grp0.thr, kq))));
  grp0.thr_can = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(grp0.thr_can)));
  const uint32_t impl0 = MKP_G_IMPL(grp0.misc);
  const int kcodes0 = (int)((grp0.misc >> 20) & 7u);   // codes of the group: bounds every per-code loop
  uint32_t t_off[NT], t_n[NT], t_ml[NT], t_cur[NT], t_nc[NT], tmu[NT], codes_t[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    t_off[t] = 0; t_n[t] = 0; t_ml[t] = 0; t_cur[t] = 0; t_nc[t] = 0; tmu[t] = 0; codes_t[t] = 0;
    if (t < n_tags) {
      const MkpTagRef tr = tagref[h.tag_off + t]; t_off[t] = tr.rank_off; t_n[t] = tr.n; t_ml[t] = tr.ml_off; t_cur[t] = rev ? tr.n : 0u;
      t_nc[t] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lay->tags[t].n_codes);
      tmu[t] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lay->tagmap[t][b0]);
      for (uint32_t i = 0; i < t_nc[t]; i++) codes_t[t] |= 1u << ((tmu[t] >> (4 + 4 * i)) & 15u);
    }
  }
  // reverse reads need the total up front (forward rank = total - inclusive count in stored order); 4 loads in flight
  uint32_t tot = 0;
  if (rev) {
    uint32_t acc = 0;
    for (uint32_t d0 = 0; d0 < nd; d0 += 256) {
      uint32_t xw[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { const uint32_t dd = d0 + 64u * j + lane; xw[j] = dd < nd ? seqw[dd] : 0u; }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t dd = d0 + 64u * j + lane;
        const int nv = min(max((int)L - (int)(8u * dd), 0), 8);
        acc += (uint32_t)__popc(match8(linearize(xw[j]), xs) & ((1u << nv) - 1u));
      }
    }
    tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(acc), 63);
  }
  bool err = false;
  const bool trimmable = !prm.edge_filter || !(L <= prm.edge_start || L <= prm.edge_end);  // read_can_be_trimmed (mod_bam.rs:1668-1671)
  const bool collapse = prm.numeric_mode == 2;
  uint32_t obs0 = 0, obs1 = 0, contribH = 0, n_ev = 0, cum = 0;
  bool any_surviving = false;
  // CIGAR window: 64 ops in registers, advanced as the batches move along the read
  uint32_t c0 = 0, wq0 = 0, wq1 = 0; int32_t wr0 = h.ref_start;
  uint32_t w_op = 5u, w_qe = 0; int32_t w_dl = 0; uint32_t w_rtot = 0;
  bool win_loaded = false;
  uint32_t qhead = 0, qcount = 0, d0 = 0;
  // the next step's SEQ dwords (two per lane = 16 bases) are always in flight
  uint32_t x_next0 = 2u * (uint32_t)lane < nd ? seqw[2u * lane] : 0u, x_next1 = 2u * (uint32_t)lane + 1u < nd ? seqw[2u * lane + 1u] : 0u;
  // a step's located calls are appended in two halves (lanes 0-31, then 32-63) so the queue never has to take more than
  // 512 entries at once; these hold the step's state between the halves
  bool pend = false;
  uint32_t st_U = 0, st_uincl = 0, st_ucnt = 0, st_HA = 0, st_H = 0, st_cntT = 0, st_d = 0;
  uint32_t st_bm[NT], st_texcl[NT], st_cur[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) { st_bm[t] = 0; st_texcl[t] = 0; st_cur[t] = 0; }

  for (;;) {
    if (err) break;
    if (qcount - qhead < 64u && (pend || d0 < nd)) {
      // ---- producer: one 1024-base step locates its calls; each half of the lanes appends them to the queue
      if (qhead) {  // move the (< 64) unconsumed entries to the front
        const uint32_t n_left = qcount - qhead;
        const bool mv = (uint32_t)lane < n_left;
        const uint32_t a = mv ? q_pos[qhead + lane] : 0u, b = mv ? q_j0[qhead + lane] : 0u, c = (NT > 1 && mv) ? q_j1[qhead + lane] : 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (mv) { q_pos[lane] = a; q_j0[lane] = b; if (NT > 1) q_j1[lane] = c; }
        qcount = n_left; qhead = 0;
      }
      if (!pend) {
        const uint32_t d = d0 + 2u * (uint32_t)lane;          // this lane's first dword: bases [8d, 8d+16)
        const uint32_t xl0 = linearize(x_next0), xl1 = linearize(x_next1);
        { const uint32_t dn = d + 128u; x_next0 = dn < nd ? seqw[dn] : 0u; x_next1 = dn + 1u < nd ? seqw[dn + 1u] : 0u; }
        // the first 64 ranks at every tag's cursor are requested now, ahead of the match/scan work that decides how many are used
        uint32_t e_pre[NT]; bool v_pre[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
          e_pre[t] = rev ? 0u : 0xffffffffu; v_pre[t] = false;
          if (t < n_tags) {
            const uint32_t i = rev ? (t_cur[t] - 64u + lane) : (t_cur[t] + lane);
            v_pre[t] = rev ? ((int32_t)i >= 0 && i < t_cur[t]) : (i < t_n[t]);
            if (v_pre[t]) e_pre[t] = ranks[t_off[t] + i];
          }
        }
        const int nv = min(max((int)L - (int)(8u * d), 0), 16);
        const uint32_t m16 = (match8(xl0, xs) | (match8(xl1, xs) << 8)) & ((1u << nv) - 1u);
        const uint32_t c = (uint32_t)__popc(m16), incl = wave_incl_scan(c);
        const uint32_t cntT = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t wlo = rev ? (tot - cum - cntT) : cum, whi = wlo + cntT;   // rank window of this step
#pragma unroll
        for (int t = 0; t < NT; t++) { st_cur[t] = t_cur[t]; if (t < n_tags && lane < MKP_ORD_WORDS) ordb[t * MKP_ORD_WORDS + lane] = 0; }
#pragma unroll
        for (int t = 0; t < NT; t++) {
          if (t >= n_tags) break;
          for (bool first = true;; first = false) {
            uint32_t e; bool hit, valid; uint32_t nh;
            if (first) { e = e_pre[t]; valid = v_pre[t]; }
            else if (!rev) { const uint32_t i = t_cur[t] + lane; valid = i < t_n[t]; e = valid ? ranks[t_off[t] + i] : 0xffffffffu; }
            else { const uint32_t i = t_cur[t] - 64u + lane; valid = (int32_t)i >= 0 && i < t_cur[t]; e = valid ? ranks[t_off[t] + i] : 0u; }
            if (!rev) { hit = valid && e < whi; nh = (uint32_t)__popcll(__ballot(hit)); t_cur[t] += nh; }
            // an entry >= whi is past the last occurrence: never consumed -> error at the end
            else { hit = valid && e >= wlo && e < whi; nh = (uint32_t)__popcll(__ballot(hit)); t_cur[t] -= nh; }
            if (nh == 0) break;
            const uint32_t ib = (rev ? (tot - 1u - e) : e) - cum;   // step-relative stored ordinal of the called base (< 1024)
            if (hit) atomicOr(&ordb[t * MKP_ORD_WORDS + (ib >> 5)], 1u << (ib & 31u));
            if (nh < 64) break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t U = impl0 ? m16 : 0u, uincl = 0, ucnt = 0;
        const uint32_t ex = incl - c;
        const uint32_t n0 = m16 & 15u, n1 = (m16 >> 4) & 15u, n2 = (m16 >> 8) & 15u, n3 = m16 >> 12;
        const uint32_t c0n = (uint32_t)__popc(n0), c1n = c0n + (uint32_t)__popc(n1), c2n = c1n + (uint32_t)__popc(n2);
#pragma unroll
        for (int t = 0; t < NT; t++) {
          st_bm[t] = 0; st_texcl[t] = 0;
          if (t < n_tags) {  // bits [excl, excl+cnt) of the ordinal bitmap deposited onto the set bits of the lane's 16-base match mask
            const uint32_t w0 = ordb[t * MKP_ORD_WORDS + (ex >> 5)], w1 = ordb[t * MKP_ORD_WORDS + (ex >> 5) + 1];
            const uint32_t f = __builtin_amdgcn_alignbit(w1, w0, ex & 31u) & ((1u << c) - 1u);
            st_bm[t] = (uint32_t)pdep4[(n0 << 4) | (f & 15u)] | ((uint32_t)pdep4[(n1 << 4) | ((f >> c0n) & 15u)] << 4) |
                       ((uint32_t)pdep4[(n2 << 4) | ((f >> c1n) & 15u)] << 8) | ((uint32_t)pdep4[(n3 << 4) | ((f >> c2n) & 15u)] << 12);
            U |= st_bm[t];
            const uint32_t c2 = (uint32_t)__popc(st_bm[t]);
            uincl = wave_incl_scan(c2); ucnt = c2; st_texcl[t] = uincl - c2;
          }
        }
        if (NT > 1 || impl0) { ucnt = (uint32_t)__popc(U); uincl = wave_incl_scan(ucnt); }   // else U == bm[0]: its scan is already there
        st_U = U; st_uincl = uincl; st_ucnt = ucnt; st_d = d; st_cntT = cntT;
        st_H = (uint32_t)__builtin_amdgcn_readlane((int)uincl, 63); st_HA = (uint32_t)__builtin_amdgcn_readlane((int)uincl, 31);
      }
      {  // append this half's calls: lanes 0-31 first, lanes 32-63 on the next visit
        const bool mine = pend ? (lane >= 32) : (lane < 32);
        uint32_t ut = mine ? st_U : 0u, sidx = qcount + st_uincl - st_ucnt - (pend ? st_HA : 0u);
        while (ut) {
          const uint32_t bit = (uint32_t)__ffs((int)ut) - 1u, below = (1u << bit) - 1u;
          q_pos[sidx] = 8u * st_d + bit;
#pragma unroll
          for (int t = 0; t < NT; t++) {
            const uint32_t idx = st_texcl[t] + (uint32_t)__popc(st_bm[t] & below);
            const uint32_t jx = rev ? (st_cur[t] - 1u - idx) : (st_cur[t] + idx);
            (t == 0 ? q_j0 : q_j1)[sidx] = ((st_bm[t] >> bit) & 1u) ? jx : 0xffffffffu;
          }
          sidx++; ut &= ut - 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        qcount += pend ? (st_H - st_HA) : st_HA;
        if (!pend && st_H > st_HA) pend = true;                  // the upper lanes still hold calls
        else { pend = false; cum += st_cntT; d0 += 128; }        // step done
      }
      continue;
    }
    if (qcount == qhead) break;
    if (prm.debug_skip & 16u) { qhead = qcount; continue; }
    // ---- consumer: up to 64 queued calls, in read order
    const uint32_t nb = min(64u, qcount - qhead);
    const bool active = (uint32_t)lane < nb;
    const uint32_t q = active ? q_pos[qhead + lane] : 0u;
    const uint32_t f = rev ? (L - 1 - q) : q;  // forward (as-sequenced) position
    // the ML bytes are requested first (up to MKP_KMAX per tag) and converted after the CIGAR mapping, whose latency they overlap
    uint32_t mlq[NT][MKP_KMAX]; bool found_t[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      found_t[t] = false;
#pragma unroll
      for (int i = 0; i < MKP_KMAX; i++) mlq[t][i] = 0;
      if (t < n_tags) {
        const uint32_t jx = active ? (t == 0 ? q_j0 : q_j1)[qhead + lane] : 0xffffffffu;
        found_t[t] = jx != 0xffffffffu;
        const uint32_t nc = t_nc[t], base = found_t[t] ? (t_ml[t] + jx * nc) : 0u;
#pragma unroll
        for (int i = 0; i < MKP_KMAX; i++) if ((uint32_t)i < nc) mlq[t][i] = ml[base + (found_t[t] ? (uint32_t)i : 0u)];
      }
    }
    // reference position through the CIGAR window (aligned pairs: M/=/X only, util.rs:122-145)
    bool mapped = false; int32_t rpos = 0;
    {
      bool pending = active;
      for (;;) {
        if (!win_loaded || (__any(pending && q >= wq1) && !__any(pending && q < wq1))) {
          if (win_loaded) { c0 += 64; wq0 = wq1; wr0 += (int32_t)w_rtot; }
          if (c0 >= h.n_cigar) break;
          const uint32_t w = (c0 + lane < h.n_cigar) ? cigar[h.cigar_off + c0 + lane] : 5u /*0H*/;
          w_op = w & 15u; const uint32_t len = w >> 4;
          const uint32_t qlen = op_consumes_query(w_op) ? len : 0u, rlen = op_consumes_ref(w_op) ? len : 0u;
          w_qe = wave_incl_scan(qlen); const uint32_t re = wave_incl_scan(rlen);
          w_dl = (wr0 + (int32_t)(re - rlen)) - (int32_t)(wq0 + w_qe - qlen);   // ref start - query start of the op
          wq1 = wq0 + (uint32_t)__builtin_amdgcn_readlane((int)w_qe, 63); w_rtot = (uint32_t)__builtin_amdgcn_readlane((int)re, 63);
          win_loaded = true;
          continue;
        }
        const bool ready = pending && q < wq1;
        const int oi = find_op(w_qe, ready ? q - wq0 : 0u) & 63;
        const uint32_t my_op = __shfl(w_op, oi, 64);
        const int32_t my_dl = __shfl(w_dl, oi, 64);
        if (ready) { mapped = op_is_match(my_op); rpos = (int32_t)q + my_dl; pending = false; }
        if (!__any(pending)) break;
      }
    }
    F4 pk = {0.f, 0.f, 0.f, 0.f};
    uint32_t SH = 0, setmask = 0;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      if (t >= n_tags) break;
      const bool found = found_t[t];
#pragma unroll
      for (int i = 0; i < MKP_KMAX; i++) {
        if ((uint32_t)i >= t_nc[t]) break;
        const float p = ((float)mlq[t][i] + 0.5f) / 256.0f;   // quals_to_probs (mod_bam.rs:808-816)
        const uint32_t kk = (tmu[t] >> (4 + 4 * i)) & 15u;   // wave-uniform local code
        setk(pk, kk, found, p);
      }
      SH |= found ? (1u << (tmu[t] & 15u)) : 0u;
      setmask |= found ? codes_t[t] : 0u;
    }
    if (NT > 1 && __popc(SH) >= 2) {  // combine_checked's sum test, once on the final map (partial sums of positive terms cannot exceed it)
      float s = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < MKP_KMAX; k2++) if (setmask & (1u << k2)) s = s + at(pk, k2);
      if (s > 1.01f) err = true;
    }
    uint32_t ev_info = 0; float sv = 0.f; bool has_ev = false;
    if (active && (SH | impl0)) {
      const bool edge_keep = !prm.edge_filter ||
          (prm.edge_inverted ? (f < prm.edge_start || f >= L - prm.edge_end) : (f >= prm.edge_start && f < L - prm.edge_end));
      int pat; uint32_t member_contrib;
      if (SH) { if (impl0 & ~SH) err = true; pat = (int)SH; member_contrib = SH; }      // ExplicitConflictInferred
      else { pat = MKP_PAT_INFERRED; member_contrib = impl0; }                          // implicit fill (mod_bam.rs:1265-1292)
      const uint32_t pv = gp0[12 + pat];
      contribH |= member_contrib;
      if (trimmable && edge_keep) {
        if (SAMPLE) {  // SeqPosBaseModProbs::filter_positions (read_ids_to_base_mod_probs.rs:966-1070)
          bool keep = !prm.only_mapped || mapped;
          if (prm.has_focus) keep = keep && mapped && rpos >= prm.win_start && rpos < prm.win_end
              && ((bedmask[rpos - prm.win_start] >> (aln ^ (uint32_t)sg0)) & 1u);
          // `extract calls`: the collapse left no code in the map: no profile row
          if (keep && prm.sample_mode == 3 && ((pv >> 3) & 7u) == 0u) any_surviving = true;
          else if (keep && prm.sample_mode >= 2) {
            any_surviving = true; uint32_t ob = 0; float am = 0.f; ev_info = summary_info(grp0, pv, pk, collapse, &ob, kcodes0, &am); obs0 |= ob;
              has_ev = true;
            // `extract calls`: the event carries the forward position
            if (prm.sample_mode == 3) { ev_info |= ((uint32_t)sg0 << 2) | ((pat == MKP_PAT_INFERRED ? 1u : 0u) << 3); sv = am; rpos = (int32_t)f; }
          }
          else if (keep) { any_surviving = true; sv = argmax_group(grp0, pv, pk, collapse, kcodes0); ev_info = MKP_G_TB(grp0.misc); has_ev = true; }
        } else {
          any_surviving = true;
          uint32_t ob = 0;
          const int cls = call_group(grp0, pv, pk, collapse, &ob, kcodes0);
          const uint32_t tally = aln ^ (uint32_t)sg0;  // read_cache.rs:181-188 / FeatureVector::add_feature
          if (tally) obs1 |= ob; else obs0 |= ob;
          if (mapped) {
            const uint32_t cid = cls == 0 ? (uint32_t)MKP_C_FAIL : cls == 1 ? MKP_G_CIDCAN(grp0.misc) : ((grp0.cids >> (8 * (cls - 2))) & 0xffu);
            ev_info = cid | (tally << 8) | ((uint32_t)b0 << 9) | (aln << 11) | (1u << 12);
            has_ev = true;
          }
        }
      }
    }
    // ballot-compacted, position-ordered append of this batch's events
    const unsigned long long b1 = __ballot(has_ev);
    const uint32_t step_total = (uint32_t)__popcll(b1);
    if (step_total) {
      const uint32_t off = n_ev + (uint32_t)__popcll(b1 & lanemask_lt());
      if (n_ev + step_total > h.event_cap) { err = true; if (lane == 0) atomicOr(dev_err, ERR_EVENT_CAP); }
      else if (has_ev) {
        MkpEvent ev; ev.pos = (uint32_t)rpos; ev.info = ev_info; events[h.event_off + off] = ev;
        if (SAMPLE) sample_vals[h.event_off + off] = sv;
      }
      n_ev += step_total;
    }
    err = __any(err);
    qhead += nb;
  }
  // a delta list must not run past the last occurrence of its base: every entry must have been consumed (mod_bam.rs:705-727)
#pragma unroll
  for (int t = 0; t < NT; t++) if (t < n_tags) { if (rev ? (t_cur[t] != 0u) : (t_cur[t] != t_n[t])) err = true; }
  err = __any(err);
  obs0 = wave_or(obs0); obs1 = wave_or(obs1);
  any_surviving = __any(any_surviving);
  // InvalidImplicitMode: the group's contributing tags all lack a mode character (read_cache.rs:122-137)
  if (!prm.force_allow && !SAMPLE) {
    const uint32_t mc = wave_or(contribH); uint32_t tagbits = 0;
#pragma unroll
    for (int mi = 0; mi < MKP_MAX_MEMBERS; mi++) if (mc & (1u << mi)) tagbits |= 1u << ((grp0.member_tags >> (4 * mi)) & 15u);
    if (tagbits && (tagbits & ~(uint32_t)lay->default_mask) == 0) err = true;
  }
  if (lane == 0) {
    if (!err && any_surviving) { out.ok = 1; out.n_events = n_ev; out.obs[0] = obs0; out.obs[1] = obs1; }
    readout[rid] = out;
  }
}

// ----------------------------------------------------------------------------------------------
// Decode, SPARSE layouts: FAST layouts whose tags are all explicit ('?': only the listed bases are calls) and, when there are
// two tags, list the same bases (`C+h?,..;C+m?,..` with one delta list — checked by the host).  The calls are then a few
// percent of the bases, and locating them must not cost per base.  Per 4096-base step (64 bases = 8 SEQ dwords per lane) the
// wave only counts: per dword a nibble-equality flag word (7 VALU for 8 bases) and its popcount, byte-packed running counts
// inside the lane, one wave prefix sum; the flag words go to LDS.  The ranks that fall into the step's window are taken 64 at a
// time from the sorted rank list; every lane owning one finds the lane whose bases contain that occurrence (6 ds_bpermute steps
// over the prefix sums), the dword inside that lane (one SWAR compare over the packed counts) and the base inside the dword
// (select on the flag word read back from LDS) — {stored position, call index} goes straight into the call queue, in read
// order.  No per-base marks, no bitmap deposit, no per-bit loops.  The consumer (per-call work on full batches of 64) is the
// one of the FAST kernels.
// flag word of a SEQ dword: bit 4i set where nibble i equals the BAM code of base k (A,C,G,T = 1,2,4,8)
__device__ __forceinline__ uint32_t nibble_eq(uint32_t x, uint32_t pat) {
  uint32_t t = x ^ pat;
  t |= t >> 1; t |= t >> 2;
  return ~t & 0x11111111u;
}
__device__ __forceinline__ uint32_t rfl_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
#define MKP_SQCAP 128   // call queue of the SPARSE kernels: < 64 left over + one round of <= 64
template <bool SAMPLE, int NT>
__device__ __forceinline__ void decode_read_sparse(const MkpReadHdr* __restrict__ hdrs, uint32_t n_reads, const uint32_t* __restrict__ cigar,
                 const uint8_t* __restrict__ seqs, const MkpTagRef* __restrict__ tagref, const uint32_t* __restrict__ ranks,
                 const uint8_t* __restrict__ ml, const MkpLayout* __restrict__ layouts, const MkpRunParams& prm,
                 MkpEvent* __restrict__ events, MkpReadOut* __restrict__ readout, uint32_t* __restrict__ dev_err,
                 const uint8_t* __restrict__ bedmask, float* __restrict__ sample_vals, uint32_t* __restrict__ lds_layouts,
                 const uint32_t* __restrict__ read_ids, uint32_t* __restrict__ lds_queue, uint32_t* __restrict__ lds_flags) {
  static_assert(NT <= 2, "one or two tags");
  const int lane = lane_id();
  const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t widx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6))) + wib;
  if (widx >= n_reads) return;
  // Duplex reads (layout.fast == 2: two (strand, base) groups on different bases) are listed twice, once per group (bit 31 = the
  // second); each listing decodes its group's tags exactly like a single-group read, into its own half of the read's event slice
  // behind the room for the merged list, with its own summary; mkp_merge_duplex interleaves the two by position afterwards.
  const uint32_t rid_raw = (uint32_t)__builtin_amdgcn_readfirstlane((int)read_ids[widx]);
  const uint32_t rid = rid_raw & 0x7fffffffu; const bool half_b = (rid_raw >> 31) != 0u;
  const MkpReadHdr h = hdrs[rid];
  MkpReadOut out; out.n_events = 0; out.ok = 0; out.obs[0] = 0; out.obs[1] = 0;
  if ((h.flags & MKP_RF_BAD) || h.n_tags == 0) { if (lane == 0) readout[rid] = out; return; }
  uint32_t* __restrict__ lds_lay = lds_layouts + wib * MKP_LAYOUT_DWORDS;
  uint32_t* __restrict__ q_pos = lds_queue + wib * (2 * MKP_SQCAP);   // queue, SoA: stored position, call index (the same for both tags)
  uint32_t* __restrict__ q_j = q_pos + MKP_SQCAP;
  uint32_t* __restrict__ flg = lds_flags + wib * 512u;                 // the step's 512 flag words
  { const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(&layouts[h.layout]);
    for (int i = lane; i < MKP_LAYOUT_DWORDS; i += 64) lds_lay[i] = src[i]; }
  __builtin_amdgcn_wave_barrier();
  const MkpLayout* lay = reinterpret_cast<const MkpLayout*>(lds_lay);
  const uint32_t* __restrict__ seqw = reinterpret_cast<const uint32_t*>(seqs + h.seq_off);
  const bool rev = (h.flags & MKP_RF_REVERSE) != 0;
  const uint32_t L = h.l_seq, nd = (L + 7u) >> 3, aln = rev ? 1u : 0u;
  const bool duplex = (uint32_t)__builtin_amdgcn_readfirstlane((int)lay->fast) == 2u;
  const int n_first = (int)__builtin_amdgcn_readfirstlane((int)lay->pad);
  const int tb = (duplex && half_b) ? n_first : 0;                     // first tag of the group this wave decodes
  const int n_tags = duplex ? (half_b ? (int)h.n_tags - n_first : n_first) : (int)h.n_tags;
  uint32_t ev_base = h.event_off, ev_cap = h.event_cap, ro_idx = rid;
  if (duplex) {
    const uint32_t nA = tagref[h.tag_off].n, nB = tagref[h.tag_off + n_first].n;   // calls listed per group (one shared rank list each)
    ev_base = h.event_off + nA + nB + (half_b ? nA : 0u); ev_cap = half_b ? nB : nA;
    if (half_b) ro_idx = prm.readout_b_off + rid;
  }
  const int b0 = (int)lay->tags[tb].fb & 3, sg0 = (int)lay->tags[tb].neg & 1;
  const int xs = rev ? 3 - b0 : b0;                                   // the stored base the tags count
  const uint32_t* gp0 = lds_lay + MKP_LAYOUT_GROUP_DW + (sg0 * 4 + b0) * 32;
  GroupRegs grp0 = load_group(gp0);
  grp0.misc = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp0.misc); grp0.slots = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp0.slots);
  grp0.cids = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp0.cids);
    grp0.member_tags = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp0.member_tags);
#pragma unroll
  for (int kq = 0; kq < MKP_KMAX; kq++) at(grp0.thr,
      kq) = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(at(grp0.thr, kq))));
  grp0.thr_can = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(grp0.thr_can)));
  const int kcodes0 = (int)((grp0.misc >> 20) & 7u);   // codes of the group: bounds every per-code loop
  uint32_t t_ml[NT], t_nc[NT], tmu[NT], codes_t[NT];
  uint32_t t_off = 0, t_n = 0, t_cur = 0;   // the (shared) rank list
#pragma unroll
  for (int t = 0; t < NT; t++) {
    t_ml[t] = 0; t_nc[t] = 0; tmu[t] = 0; codes_t[t] = 0;
    if (t < n_tags) {
      const MkpTagRef tr = tagref[h.tag_off + tb + t]; t_ml[t] = tr.ml_off;
      if (t == 0) { t_off = tr.rank_off; t_n = tr.n; t_cur = rev ? tr.n : 0u; }
      t_nc[t] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lay->tags[tb + t].n_codes);
      tmu[t] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lay->tagmap[tb + t][b0]);
      for (uint32_t i = 0; i < t_nc[t]; i++) codes_t[t] |= 1u << ((tmu[t] >> (4 + 4 * i)) & 15u);
    }
  }
  uint32_t SH_all = 0, setmask_all = 0;   // every call is listed by every tag: hit pattern and code set are wave constants
#pragma unroll
  for (int t = 0; t < NT; t++) if (t < n_tags) { SH_all |= 1u << (tmu[t] & 15u); setmask_all |= codes_t[t]; }
  const uint32_t pat = 0x11111111u << xs;
  // the low nibble of the last byte is not a base when L is odd: its flag is cleared wherever that dword is looked at
  const uint32_t odd_dw = (L & 1u) ? ((L - 1u) >> 3) : 0xffffffffu, odd_clear = ~(1u << (8u * (((L - 1u) >> 1) & 3u)));
  // reverse reads need the total up front (forward rank = total - inclusive count in stored order); 4 loads in flight.
  // (SEQ is zero-padded to a dword per read and code 0 matches no base; dwords past the read are not loaded.)
  uint32_t tot = 0;
  if (rev) {
    uint32_t acc = 0;
    for (uint32_t d0 = 0; d0 < nd; d0 += 256) {
      uint32_t xw[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { const uint32_t dd = d0 + 64u * j + lane; xw[j] = dd < nd ? seqw[dd] : 0u; }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t dd = d0 + 64u * j + lane;
        uint32_t F = nibble_eq(xw[j], pat); if (dd == odd_dw) F &= odd_clear;
        acc += (uint32_t)__popc(F);
      }
    }
    tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(acc), 63);
  }
  bool err = false;
  // reverse reads consume the (ascending) rank list from its end: a last entry past the last occurrence of the base
  // (mod_bam.rs:705-727) would never be consumed — the read is rejected here so that hits always form a suffix of the cursor window
  if (rev && t_n && ranks[t_off + t_n - 1u] >= tot) err = true;
  const bool trimmable = !prm.edge_filter || !(L <= prm.edge_start || L <= prm.edge_end);  // read_can_be_trimmed (mod_bam.rs:1668-1671)
  const bool collapse = prm.numeric_mode == 2;
  uint32_t obs0 = 0, obs1 = 0, contribH = 0, n_ev = 0, cum = 0;
  bool any_surviving = false;
  // CIGAR window: 64 ops in registers, advanced as the batches move along the read
  // (128 ops per window, two per lane: w_qe = inclusive query end of the lane's pair, w_mid = where its second op starts,
  //  w_a / w_b = (reference start - query start) << 1 | is-match of the two ops)
  // The window's running values (first op, query / reference offsets, totals) are kept in scalar registers — read back through readfirstlane
  // after every update; carried as vectors they were advanced with v_cndmask and copied at every loop head — and "nothing loaded yet" is an
  // empty window in front of op 0: no flag, no conditional advance (the same rewrite as refwin_s_* of mkp_slots.hip).
  uint32_t c0 = 0u - 128u, wq0 = 0, wq1 = 0; int32_t wr0 = h.ref_start;
  uint32_t w_qe = 0, w_mid = 0, w_a = 0, w_b = 0, w_rtot = 0;
  auto cigar2 = [&](uint32_t c) { uint2 r; const uint32_t k = c + 2u * (uint32_t)lane; r.x = k < h.n_cigar ? cigar[h.cigar_off + k] : 5u /*0H*/;
    r.y = k + 1u < h.n_cigar ? cigar[h.cigar_off + k + 1u] : 5u; return r; };
  uint2 w_pref = cigar2(0);   // the first CIGAR window, requested before the read is walked
  uint32_t qhead = 0, qcount = 0, d0 = 0;
  // the next step's SEQ dwords (eight per lane = 64 bases, two 16-byte loads) are always in flight; SEQ buffers end with
  // slack, so whole vectors are loaded and the dwords past the read are discarded when the flags are made
  // reads start 4-byte aligned: vector loads may be unaligned (fine on global memory)
  const uint4* __restrict__ seqv = reinterpret_cast<const uint4*>(seqw);
  uint4 xa = make_uint4(0, 0, 0, 0), xb = make_uint4(0, 0, 0, 0);
  auto load_step = [&](uint32_t dstep) {
    const uint32_t d = dstep + 8u * (uint32_t)lane;
    if (d + 4u <= nd) xa = *reinterpret_cast<const uint4*>(seqw + d);
    else { xa.x = d < nd ? seqw[d] : 0u; xa.y = d + 1u < nd ? seqw[d + 1u] : 0u; xa.z = d + 2u < nd ? seqw[d + 2u] : 0u; xa.w = 0u; }
    if (d + 8u <= nd) xb = *reinterpret_cast<const uint4*>(seqw + d + 4u);
    else { xb.x = d + 4u < nd ? seqw[d + 4u] : 0u; xb.y = d + 5u < nd ? seqw[d + 5u] : 0u; xb.z = d + 6u < nd ? seqw[d + 6u] : 0u; xb.w = 0u; }
  };
  (void)seqv;
  load_step(0);
  // the current step: per-lane counts (byte-packed running counts of its 8 dwords), wave prefix sums, the step's rank window
  bool step_loaded = false;
  uint32_t st_incl = 0, st_excl = 0, st_cumlo = 0, st_cumhi = 0, st_cntT = 0, st_wlo = 0, st_whi = 0;

  for (;;) {
    if (err) break;
    // ---- producer: a loop of its own — the steps and rank batches that refill the queue touch none of the consumer's state, and as
    // iterations of the one outer loop each of them went through that loop's head with all of it
    while (qcount - qhead < 64u && (step_loaded || d0 < nd)) {
      // the next 64 ranks at the cursor are requested first: the flag / scan work below hides the load
      uint32_t e; bool valid;
      if (!rev) { const uint32_t i = t_cur + lane; valid = i < t_n; e = valid ? ranks[t_off + i] : 0xffffffffu; }
      else { const uint32_t i = t_cur - 64u + lane; valid = (int32_t)i >= 0 && i < t_cur; e = valid ? ranks[t_off + i] : 0u; }
      if (!step_loaded) {
        const uint32_t d = d0 + 8u * (uint32_t)lane;          // this lane's first dword: bases [8d, 8d+64)
        uint32_t F[8] = {nibble_eq(xa.x, pat), nibble_eq(xa.y, pat), nibble_eq(xa.z, pat), nibble_eq(xa.w, pat),
                         nibble_eq(xb.x, pat), nibble_eq(xb.y, pat), nibble_eq(xb.z, pat), nibble_eq(xb.w, pat)};
        if (odd_dw - d < 8u) {   // the read's last dword sits in this lane and L is odd
#pragma unroll
          for (int j = 0; j < 8; j++) if (d + (uint32_t)j == odd_dw) F[j] &= odd_clear;
        }
        load_step(d0 + 512u);
        __builtin_amdgcn_wave_barrier();   // the previous step's readers of the flag words are done (same wave: program order)
        *reinterpret_cast<uint4*>(flg + 8u * (uint32_t)lane) = make_uint4(F[0], F[1], F[2], F[3]);
        *reinterpret_cast<uint4*>(flg + 8u * (uint32_t)lane + 4u) = make_uint4(F[4], F[5], F[6], F[7]);
        // byte-packed counts, then running counts inside the lane (byte k = count of the dwords before k)
        const uint32_t clo = (uint32_t)__popc(F[0]) | ((uint32_t)__popc(F[1]) << 8) | ((uint32_t)__popc(F[2]) << 16) | ((uint32_t)__popc(F[3]) << 24);
        const uint32_t chi = (uint32_t)__popc(F[4]) | ((uint32_t)__popc(F[5]) << 8) | ((uint32_t)__popc(F[6]) << 16) | ((uint32_t)__popc(F[7]) << 24);
        const uint32_t plo = clo + (clo << 8) + (clo << 16) + (clo << 24);   // inclusive prefix per byte (sums stay below 256)
        const uint32_t tlo = plo >> 24;                                     // count of dwords 0..3
        const uint32_t phi = chi + (chi << 8) + (chi << 16) + (chi << 24) + tlo * 0x01010101u;
        st_cumlo = plo << 8; st_cumhi = (phi << 8) | tlo;                    // exclusive
        const uint32_t c = phi >> 24;                                       // the lane's 64 bases
        st_incl = wave_incl_scan(c); st_excl = st_incl - c;
        st_cntT = (uint32_t)__builtin_amdgcn_readlane((int)st_incl, 63);
        st_wlo = rev ? (tot - cum - st_cntT) : cum; st_whi = st_wlo + st_cntT;   // rank window of this step
        step_loaded = true;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      // (reverse reads: the list's last entry was checked against the total, so every entry is below the first window's end)
      const bool hit = valid && (rev ? (e >= st_wlo) : (e < st_whi));
      const unsigned long long hb = __ballot(hit);
      const uint32_t nh = (uint32_t)__popcll(hb);
      if (nh) {
        if (qhead) {  // move the (< 64) unconsumed entries to the front
          const uint32_t n_left = qcount - qhead;
          const bool mv = (uint32_t)lane < n_left;
          const uint32_t a = mv ? q_pos[qhead + lane] : 0u, b = mv ? q_j[qhead + lane] : 0u;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (mv) { q_pos[lane] = a; q_j[lane] = b; }
          qcount = n_left; qhead = 0;
        }
        const uint32_t ib = hit ? ((rev ? (tot - 1u - e) : e) - cum) : 0u;   // step-relative stored ordinal of the called base (< cntT)
        const int owner = find_op(st_incl, ib) & 63;                          // the lane whose 64 bases hold that occurrence
        const uint32_t o_excl = (uint32_t)__shfl((int)st_excl, owner, 64), o_lo = (uint32_t)__shfl((int)st_cumlo, owner, 64),
            o_hi = (uint32_t)__shfl((int)st_cumhi, owner, 64);
        const uint32_t k = ib - o_excl;                                       // occurrence inside the owner's 64 bases
        // dword: the last of the 8 running counts that is <= k (SWAR: byte i of t keeps 0x80 where count_i > k; counts < 128)
        const uint32_t kk1 = (k + 1u) * 0x01010101u;
        const uint32_t t_lo = ((o_lo | 0x80808080u) - kk1) & 0x80808080u, t_hi = ((o_hi | 0x80808080u) - kk1) & 0x80808080u;
        const uint32_t jd = 7u - (uint32_t)__popc(t_lo) - (uint32_t)__popc(t_hi);
        const uint32_t cj = ((jd < 4u ? o_lo : o_hi) >> (8u * (jd & 3u))) & 0xffu;
        uint32_t r = k - cj;
        uint32_t Fw = hit ? flg[8u * (uint32_t)owner + jd] : 1u;
        Fw = ((Fw & 0x01010101u) << 4) | ((Fw >> 4) & 0x01010101u);         // base order: bit 4b = base b of the dword
        uint32_t bpos = 0, cc;
        cc = (uint32_t)__popc(Fw & 0xffffu); if (r >= cc) { r -= cc; bpos += 4u; Fw >>= 16; }
        cc = (uint32_t)__popc(Fw & 0xffu);   if (r >= cc) { r -= cc; bpos += 2u; Fw >>= 8; }
        cc = Fw & 1u;                         if (r >= cc) { bpos += 1u; }
        // read order: forward reads hit on the low lanes with ascending positions, reverse reads on the high lanes with descending ones
        const uint32_t slot = qcount + (uint32_t)__popcll(rev ? (hb & ~lanemask_le()) : (hb & lanemask_lt()));
        if (hit) { q_pos[slot] = 8u * (d0 + 8u * (uint32_t)owner + jd) + bpos;
          q_j[slot] = rev ? (t_cur - 64u + (uint32_t)lane) : (t_cur + (uint32_t)lane); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        qcount += nh;
        if (rev) t_cur -= nh; else t_cur += nh;
      }
      if (nh < 64u) {   // the window holds no further rank: step done; nothing left in the list: the rest of the read holds no call
        cum += st_cntT; d0 += 512; step_loaded = false;
        if (rev ? (t_cur == 0u) : (t_cur == t_n)) d0 = nd;
      }
    }
    if (qcount == qhead) break;
    // ---- consumer: up to 64 queued calls, in read order
    const uint32_t nb = min(64u, qcount - qhead);
#ifdef MKP_DEBUG
    if (prm.debug_skip & 16u) { qhead += nb; any_surviving = true; continue; }   // ablation: producer only
#endif
    const bool active = (uint32_t)lane < nb;
    // The descriptor words the per-call code branches on are made opaque here, once per batch: left visible as loop invariants, every
    // uniform compare derived from them (which code a slot holds, how many codes a tag lists, ...) was hoisted out of the loop as a 64-bit
    // lane mask, dozens of them, spilled to VGPR lanes and read back with two v_readlane each inside the loop — VALU instructions in a
    // VALU-bound kernel; derived in place they are a few scalar instructions that live for a handful of cycles.
    GroupRegs g0 = grp0; uint32_t kc_l = (uint32_t)kcodes0, SH_l = SH_all, sm_l = setmask_all, tmu_l[NT], tnc_l[NT];
    asm volatile("" : "+s"(g0.misc), "+s"(g0.slots), "+s"(g0.cids), "+s"(kc_l), "+s"(SH_l), "+s"(sm_l));
#pragma unroll
    for (int t = 0; t < NT; t++) { tmu_l[t] = tmu[t]; tnc_l[t] = t_nc[t]; asm volatile("" : "+s"(tmu_l[t]), "+s"(tnc_l[t])); }
    const uint32_t q = active ? q_pos[qhead + lane] : 0u;
    const uint32_t jx = active ? q_j[qhead + lane] : 0u;
    const uint32_t f = rev ? (L - 1 - q) : q;  // forward (as-sequenced) position
    // the ML bytes are requested first (up to MKP_KMAX per tag) and converted after the CIGAR mapping, whose latency they overlap
    uint32_t mlq[NT][MKP_KMAX];
#pragma unroll
    for (int t = 0; t < NT; t++) {
#pragma unroll
      for (int i = 0; i < MKP_KMAX; i++) mlq[t][i] = 0;
      if (t < n_tags) {
        const uint32_t nc = tnc_l[t], base = active ? (t_ml[t] + jx * nc) : 0u;
#pragma unroll
        for (int i = 0; i < MKP_KMAX; i++) if ((uint32_t)i < nc) mlq[t][i] = ml[base + (active ? (uint32_t)i : 0u)];
      }
    }
    // reference position through the CIGAR window (aligned pairs: M/=/X only, util.rs:122-145)
    bool mapped = false; int32_t rpos = 0;
#ifdef MKP_DEBUG
    if (prm.debug_skip & 32u) { mapped = active; rpos = h.ref_start + (int32_t)q; } else   // ablation: no CIGAR mapping
#endif
    {
      bool pending = active;
      // the next 128 ops; false behind the last op (cannot happen for positions inside SEQ: the packer checks the lengths)
      auto win_next = [&]() -> bool {
        c0 = rfl_u(c0 + 128u); wq0 = wq1; wr0 = (int32_t)rfl_u((uint32_t)wr0 + w_rtot);
        if (c0 >= h.n_cigar) { w_rtot = 0; return false; }
        const uint2 w = w_pref;   // requested one window ahead: the load is off the mapping's dependency chain
        w_pref = cigar2(c0 + 128u);
        const uint32_t op0 = w.x & 15u, len0 = w.x >> 4, op1 = w.y & 15u, len1 = w.y >> 4;
        const uint32_t ql0 = op_consumes_query(op0) ? len0 : 0u, rl0 = op_consumes_ref(op0) ? len0 : 0u;
        const uint32_t ql1 = op_consumes_query(op1) ? len1 : 0u, rl1 = op_consumes_ref(op1) ? len1 : 0u;
        w_qe = wave_incl_scan(ql0 + ql1); const uint32_t re = wave_incl_scan(rl0 + rl1);
        w_mid = w_qe - ql1;                                                       // window-relative query offset of the second op
        const int32_t dl0 = (wr0 + (int32_t)(re - rl0 - rl1)) - (int32_t)(wq0 + w_qe - ql0 - ql1);   // ref start - query start of the first op
        const int32_t dl1 = (wr0 + (int32_t)(re - rl1)) - (int32_t)(wq0 + w_mid);
        w_a = ((uint32_t)dl0 << 1) | (op_is_match(op0) ? 1u : 0u); w_b = ((uint32_t)dl1 << 1) | (op_is_match(op1) ? 1u : 0u);
        wq1 = wq0 + (uint32_t)__builtin_amdgcn_readlane((int)w_qe, 63); w_rtot = (uint32_t)__builtin_amdgcn_readlane((int)re, 63);
        return true;
      };
      auto win_pass = [&]() {   // the lanes whose position lies in the loaded window (positions ascend from batch to batch: none lies before it)
        const bool ready = pending && q < wq1;
        if (!__any(ready)) return;
        const uint32_t qrel = ready ? q - wq0 : 0u;
        const int oi = find_op(w_qe, qrel) & 63;
        const uint32_t o_mid = (uint32_t)__shfl((int)w_mid, oi, 64), o_a = (uint32_t)__shfl((int)w_a, oi, 64),
            o_b = (uint32_t)__shfl((int)w_b, oi, 64);
        const uint32_t pick = qrel < o_mid ? o_a : o_b;
        if (ready) { mapped = (pick & 1u) != 0u; rpos = (int32_t)q + ((int32_t)pick >> 1); pending = false; }
      };
      win_pass();
      while (__any(pending)) { if (!win_next()) break; win_pass(); }
    }
    F4 pk = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; t++) {
      if (t >= n_tags) break;
#pragma unroll
      for (int i = 0; i < MKP_KMAX; i++) {
        if ((uint32_t)i >= tnc_l[t]) break;
        const float p = ((float)mlq[t][i] + 0.5f) / 256.0f;   // quals_to_probs (mod_bam.rs:808-816)
        const uint32_t kk = (tmu_l[t] >> (4 + 4 * i)) & 15u;   // wave-uniform local code
        setk(pk, kk, true, p);
      }
    }
    if (NT > 1 && n_tags > 1) {  // combine_checked's sum test, once on the final map (partial sums of positive terms cannot exceed it)
      float s = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < MKP_KMAX; k2++) if (sm_l & (1u << k2)) s = s + at(pk, k2);
      if (active && s > 1.01f) err = true;
    }
    uint32_t ev_info = 0; float sv = 0.f; bool has_ev = false;
    if (active) {
      const bool edge_keep = !prm.edge_filter ||
          (prm.edge_inverted ? (f < prm.edge_start || f >= L - prm.edge_end) : (f >= prm.edge_start && f < L - prm.edge_end));
      const uint32_t pv = gp0[12 + SH_l];
      contribH |= SH_l;
      if (trimmable && edge_keep) {
        if (SAMPLE) {  // SeqPosBaseModProbs::filter_positions (read_ids_to_base_mod_probs.rs:966-1070)
          bool keep = !prm.only_mapped || mapped;
          if (prm.has_focus) keep = keep && mapped && rpos >= prm.win_start && rpos < prm.win_end
              && ((bedmask[rpos - prm.win_start] >> (aln ^ (uint32_t)sg0)) & 1u);
          // `extract calls`: the collapse left no code in the map: no profile row
          if (keep && prm.sample_mode == 3 && ((pv >> 3) & 7u) == 0u) any_surviving = true;
          else if (keep && prm.sample_mode >= 2) {
            any_surviving = true; uint32_t ob = 0; float am = 0.f; ev_info = summary_info(g0, pv, pk, collapse, &ob, (int)kc_l, &am); obs0 |= ob;
              has_ev = true;
            // `extract calls`: the event carries the forward position (explicit tags: never inferred)
            if (prm.sample_mode == 3) { ev_info |= (uint32_t)sg0 << 2; sv = am; rpos = (int32_t)f; }
          }
          else if (keep) { any_surviving = true; sv = argmax_group(g0, pv, pk, collapse, (int)kc_l); ev_info = MKP_G_TB(g0.misc); has_ev = true; }
        } else {
          any_surviving = true;
          uint32_t ob = 0;
          const int cls = call_group(g0, pv, pk, collapse, &ob, (int)kc_l);
          const uint32_t tally = aln ^ (uint32_t)sg0;  // read_cache.rs:181-188 / FeatureVector::add_feature
          if (tally) obs1 |= ob; else obs0 |= ob;
          if (mapped) {
            const uint32_t cid = cls == 0 ? (uint32_t)MKP_C_FAIL : cls == 1 ? MKP_G_CIDCAN(g0.misc) : ((g0.cids >> (8 * (cls - 2))) & 0xffu);
            ev_info = cid | (tally << 8) | ((uint32_t)b0 << 9) | (aln << 11) | (1u << 12);
            has_ev = true;
          }
        }
      }
    }
    // ballot-compacted, position-ordered append of this batch's events
    const unsigned long long b1 = __ballot(has_ev);
    const uint32_t step_total = (uint32_t)__popcll(b1);
    if (step_total) {
      const uint32_t off = n_ev + (uint32_t)__popcll(b1 & lanemask_lt());
      if (n_ev + step_total > ev_cap) { err = true; if (lane == 0) atomicOr(dev_err, ERR_EVENT_CAP); }
      else if (has_ev) {
        MkpEvent ev; ev.pos = (uint32_t)rpos; ev.info = ev_info; events[ev_base + off] = ev;
        if (SAMPLE) sample_vals[ev_base + off] = sv;
      }
      n_ev += step_total;
    }
    err = __any(err);
    qhead += nb;
  }
  // a delta list must not run past the last occurrence of its base: every entry must have been consumed (mod_bam.rs:705-727)
  if (rev ? (t_cur != 0u) : (t_cur != t_n)) err = true;
  err = __any(err);
  obs0 = wave_or(obs0); obs1 = wave_or(obs1);
  any_surviving = __any(any_surviving);
  // InvalidImplicitMode cannot arise (every tag carries '?'), kept for symmetry with the FAST kernels (read_cache.rs:122-137)
  if (!prm.force_allow && !SAMPLE) {
    const uint32_t mc = wave_or(contribH); uint32_t tagbits = 0;
#pragma unroll
    for (int mi = 0; mi < MKP_MAX_MEMBERS; mi++) if (mc & (1u << mi)) tagbits |= 1u << ((grp0.member_tags >> (4 * mi)) & 15u);
    if (tagbits && (tagbits & ~(uint32_t)lay->default_mask) == 0) err = true;
  }
  if (lane == 0) {
    if (!err && any_surviving) { out.ok = 1; out.n_events = n_ev; out.obs[0] = obs0; out.obs[1] = obs1; }
    else if (!err && duplex) out.ok = 2;   // this group added nothing (all of it edge-filtered): the record still stands if the other group did
    readout[ro_idx] = out;
  }
}

#define DECODE_PARAMS(PRM) const MkpReadHdr* __restrict__ hdrs, uint32_t n_reads, const uint32_t* __restrict__ cigar, const uint8_t* __restrict__ seqs, \
                    const MkpTagRef* __restrict__ tagref, const uint32_t* __restrict__ ranks, const uint8_t* __restrict__ ml, \
                    const MkpLayout* __restrict__ layouts, PRM prm, MkpEvent* __restrict__ events, MkpReadOut* __restrict__ readout, \
                    uint32_t* __restrict__ dev_err, const uint8_t* __restrict__ bedmask, float* __restrict__ sample_vals, const uint32_t* __restrict__ read_ids
#define DECODE_PASS hdrs, n_reads, cigar, seqs, tagref, ranks, ml, layouts, prm, events, readout, dev_err, bedmask, sample_vals, read_ids
// pdep4[(mask << 4) | bits]: the low bits of `bits` deposited onto the set bits of a 4-bit mask
__device__ __forceinline__ void init_pdep4(uint8_t* pdep4) {
  const uint32_t m = (threadIdx.x >> 4) & 15u; uint32_t f = threadIdx.x & 15u, o = 0;
  for (uint32_t i = 0; i < 4; i++) if ((m >> i) & 1u) { o |= (f & 1u) << i; f >>= 1; }
  pdep4[threadIdx.x & 255u] = (uint8_t)o;
  __syncthreads();
}
// Reads are split by the host into three lists (read_ids): FAST layouts with one tag, FAST layouts with two tags, and
// everything else; each list has its own kernel so the common `C+m?` / `C+h?;C+m?` reads run with few registers and
// little LDS (more waves per SIMD) while the general decoder keeps its full generality.
template <bool SAMPLE, int NT> __device__ __forceinline__ void decode_fast_entry(DECODE_PARAMS(const MkpRunParams&)) {
  __shared__ uint32_t lds_queue[4][(1 + NT) * MKP_QCAP];
  __shared__ __attribute__((aligned(16))) uint32_t lds_layouts[4][MKP_LAYOUT_DWORDS];
  __shared__ uint32_t lds_ord[4][NT * MKP_ORD_WORDS];
  __shared__ uint8_t pdep4[256];
  init_pdep4(pdep4);
  decode_read_fast<SAMPLE, NT>(hdrs, n_reads, cigar, seqs, tagref, ranks, ml, layouts, prm, events, readout, dev_err, bedmask, sample_vals,
                               &lds_layouts[0][0], read_ids, &lds_ord[0][0], pdep4, &lds_queue[0][0]);
}
template <bool SAMPLE, int NT> __device__ __forceinline__ void decode_sparse_entry(DECODE_PARAMS(const MkpRunParams&)) {
  __shared__ uint32_t lds_queue[4][2 * MKP_SQCAP];
  __shared__ __attribute__((aligned(16))) uint32_t lds_layouts[4][MKP_LAYOUT_DWORDS];
  __shared__ __attribute__((aligned(16))) uint32_t lds_flags[4][512];
  decode_read_sparse<SAMPLE, NT>(hdrs, n_reads, cigar, seqs, tagref, ranks, ml, layouts, prm, events, readout, dev_err, bedmask, sample_vals,
      &lds_layouts[0][0], read_ids, &lds_queue[0][0], &lds_flags[0][0]);
}
template <bool SAMPLE> __device__ __forceinline__ void decode_general_entry(DECODE_PARAMS(const MkpRunParams&)) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_layouts[4][MKP_LAYOUT_DWORDS];
  __shared__ uint32_t lds_marks[4 * 7][64];   // per wave 448 dwords: ordinal bitmaps [<=8][18] at 0, 512 u16 slots at 192
  __shared__ uint8_t pdep4[256];
  init_pdep4(pdep4);
  const uint32_t widx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  if (widx >= n_reads) return;
  const uint32_t nt = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdrs[read_ids[widx]].n_tags);
#define DECODE_CALL(S, N) decode_read_body<S, N>(hdrs, n_reads, cigar, seqs, tagref, ranks, ml, layouts, prm, events, readout, dev_err, bedmask, sample_vals, read_ids, &lds_layouts[0][0], lds_marks, pdep4)
  if (nt <= 2) DECODE_CALL(SAMPLE, 2); else if (nt <= 4) DECODE_CALL(SAMPLE, 4); else DECODE_CALL(SAMPLE, MKP_MAX_TAGS);
}
extern "C" __global__ void __launch_bounds__(256) mkp_decode_reads(DECODE_PARAMS(MkpRunParams)) { decode_general_entry<false>(DECODE_PASS); }
extern "C" __global__ void __launch_bounds__(256) mkp_decode_fast1(DECODE_PARAMS(MkpRunParams)) { decode_fast_entry<false, 1>(DECODE_PASS); }
extern "C" __global__ void __launch_bounds__(256) mkp_decode_fast2(DECODE_PARAMS(MkpRunParams)) { decode_fast_entry<false, 2>(DECODE_PASS); }
// Waves per SIMD of the SPARSE event decoders.  Left alone the compiler took 77 / 85 VGPRs and 106 SGPRs = six / five waves; seven
// (72 VGPRs, <= 96 SGPRs: the scalar file admits seven, MI355X_MICROARCH.md "Residency") gave C2 decode 0.509 -> 0.481 ms, hemi 2.40 -> 2.28,
// and eight spilled (0.57 / 2.60).  Since the producer runs as a loop of its own and the consumer's descriptor words are opaque (both
// in decode_read_sparse) the kernels need 59 / 60 VGPRs, and EIGHT waves (<= 64 VGPRs, <= 80 SGPRs) win: C2 decode 0.428 -> 0.405 ms, hemi
// decode 1.955 -> 1.859 (0.452 / 2.03 before both changes).  A/B on one box, tools/dbg/ab.sh.
#ifndef MKP_SPARSE_WAVES
#define MKP_SPARSE_WAVES 8
#endif
#define MKP_SPARSE_LB __launch_bounds__(256, MKP_SPARSE_WAVES)
extern "C" __global__ void MKP_SPARSE_LB mkp_decode_sparse1(DECODE_PARAMS(MkpRunParams)) { decode_sparse_entry<false, 1>(DECODE_PASS); }
extern "C" __global__ void MKP_SPARSE_LB mkp_decode_sparse2(DECODE_PARAMS(MkpRunParams)) { decode_sparse_entry<false, 2>(DECODE_PASS); }
// the same walks in threshold-sampling mode (reads_sampler / thresholds.rs:121-159): separate kernels so profiles keep the two apart
extern "C" __global__ void __launch_bounds__(256) mkp_sample_reads(DECODE_PARAMS(MkpRunParams)) { decode_general_entry<true>(DECODE_PASS); }
extern "C" __global__ void __launch_bounds__(256) mkp_sample_fast1(DECODE_PARAMS(MkpRunParams)) { decode_fast_entry<true, 1>(DECODE_PASS); }
extern "C" __global__ void __launch_bounds__(256) mkp_sample_fast2(DECODE_PARAMS(MkpRunParams)) { decode_fast_entry<true, 2>(DECODE_PASS); }
extern "C" __global__ void __launch_bounds__(256) mkp_sample_sparse1(DECODE_PARAMS(MkpRunParams)) { decode_sparse_entry<true, 1>(DECODE_PASS); }
extern "C" __global__ void __launch_bounds__(256) mkp_sample_sparse2(DECODE_PARAMS(MkpRunParams)) { decode_sparse_entry<true, 2>(DECODE_PASS); }

// ----------------------------------------------------------------------------------------------

// mkp_pileup_tiles — accumulate + emit.  Persistent 1024-thread workgroups, two per CU (8 waves per SIMD).  LDS holds one
// tile's tallies as [row][slot] u32 with the '+' strand tally in the low and the '-' strand tally in the high 16 bits (the host
// refuses shards in which more than 65535 reads overlap one position); rows = the strand tally's counters followed by the
// observed-code slots; consecutive slots sit on consecutive banks.  Waves draw the tile's reads from an LDS ticket:
//   * observed codes: +1 / -1 at the slots bounding the read's span (and around ref-skips) — an interval-OR as a difference array;
//   * the read's call events inside the tile (position-sorted slice): one LDS atomic on the call's counter and -1 on NoCall(read base);
//   * depth walk (htslib pileup columns, pileup/mod.rs:783-939): per 64-op CIGAR window the reference-consuming ops that
//     cover at least one slot are compacted through LDS and their first slots marked in the wave's bitmap; deletions are
//     +1 / -1 on a difference array; then one lane per slot, 64 slots per step: op index = ballot over the compacted starts
//     + mbcnt of the aligned 64-bit bitmap window, one ds_bpermute fetches the op's packed (query offset, kind), one byte load
//     fetches the base, a 1 KB table in LDS maps (strand, query parity, SEQ byte) to the NoCall row, one LDS atomic.
// With focus positions (FOCUS) the walk touches only the focus slots of each op — a --cpg run visits ~2 % of the aligned bases
// and a tile covers ~50x more reference for the same LDS.  After a barrier the difference arrays are prefix-summed in place and
// the rows of FeatureVector::decode (412-446) / add_tally_to_counts (283-410) / combine_strand_features (469-561) are produced
// straight from LDS: counted, reserved in the row buffer with one atomic per tile, written.  mkp_scan_tiles / mkp_gather_rows
// put the tiles' row runs in genome order.
// HEMI (pileup-hemi, duplex.rs:241-339): the tally columns are the '+' motif positions; a read's '+' tally call at such a position
// and its '-' tally call at the partner position form one pattern count; everything else about the walk is the focus kernel's.
template <bool FOCUS, int UNROLL, bool KEYED, bool HEMI = false>
__device__ __forceinline__ void pileup_tiles_body(const MkpReadHdr* __restrict__ hdrs, const uint32_t* __restrict__ cigar,
    const uint8_t* __restrict__ seqs,
                 const MkpEvent* __restrict__ events, const MkpReadOut* __restrict__ readout, const MkpTile* __restrict__ tiles, uint32_t n_tiles,
                 const MkpRunParams* __restrict__ prmp, const uint32_t* __restrict__ slotbm, const uint8_t* __restrict__ focus,
                     const MkpCombo* __restrict__ combos,
                 uint32_t* __restrict__ rows_base, uint32_t* __restrict__ row_cursor, uint32_t* __restrict__ tile_row_off,
                     uint32_t* __restrict__ tile_row_cnt,
                 const uint2* __restrict__ chunk_pfx, uint32_t* __restrict__ dev_err, uint32_t key_arg) {
  // --partition-tag (KEYED kernels): low 16 bits = the key this pass tallies, high 16 bits = index of the pass; otherwise unused
  const uint32_t key_filter = KEYED ? (key_arg & 0xffffu) : 0u, key_run = KEYED ? (key_arg >> 16) : 0u;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t next_read;
  __shared__ uint32_t wave_tot[PILEUP_WAVES];
  __shared__ uint32_t row_base, scan_carry;
  __shared__ StreamProg rowprog;   // dense tiles: the row program of the emission (mkp_dev_rows.hpp)
  // SEQ byte -> NoCall row of the base a query index selects: rowlut[strand][query parity][byte]; 15 = not A/C/G/T.
  // (BAM packs two bases per byte, high nibble first; on the '-' strand the tallied base is the complement.)
  __shared__ uint8_t rowlut[2][2][256];
  __shared__ __attribute__((aligned(16))) uint32_t prm_lds[(sizeof(MkpRunParams) + 3) / 4];
  __shared__ __attribute__((aligned(16))) uint32_t combo_lds[64 * sizeof(MkpCombo) / 4];   // <= 64 motif-id combos (checked at mkp_shard_begin)
  {
    static_assert(PILEUP_THREADS == 1024, "the row table below is filled one entry per thread");
    const uint32_t t = threadIdx.x, byte = t & 255u, par = (t >> 8) & 1u, st = (t >> 9) & 1u;
    const uint32_t nib = par ? (byte & 15u) : (byte >> 4);
    const unsigned long long LUT = st ? 0xfffffff0fff1f23fULL : 0xfffffff3fff2f10fULL;
    rowlut[st][par][byte] = (uint8_t)((LUT >> (4u * nib)) & 15u);
    for (uint32_t kq = threadIdx.x; kq < sizeof(MkpRunParams) / 4; kq += PILEUP_THREADS) prm_lds[kq] = reinterpret_cast<const uint32_t*>(prmp)[kq];
    if (prmp->has_focus) for (uint32_t kq = threadIdx.x; kq < prmp->n_combos * (sizeof(MkpCombo) / 4); kq += PILEUP_THREADS) combo_lds[kq] = reinterpret_cast<const uint32_t*>(combos)[kq];
  }
  __syncthreads();
  const MkpRunParams& prm = *reinterpret_cast<const MkpRunParams*>(prm_lds);
  const MkpCombo* combos_l = reinterpret_cast<const MkpCombo*>(combo_lds);
  const uint32_t S = prm.slot_cap, W = prm.focus_words;
#ifdef MKP_DEBUG
  const uint32_t dbg = prm.debug_skip;   // ablation runs (env MKP_DEBUG_SKIP): 1 depth walk, 2 events, 4 row emission, 8 SEQ phase of the focus walk
#else
  constexpr uint32_t dbg = 0;
#endif
  const uint32_t n_counters = HEMI ? prm.hemi_counters : prm.n_counters, n_oslots = HEMI ? 0u : prm.n_slots;
  const uint32_t tal_words = (n_counters + n_oslots) * S;
  uint32_t* __restrict__ tal = lds;                       // [n_counters + n_oslots][S], packed
  uint32_t* __restrict__ obs = lds + n_counters * S;      // observed-code difference arrays
  uint32_t* __restrict__ fbm = lds + tal_words;           // FOCUS: bitmap words [W], running popcount [W], slot -> position [S]
  uint32_t* __restrict__ fpfx = fbm + W;
  int32_t* __restrict__ fpos = reinterpret_cast<int32_t*>(fpfx + W);
  const uint32_t focus_total = FOCUS ? 2u * W + S : 0u;
  const int lane = lane_id();
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // per-wave scratch behind that: op-start bitmap over the tile's slots + CIGAR compaction buffer
  const uint32_t bm_words = MKP_PILEUP_BM_WORDS(S), wave_words = MKP_PILEUP_WAVE_WORDS(S, W);
  uint32_t* __restrict__ bm = lds + tal_words + focus_total + wave * wave_words;
  uint2* __restrict__ comp = reinterpret_cast<uint2*>(bm + bm_words);
  uint32_t* __restrict__ qk = bm;   // focus kernel: per slot of the current read's visit, (query index << 2) | kind
  const uint32_t TS4 = S * 4u;
  // XCD-aware mapping: consecutive workgroups land on different XCDs (b % 8); give each XCD a
  // contiguous run of tiles so the reads shared by neighbouring tiles stay in one L2.
  // One workgroup per tile; the dispatcher hands tiles to CUs as workgroups retire (two fit a CU).
  {
  uint32_t tix = blockIdx.x;
  { const uint32_t per = n_tiles / 8u; if (per && tix < per * 8u) tix = (tix & 7u) * per + (tix >> 3); }
  const MkpTile tl = tiles[tix];
  const int32_t T0h = tl.r0 - MKP_HALO, T1h = tl.r1 + MKP_HALO;
  // zero the tallies and the per-wave scratch (the focus arrays are rewritten below)
  for (uint32_t k = threadIdx.x; k < tal_words; k += PILEUP_THREADS) lds[k] = 0;
  // focus: kind 3 = no base
  for (uint32_t k = threadIdx.x; k < PILEUP_WAVES * wave_words; k += PILEUP_THREADS) lds[tal_words + focus_total + k] = FOCUS ? 3u : 0u;
  if (threadIdx.x == 0) { next_read = tl.first; scan_carry = 0; }
  SlotMap<FOCUS> sm; sm.bm = fbm; sm.pfx = fpfx; sm.fpos = fpos; sm.T0h = T0h; sm.lbase = T0h;
  uint32_t n_tslots = (uint32_t)(T1h - T0h);   // tally columns in use
  if (FOCUS) {
    // the tile's slice of the slot bitmap: words masked to [T0h, T1h), their running popcount, and the slot -> position list
    const uint32_t g0 = (uint32_t)(T0h - (prm.win_start - MKP_SLOTBM_MARGIN)), sh = g0 & 31u, nbits = (uint32_t)(T1h - T0h);
    const uint32_t nw = (sh + nbits + 31u) >> 5;
    sm.lbase = T0h - (int32_t)sh;
    __syncthreads();   // scan_carry = 0 visible; previous tile's readers of the focus arrays are done
    for (uint32_t k0 = 0; k0 < nw + 1u; k0 += PILEUP_THREADS) {
      const uint32_t k = k0 + threadIdx.x;
      uint32_t w = 0;
      if (k < nw) {
        w = slotbm[(g0 >> 5) + k];
        if (k == 0) w &= ~((1u << sh) - 1u);
        const uint32_t endb = sh + nbits - 32u * k;   // bits of this word below the range end
        if (endb < 32u) w &= (1u << endb) - 1u;
      }
      const uint32_t cnt = (uint32_t)__popc(w), inc = wave_incl_scan(cnt);
      if (lane == 63) wave_tot[wave] = inc;
      __syncthreads();
      uint32_t woff = scan_carry;
      for (uint32_t w2 = 0; w2 < wave; w2++) woff += wave_tot[w2];
      const uint32_t excl = woff + inc - cnt;
      if (k <= nw) { fbm[k] = w; fpfx[k] = excl; }
      for (uint32_t ww = w, j = 0; ww; ww &= ww - 1u, j++) fpos[excl + j] = sm.lbase + (int32_t)(32u * k) + (__ffs((int)ww) - 1);
      __syncthreads();
      if (threadIdx.x == PILEUP_THREADS - 1) scan_carry = woff + inc;
    }
    __syncthreads();
    n_tslots = scan_carry;
  }
  __syncthreads();

  // reads are handed out one at a time (LDS ticket) so waves finish the tile together whatever the reads' spans; the next
  // read's header and decode summary are requested while the current read is processed (one global round trip less per read)
  const uint32_t rid_end = tl.last;
  uint32_t rid_nx; MkpReadHdr h_nx; MkpReadOut ro_nx;
  { uint32_t ticket = 0; if (lane == 0) ticket = atomicAdd(&next_read, 1u); rid_nx = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket); }
  { const uint32_t r0 = min(rid_nx, rid_end - 1u); h_nx = hdrs[r0]; ro_nx = readout[r0]; }   // (tiles hold at least one candidate read)
  if (n_tslots) for (;;) {
    const uint32_t rid = rid_nx;
    if (rid >= rid_end) break;
    const MkpReadHdr h = h_nx; const MkpReadOut ro = ro_nx;
    { uint32_t ticket = 0; if (lane == 0) ticket = atomicAdd(&next_read, 1u); rid_nx = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket); }
    { const uint32_t r0 = min(rid_nx, rid_end - 1u); h_nx = hdrs[r0]; ro_nx = readout[r0]; }
    if (h.ref_end <= T0h || h.ref_start >= T1h) continue;
    if (KEYED && (h.flags >> MKP_RF_KEY_SHIFT) != key_filter) continue;   // --partition-tag: one pass per key
    // the read's slot range in this tile; a read that covers no slot leaves nothing here
    const int32_t span_a = max(h.ref_start, T0h), span_b = min(h.ref_end, T1h);
    const uint32_t rs_a = sm.rank(span_a), rs_b = sm.rank(span_b);
    if (FOCUS && rs_a == rs_b) continue;
    const uint32_t aln = (h.flags & MKP_RF_REVERSE) ? 1u : 0u;
    const uint8_t* __restrict__ seq = seqs + h.seq_off;
    // depth walk: htslib pileup columns (match -> base, D -> delete, N -> ref-skip).  One lane per slot;
    // the op covering a slot = (ops whose first slot is at or before it) - 1, counted with the wave's op-start bitmap.
    uint32_t q_run = 0; int32_t r_run = h.ref_start; uint32_t c_first = 0;
    {  // skip the 64-op CIGAR chunks that end before the tile: the host's per-chunk offsets say where the walk starts
      const uint32_t nch = (h.n_cigar + 63u) >> 6;
      if (nch > 1 && T0h > h.ref_start) {
        const uint32_t rel = (uint32_t)(T0h - h.ref_start);
        for (uint32_t b0 = 0; b0 < nch; b0 += 64) {
          const uint32_t kidx = b0 + (uint32_t)lane;
          const uint2 e = kidx < nch ? chunk_pfx[h.chunk_off + kidx] : make_uint2(0u, 0xffffffffu);
          const uint32_t cnt = (uint32_t)__popcll(__ballot(kidx < nch && e.y <= rel));   // offsets ascend: a prefix of the lanes
          if (cnt) {
            q_run = (uint32_t)__builtin_amdgcn_readlane((int)e.x, (int)(cnt - 1u));
            r_run = h.ref_start + (int32_t)__builtin_amdgcn_readlane((int)e.y, (int)(cnt - 1u));
            c_first = 64u * (b0 + cnt - 1u);
          }
          if (cnt < 64u) break;
        }
      }
    }
    // the first chunk's CIGAR words are requested now, ahead of the event work; every later chunk is requested one chunk ahead
    auto load2 = [&](uint32_t c) { uint2 r; const uint32_t i = c + 2u * (uint32_t)lane; r.x = i < h.n_cigar ? cigar[h.cigar_off + i] : 5u;
      r.y = i + 1u < h.n_cigar ? cigar[h.cigar_off + i + 1u] : 5u; return r; };
    uint32_t w_next = 5u; uint2 w2_next = make_uint2(5u, 5u);   // focus kernel: two ops per lane
    if (FOCUS) w2_next = load2(c_first); else w_next = (c_first + (uint32_t)lane < h.n_cigar) ? cigar[h.cigar_off + c_first + lane] : 5u;
    // observed mod codes: +1 over the read's span (add_mod_codes_for_record, pileup/mod.rs:831-835)
    if (!HEMI && ro.ok && lane < 2) {
      uint32_t m = lane ? ro.obs[1] : ro.obs[0];
      while (m) {
        const uint32_t sl = (uint32_t)__ffs((int)m) - 1u; m &= m - 1u;
        atomicAdd(&obs[sl * S + rs_a], lane ? 0x10000u : 1u);
        if (rs_b < n_tslots) atomicAdd(&obs[sl * S + rs_b], 0u - (lane ? 0x10000u : 1u));
      }
    }
    if (HEMI) {
      // get_duplex_mod_call (read_cache.rs:422-462) over the read's call events inside the tile (sorted by position): a '+' tally
      // call at a slot = the positive-strand half; its partner is the read's '-' tally call on the same primary base at
      // position + hemi_off (a few events away).  Both present -> one pattern (or Filtered) count and the NoCall the walk below
      // adds for this base is taken back; otherwise the base stays a NoCall.  A record whose tags failed carries, instead of
      // calls, the one NoCall per interval the reference's cache leaves for it (mkp_hemi_failed_reads).
      if (ro.n_events) {
        const MkpEvent* __restrict__ ev = events + h.event_off;
        const uint32_t lo = h.ref_start >= T0h ? 0u : event_lower_bound(ev, ro.n_events, T0h);
        const int32_t hoff = prm.hemi_off;
        // threshold base of the call (read_cache.rs:147-150)
        auto pb_of = [](uint32_t info) { const uint32_t b = (info >> 9) & 3u; return (((info >> 11) ^ (info >> 8)) & 1u) ? 3u - b : b; };
        for (uint32_t k = lo + lane;; k += 64) {
          bool in = k < ro.n_events;
          MkpEvent e; e.pos = 0; e.info = 0;
          if (in) { e = ev[k]; in = (int32_t)e.pos < T1h; }
          if (in && sm.is_slot((int32_t)e.pos)) {
            const uint32_t i = sm.rank((int32_t)e.pos);
            if (!ro.ok) atomicAdd(&tal[(e.info & 0xffu) * S + i], 1u);
            else if (!(e.info & 0x100u)) {
              const uint32_t pbA = pb_of(e.info);
              const int32_t q = (int32_t)e.pos + hoff;
              uint32_t binfo = 0xffffffffu;
              if (hoff >= 0) for (uint32_t j = k + 1u; j < ro.n_events; j++) {
                const MkpEvent f = ev[j];
                if ((int32_t)f.pos > q) break;
                if ((int32_t)f.pos == q && (f.info & 0x100u) && pb_of(f.info) == pbA) { binfo = f.info; break; }
              }
              if (hoff <= 0 && binfo == 0xffffffffu) for (uint32_t j = k; j-- > 0u;) {
                const MkpEvent f = ev[j];
                if ((int32_t)f.pos < q) break;
                if ((int32_t)f.pos == q && (f.info & 0x100u) && pb_of(f.info) == pbA) { binfo = f.info; break; }
              }
              if (binfo != 0xffffffffu && q >= 0) {
                const uint32_t elA = prm.hemi_el[e.info & 0xffu], elB = prm.hemi_el[binfo & 0xffu];
                const uint32_t cid = (elA == 0xffu || elB == 0xffu) ? (uint32_t)MKP_H_FAIL + pbA
                    : (uint32_t)prm.hemi_pat_base[pbA] + elA * prm.hemi_nel[pbA] + elB;
                atomicAdd(&tal[cid * S + i], 1u);
                atomicAdd(&tal[(MKP_H_NC + pbA) * S + i], 0u - 1u);
              }
            }
          }
          if (!__any(in)) break;
        }
      }
    } else
    // the read's call events inside the tile (sorted by position)
    if (ro.ok && ro.n_events && !(dbg & 2u)) {
      const MkpEvent* __restrict__ ev = events + h.event_off;
      const uint32_t lo = h.ref_start >= T0h ? 0u : event_lower_bound(ev, ro.n_events, T0h);   // a read that starts in the tile: no search
      for (uint32_t k = lo + lane;; k += 64) {
        bool in = k < ro.n_events;
        MkpEvent e; e.pos = 0; e.info = 0;
        if (in) { e = ev[k]; in = (int32_t)e.pos < T1h; }
        if (in && sm.is_slot((int32_t)e.pos)) {
          const uint32_t i = sm.rank((int32_t)e.pos);
          atomicAdd(&tal[(e.info & 0xffu) * S + i], (e.info & 0x100u) ? 0x10000u : 1u);
          if (e.info & (1u << 12))  // the base is a call, not a NoCall (pileup/mod.rs:889-938)
            atomicAdd(&tal[(MKP_C_NC + ((e.info >> 9) & 3u)) * S + i], 0u - ((e.info & 0x800u) ? 0x10000u : 1u));
        }
        if (!__any(in)) break;
      }
    }
    const uint32_t inc = HEMI ? 1u : (aln ? 0x10000u : 1u);   // this alignment strand's half of the packed tallies (hemi: plain counters)
    const uint8_t* __restrict__ lut = &rowlut[0][0][0];
    const uint32_t aln2 = HEMI ? 0u : aln << 1;   // hemi: the primary base is the SEQ base as stored, whatever the strand (duplex.rs:308-313)
    const uint32_t lanebase = lds_addr(tal) + 4u * (uint32_t)lane;   // LDS byte address of (row 0, slot `lane`)
    const uint32_t fposbase = lds_addr(fpos) + 4u * (uint32_t)lane;
    const uint32_t qbase = (uint32_t)(0 - h.ref_start) - (1u << 26);   // query index = position + qbase + packed offset
    const uint32_t last_byte = (h.l_seq - 1u) >> 1;
    if (FOCUS && (dbg & 1u)) {} else
    if (FOCUS) {
      // Focus runs: the walk is driven by the slots, not by the ops, 128 CIGAR ops per window (two per lane).  The window's slots
      // [S_lo, S_hi) are enumerated 64 at a time (lane = slot, position from the tile's slot list); the lane holding a position's
      // op is the number of lanes whose inclusive reference end is at or before it (6 ds_bpermute steps over the prefix sums the
      // window has anyway), three more bpermutes bring that lane's split point and the packed (query offset, kind) of its two
      // ops.  No per-op rank queries, no compaction, no op-start bitmap; deletions are counted directly.
      // Phase 1 (here): the slots get their packed (query index, kind); phase 2 (after the loop) fetches the bases of all the
      // read's slots with several loads in flight — the SEQ loads do not sit in the CIGAR dependency chain.
      for (uint32_t c0 = c_first; c0 < h.n_cigar; c0 += 128) {
        if (r_run >= T1h) break;
        const uint2 w2 = w2_next;
        if (c0 + 128u < h.n_cigar) w2_next = load2(c0 + 128u);
        const uint32_t op0 = w2.x & 15u, len0 = w2.x >> 4, op1 = w2.y & 15u, len1 = w2.y >> 4;
        const uint32_t ql0 = op_consumes_query(op0) ? len0 : 0u, rl0 = op_consumes_ref(op0) ? len0 : 0u;
        const uint32_t ql1 = op_consumes_query(op1) ? len1 : 0u, rl1 = op_consumes_ref(op1) ? len1 : 0u;
        const uint32_t qe = wave_incl_scan(ql0 + ql1), re = wave_incl_scan(rl0 + rl1);
        const uint32_t Qtot = (uint32_t)__builtin_amdgcn_readlane((int)qe, 63), Rtot = (uint32_t)__builtin_amdgcn_readlane((int)re, 63);
        const int32_t c_lo = max(r_run, T0h), c_hi = min(r_run + (int32_t)Rtot, T1h);
        if (c_lo < c_hi) {
          const uint32_t S_lo = sm.rank(c_lo), S_hi = sm.rank(c_hi);
          if (S_lo < S_hi) {
            const uint32_t qs0 = q_run + qe - (ql0 + ql1), qs1 = qs0 + ql0;
            const uint32_t mid = re - rl1;                                     // window-relative reference offset where the lane's second op starts
            const int32_t rs0 = r_run + (int32_t)(re - (rl0 + rl1)), rs1 = r_run + (int32_t)mid;
            const uint32_t kind0 = op_is_match(op0) ? 0u : (op0 == 2 ? 1u : 2u), kind1 = op_is_match(op1) ? 0u : (op1 == 2 ? 1u : 2u);
            const uint32_t pk0 = ((uint32_t)((int32_t)qs0 - (rs0 - h.ref_start) + (1 << 26)) << 5) | (kind0 << 3);  // q = (pos - ref_start) + D
            const uint32_t pk1 = ((uint32_t)((int32_t)qs1 - (rs1 - h.ref_start) + (1 << 26)) << 5) | (kind1 << 3);
            // ref-skips: the read is not in these columns (alignment.is_refskip())
            if (!HEMI && ro.ok && __any((op0 == 3 && rl0 > 0) || (op1 == 3 && rl1 > 0))) {
              for (int j = 0; j < 2; j++) {
                const bool skipop = j ? (op1 == 3 && rl1 > 0) : (op0 == 3 && rl0 > 0);
                const int32_t a0 = j ? rs1 : rs0, b0 = a0 + (int32_t)(j ? rl1 : rl0);
                const int32_t pa = min(max(a0, c_lo), c_hi), pb = min(max(b0, c_lo), c_hi);
                const uint32_t sa = sm.rank(pa), sb = sm.rank(pb);
                if (skipop && sa < sb) for (uint32_t s = 0; s < 2; s++) {
                  uint32_t m = ro.obs[s];
                  while (m) {
                    const uint32_t sl = (uint32_t)__ffs((int)m) - 1u; m &= m - 1u;
                    atomicAdd(&obs[sl * S + sa], 0u - (s ? 0x10000u : 1u));
                    if (sb < n_tslots) atomicAdd(&obs[sl * S + sb], s ? 0x10000u : 1u);
                  }
                }
              }
            }
            for (uint32_t s0 = S_lo; s0 < S_hi; s0 += 64) {
              const uint32_t c = s0 + (uint32_t)lane;
              const bool valid = c < S_hi;
              const int32_t p = valid ? fpos[c] : c_lo;
              const uint32_t rel = (uint32_t)(p - r_run);
              const int ol = find_op(re, rel) & 63;
              const uint32_t o_mid = (uint32_t)__shfl((int)mid, ol, 64), o_pk0 = (uint32_t)__shfl((int)pk0, ol, 64),
                  o_pk1 = (uint32_t)__shfl((int)pk1, ol, 64);
              const uint32_t my_pk = rel < o_mid ? o_pk0 : o_pk1;
              const uint32_t qq = (uint32_t)p + qbase + (my_pk >> 5);
              if (valid) qk[c - rs_a] = (qq << 2) | ((my_pk >> 3) & 3u);
            }
          }
        }
        q_run += Qtot; r_run += (int32_t)Rtot;
      }
    } else
    for (uint32_t c0 = c_first; c0 < h.n_cigar; c0 += 64) {
      if (r_run >= T1h) break;
      const uint32_t w = w_next;
      if (c0 + 64u < h.n_cigar) w_next = (c0 + 64u + (uint32_t)lane < h.n_cigar) ? cigar[h.cigar_off + c0 + 64u + lane] : 5u;
      const uint32_t op = w & 15u, len = w >> 4;
      const uint32_t qlen = op_consumes_query(op) ? len : 0u, rlen = op_consumes_ref(op) ? len : 0u;
      const uint32_t qe = wave_incl_scan(qlen), re = wave_incl_scan(rlen);
      const uint32_t qs = q_run + qe - qlen;
      const int32_t rs = r_run + (int32_t)(re - rlen);
      const uint32_t Qtot = (uint32_t)__builtin_amdgcn_readlane((int)qe, 63), Rtot = (uint32_t)__builtin_amdgcn_readlane((int)re, 63);
      const int32_t c_lo = max(r_run, T0h), c_hi = min(r_run + (int32_t)Rtot, T1h);
      if (c_lo < c_hi) {
        // the op's slots inside the tile: [sa, sb)
        const int32_t pa = min(max(rs, c_lo), c_hi), pb = min(max(rs + (int32_t)rlen, c_lo), c_hi);
        const uint32_t sa = sm.rank(pa), sb = sm.rank(pb);
        const bool covers = rlen > 0 && sa < sb;
        if (op == 3 && ro.ok && covers) {  // ref-skip: the read is not in these columns (alignment.is_refskip())
          for (uint32_t s = 0; s < 2; s++) {
            uint32_t m = ro.obs[s];
            while (m) {
              const uint32_t sl = (uint32_t)__ffs((int)m) - 1u; m &= m - 1u;
              atomicAdd(&obs[sl * S + sa], 0u - (s ? 0x10000u : 1u));
              if (sb < n_tslots) atomicAdd(&obs[sl * S + sb], s ? 0x10000u : 1u);
            }
          }
        }
        if (op == 2 && covers) {  // deletion columns (alignment.is_del()): +1/-1 on the strand's DEL row, summed after the barrier
          atomicAdd(&tal[MKP_C_DEL * S + sa], inc);
          if (sb < n_tslots) atomicAdd(&tal[MKP_C_DEL * S + sb], 0u - inc);
        }
        const uint32_t S_lo = sm.rank(c_lo), S_hi = sm.rank(c_hi);   // the chunk's slots
        if (S_lo < S_hi) {
        // compact the window's slot-covering ops to the low lanes: {first slot, packed(query offset, kind)}
        const unsigned long long refbal = __ballot(covers);
        const uint32_t nref = (uint32_t)__popcll(refbal);
        const uint32_t ci = (uint32_t)__popcll(refbal & lanemask_lt());
        const uint32_t kind = op_is_match(op) ? 0u : (op == 2 ? 1u : 2u);
        // q = (pos - ref_start) + D; kind sits where it ORs into the row
        const uint32_t pk = ((uint32_t)((int32_t)qs - (rs - h.ref_start) + (1 << 26)) << 5) | (kind << 3);
        if (covers) comp[ci] = make_uint2(sa, pk);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint2 cc = comp[lane];
        const bool cvalid = (uint32_t)lane < nref;
        const uint32_t c_sa = cc.x; const uint32_t c_pk = cc.y;
        const uint32_t c_key = cvalid ? c_sa : 0x7fffffffu;
        const uint32_t e_lo = lds_addr(tal) + 4u * S_lo, e_span = 4u * (S_hi - S_lo);
        const bool mark = cvalid && c_sa > S_lo;
        const uint32_t mrel = c_sa - 1u;   // bit m set <=> an op's first slot is m+1: "starts at or before slot c" = bits strictly below c
        if (mark) atomicOr(&bm[mrel >> 5], 1u << (mrel & 31u));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // 64 slots per step, aligned so a step reads one aligned 64-bit window of the bitmap;
        // UNROLL steps are issued together (bitmap read -> bpermute -> SEQ byte load) before any tally update.
        // ops started at or before the first slot of window kk: a ballot over the compacted starts (no LDS dependency)
        const uint32_t k0 = S_lo >> 6, k1 = (S_hi - 1u) >> 6;
        // Only the first and the last group of a chunk hold out-of-range lanes or repeated (clamped) windows; the groups in
        // between run a version without the clamps and the range check.
        auto group = [&](uint32_t kb, auto edge_c) {
          constexpr bool EDGE = decltype(edge_c)::value;
          uint32_t pkv[UNROLL], byte[UNROLL], qq[UNROLL], idx[UNROLL], kk[UNROLL]; uint2 Wd[UNROLL]; int32_t pp[UNROLL];
          // stage by stage across the UNROLL windows, so the LDS reads, the bpermutes and the global loads of the group overlap
#pragma unroll
          for (int j = 0; j < UNROLL; j++) kk[j] = EDGE ? min(kb + (uint32_t)j, k1) : kb + (uint32_t)j;
#pragma unroll
          for (int j = 0; j < UNROLL; j++) Wd[j] = *reinterpret_cast<const uint2*>(bm + 2u * kk[j]);
#pragma unroll
          for (int j = 0; j < UNROLL; j++) {
            if (FOCUS) pp[j] = (64u * kk[j] + (uint32_t)lane < n_tslots)
                ? *(const __attribute__((address_space(3))) int32_t*)(uintptr_t)(fposbase + 256u * kk[j]) : T0h;
            else pp[j] = T0h + (int32_t)(64u * kk[j]) + lane;
          }
#pragma unroll
          for (int j = 0; j < UNROLL; j++) {
            const uint32_t wstart = 64u * kk[j];
            idx[j] = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(c_key <= (EDGE ? max(S_lo, wstart) : wstart))) - 1u;
          }
#pragma unroll
          for (int j = 0; j < UNROLL; j++)
            pkv[j] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx[j] + __builtin_amdgcn_mbcnt_hi(Wd[j].y,
                __builtin_amdgcn_mbcnt_lo(Wd[j].x, 0u))) << 2), (int)c_pk);
#pragma unroll
          for (int j = 0; j < UNROLL; j++) {
            qq[j] = (uint32_t)pp[j] + qbase + (pkv[j] >> 5);
            byte[j] = seq[min(qq[j] >> 1, last_byte)];   // lanes on D/N ops or outside the span read a clamped (ignored) byte
          }
#pragma unroll
          for (int j = 0; j < UNROLL; j++) {
            const uint32_t t = (qq[j] & 1u) | aln2;
            const uint32_t rowt = (uint32_t)lut[(t << 8) | byte[j]] | (pkv[j] & 0x18u);   // >= 8: not ACGT, or the lane sits on a D/N op
            const uint32_t la = lanebase + 256u * kk[j];
            bool ok = rowt < 8u;
            if (EDGE) ok = ok && (la - e_lo) < e_span && kb + (uint32_t)j <= k1;
            if (ok) lds_add(la + __umul24(rowt, TS4), inc);
          }
        };
        for (uint32_t kb = k0; kb <= k1; kb += UNROLL) {
          if (kb == k0 || kb + UNROLL > k1) group(kb, std::true_type{}); else group(kb, std::false_type{});
        }
        if (mark) bm[mrel >> 5] = 0;
        }
      }
      q_run += Qtot; r_run += (int32_t)Rtot;
    }
    if (FOCUS && !(dbg & 9u)) {
      // phase 2: bases of the read's slots [rs_a, rs_b) in this tile, 4 x 64 slots per round with their SEQ loads in flight together
      // (slots the CIGAR phase did not reach — a read whose CIGAR ends early — keep kind 3 = nothing)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t nsl = rs_b - rs_a;
      for (uint32_t g0 = 0; g0 < nsl; g0 += 256) {
        uint32_t v[4], byte[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint32_t i = g0 + 64u * j + (uint32_t)lane; v[j] = i < nsl ? qk[i] : 3u; }
#pragma unroll
        for (int j = 0; j < 4; j++) byte[j] = seq[min(v[j] >> 3, last_byte)];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t i = g0 + 64u * j + (uint32_t)lane, kind = v[j] & 3u;
          const uint32_t t = ((v[j] >> 2) & 1u) | aln2;
          const uint32_t rowt = (uint32_t)lut[(t << 8) | byte[j]];
          const uint32_t a0 = lds_addr(tal) + 4u * (rs_a + i);
          // hemi: a failed record gives no feature (its one NoCall came in as an event)
          if (kind == 0u && rowt < 8u && (!HEMI || ro.ok)) lds_add(a0 + __umul24(rowt, TS4), inc);
          if (kind == 1u) lds_add(a0 + __umul24((uint32_t)MKP_C_DEL, TS4), inc);   // alignment.is_del()
          if (i < nsl) qk[i] = 3u;   // left clean for the wave's next read
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __syncthreads();
  // difference arrays -> counts, in place and still packed (the sums are exact): deletions, then observed codes per slot
  for (uint32_t a = wave + (FOCUS ? 1u : 0u); a < n_oslots + 1u; a += PILEUP_WAVES) {   // (focus runs count deletions directly)
    uint32_t* __restrict__ arr = a == 0 ? tal + MKP_C_DEL * S : obs + (a - 1u) * S;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_tslots; b0 += 64) {
      const uint32_t v = (b0 + lane < n_tslots) ? arr[b0 + lane] : 0u;
      const uint32_t sc = wave_incl_scan(v);
      if (b0 + lane < n_tslots) arr[b0 + lane] = sc + carry;
      carry += (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
    }
  }
  if (threadIdx.x == 0) scan_carry = 0;
  __syncthreads();
  // rows of the tile straight from LDS: count, reserve, write (slot order = position order).  One tile per workgroup: nothing of
  // the accumulate phase is live here and nothing of this phase is live there, so neither raises the other's register count
  if (dbg & 4u) {}
  // dense tiles: row-major emission; the row map lives in the per-wave scratch of the accumulate phase (dead behind the barrier above)
  else if (!FOCUS && !HEMI)
    emit_dense_rows(tal, S, n_counters, n_tslots, T0h, tl, KEYED ? key_run * n_tiles + tix : tix, key_filter, prm, rowprog,
        lds + tal_words + focus_total, min(8192u, PILEUP_WAVES * wave_words),
                    rows_base, row_cursor, tile_row_off, tile_row_cnt, dev_err, wave_tot, &row_base, &scan_carry);
  else emit_tile_rows<FOCUS, HEMI>(tal, sm, n_tslots, tl, KEYED ? key_run * n_tiles + tix : tix, key_filter, &prm, focus, combos_l, rows_base,
      row_cursor, tile_row_off, tile_row_cnt, dev_err, wave_tot, &row_base, &scan_carry);
  }
}

#define PILEUP_PARAMS const MkpReadHdr* __restrict__ hdrs, const uint32_t* __restrict__ cigar, const uint8_t* __restrict__ seqs, \
                 const MkpEvent* __restrict__ events, const MkpReadOut* __restrict__ readout, const MkpTile* __restrict__ tiles, uint32_t n_tiles, \
                 const MkpRunParams* __restrict__ prmp, const uint32_t* __restrict__ slotbm, const uint8_t* __restrict__ focus, const MkpCombo* __restrict__ combos, \
                 uint32_t* __restrict__ rows_base /* 11 SoA arrays of row_capacity entries */, uint32_t* __restrict__ row_cursor, uint32_t* __restrict__ tile_row_off, uint32_t* __restrict__ tile_row_cnt, \
                 const uint2* __restrict__ chunk_pfx, uint32_t* __restrict__ dev_err, uint32_t key_arg /* KEYED kernels: partition key to tally | index of this key pass << 16 */
#define PILEUP_PASS hdrs, cigar, seqs, events, readout, tiles, n_tiles, prmp, slotbm, focus, combos, rows_base, row_cursor, tile_row_off, tile_row_cnt, chunk_pfx, dev_err, key_arg
// every position owns a tally column (no focus positions): the dense walk, 4 windows of 64 positions in flight
extern "C" __global__ void __launch_bounds__(PILEUP_THREADS, 8) mkp_pileup_tiles(PILEUP_PARAMS) { pileup_tiles_body<false, 4, false>(PILEUP_PASS); }
// (focus positions — --cpg / --motif / --include-bed — go through the slot pipeline of mkp_slots.hip; the FOCUS instantiation of this body
// serves pileup-hemi below)
// --partition-tag: the same kernel tallying only the reads of one partition key per launch
extern "C" __global__ void __launch_bounds__(PILEUP_THREADS, 8) mkp_pileup_tiles_keyed(PILEUP_PARAMS) {
  pileup_tiles_body<false, 4, true>(PILEUP_PASS); }
// pileup-hemi: the focus kernel with duplex pattern tallies
extern "C" __global__ void __launch_bounds__(PILEUP_THREADS, 8) mkp_pileup_tiles_hemi(PILEUP_PARAMS) {
  pileup_tiles_body<true, 1, false, true>(PILEUP_PASS); }

// Duplex reads decoded one group per wave (decode_read_sparse): interleave the two position-sorted event lists of a read into the
// front of its slice and combine the two summaries.  The record fails if either group failed (add_record returns at the first
// error, read_cache.rs:111-211) and counts as "no modified base information" if neither group added anything.
extern "C" __global__ void __launch_bounds__(256)
mkp_merge_duplex(const MkpReadHdr* __restrict__ hdrs, const uint32_t* __restrict__ read_ids, uint32_t n, const MkpTagRef* __restrict__ tagref,
    const MkpLayout* __restrict__ layouts,
                 MkpEvent* __restrict__ events, MkpReadOut* __restrict__ readout, uint32_t roff) {
  const uint32_t lane = (uint32_t)lane_id();
  const uint32_t widx = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (widx >= n) return;
  const uint32_t rid_raw = (uint32_t)__builtin_amdgcn_readfirstlane((int)read_ids[widx]);
  if (rid_raw >> 31) return;   // the listing of the second group
  const uint32_t rid = rid_raw;
  const MkpReadHdr h = hdrs[rid];
  const uint32_t n_first = layouts[h.layout].pad;
  const uint32_t capA = tagref[h.tag_off].n, capB = tagref[h.tag_off + n_first].n;
  const MkpReadOut a = readout[rid], b = readout[roff + rid];
  MkpReadOut out; out.n_events = 0; out.ok = 0; out.obs[0] = 0; out.obs[1] = 0;
  if (a.ok && b.ok && (a.ok == 1u || b.ok == 1u)) {
    const MkpEvent* __restrict__ eA = events + h.event_off + capA + capB; const MkpEvent* __restrict__ eB = eA + capA;
    MkpEvent* __restrict__ dst = events + h.event_off;
    const uint32_t nA = a.ok == 1u ? a.n_events : 0u, nB = b.ok == 1u ? b.n_events : 0u;
    // An event of the first group goes after the second group's events at lower positions, one of the second group after the first group's
    // events at lower or equal positions.  Both lists are sorted, so the ranks of 64 consecutive events of one list lie in a short window of
    // the other: the window's positions are taken 64 at a time, one per lane, and every lane finds its rank with six cross-lane steps
    // (round 5: a bisection over global memory per event — eight dependent loads; 2 x 0.33 ms per pass of the hemi workload).
    auto place = [&](const MkpEvent* __restrict__ X, uint32_t nX, const MkpEvent* __restrict__ Y, uint32_t nY, uint32_t le) {
      uint32_t y0 = 0;   // (uniform) every Y entry before y0 ranks below every X event still to come
      for (uint32_t i0 = 0; i0 < nX; i0 += 64u) {
        const uint32_t i = i0 + lane; const bool valid = i < nX;
        MkpEvent e; e.pos = 0xfffffffeu; e.info = 0; if (valid) e = X[i];
        const uint32_t key = e.pos + le;   // "<= pos" as "< pos + 1"
        uint32_t rank = y0;
        for (uint32_t yw = y0;; yw += 64u) {
          const uint32_t yv = yw + lane < nY ? Y[yw + lane].pos : 0xffffffffu;
          uint32_t c = (uint32_t)find_sorted(yv, key);         // entries of this window below the key (find_sorted stops at 63: the last lane's own
          if (c == 63u && (uint32_t)__shfl((int)yv, 63, 64) < key) c = 64u;   // entry is looked at here)
          const bool more = valid && rank == yw && c == 64u;   // the whole window ranks below this lane's key: the next one may too
          if (valid && rank == yw) rank += c;
          if (!__any(more) || yw + 64u >= nY) break;
        }
        if (valid) dst[i + rank] = e;
        y0 = (uint32_t)__builtin_amdgcn_readlane((int)rank, (int)(min(nX - i0, 64u) - 1u));
      }
    };
    place(eA, nA, eB, nB, 0u);
    place(eB, nB, eA, nA, 1u);
    out.ok = 1; out.n_events = nA + nB;
    out.obs[0] = (a.ok == 1u ? a.obs[0] : 0u) | (b.ok == 1u ? b.obs[0] : 0u);
      out.obs[1] = (a.ok == 1u ? a.obs[1] : 0u) | (b.ok == 1u ? b.obs[1] : 0u);
  }
  if (lane == 0) readout[rid] = out;
}

// pileup-hemi, records whose tags failed (readout.ok == 0): DuplexReadCache::get_duplex_mod_call (read_cache.rs:422-462) finds such a
// record in no map the first time it is asked about it, fails to add it, and answers NoCall(primary base) for that one position;
// from then on the record is in the skip set and yields no feature.  The cache lives for one interval (process_region_duplex,
// duplex.rs:241-339), so the record leaves one NoCall per interval it crosses: at its first '+' motif position there that it
// covers with an A/C/G/T base (not a deletion or ref-skip).  One thread per record writes those as events into the record's own
// (unused) event slice, where mkp_pileup_tiles_hemi picks them up.  iv_start = ascending starts of the shard's intervals.
extern "C" __global__ void __launch_bounds__(256)
mkp_hemi_failed_reads(const MkpReadHdr* __restrict__ hdrs, const uint32_t* __restrict__ cigar, const uint8_t* __restrict__ seqs,
    MkpEvent* __restrict__ events,
                      MkpReadOut* __restrict__ readout, uint32_t n_reads, const uint32_t* __restrict__ slotbm, const uint32_t* __restrict__ iv_start,
                          uint32_t n_iv,
                      int32_t win_start, int32_t win_end, uint32_t* __restrict__ dev_err) {
  const uint32_t rid = blockIdx.x * 256u + threadIdx.x;
  if (rid >= n_reads) return;
  if (readout[rid].ok) return;
  const MkpReadHdr h = hdrs[rid];
  const uint8_t* __restrict__ seq = seqs + h.seq_off;
  MkpEvent* __restrict__ ev = events + h.event_off;
  auto next_slot = [&](int32_t a, int32_t b) -> int32_t {   // first slot in [a, b), or -1
    uint32_t bit = (uint32_t)(a - (win_start - MKP_SLOTBM_MARGIN));
    const uint32_t end = (uint32_t)(b - (win_start - MKP_SLOTBM_MARGIN));
    uint32_t w = slotbm[bit >> 5] & ~((1u << (bit & 31u)) - 1u);
    for (;;) {
      if (w) { const uint32_t f = (bit & ~31u) + (uint32_t)__ffs((int)w) - 1u; return f < end ? (int32_t)f + (win_start - MKP_SLOTBM_MARGIN) : -1; }
      bit = (bit & ~31u) + 32u;
      if (bit >= end) return -1;
      w = slotbm[bit >> 5];
    }
  };
  uint32_t n = 0, q = 0; bool overflow = false;
  int32_t rp = h.ref_start, served = max(h.ref_start, win_start);   // positions below `served` lie in intervals that have their NoCall
  for (uint32_t c = 0; c < h.n_cigar && rp < win_end && served < win_end; c++) {
    const uint32_t w = cigar[h.cigar_off + c], op = w & 15u, len = w >> 4;
    if (op_is_match(op)) {
      int32_t a = max(rp, served); const int32_t b = min(rp + (int32_t)len, win_end);
      while (a < b) {
        const int32_t p = next_slot(a, b);
        if (p < 0) break;
        const uint32_t qq = q + (uint32_t)(p - rp);
        const int x = qq < h.l_seq ? nib2base(seq_nibble(seq, qq)) : -1;
        if (x < 0) { a = p + 1; continue; }
        if (n < h.event_cap) { ev[n].pos = (uint32_t)p; ev[n].info = (uint32_t)(MKP_H_NC + x); } else overflow = true;
        n++;
        uint32_t lo = 0, hi = n_iv;   // first interval starting after p
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (iv_start[mid] <= (uint32_t)p) lo = mid + 1u; else hi = mid; }
        served = lo < n_iv ? (int32_t)iv_start[lo] : win_end;
        a = served;
      }
    }
    if (op_consumes_query(op)) q += len;
    if (op_consumes_ref(op)) rp += (int32_t)len;
  }
  if (overflow) { atomicOr(dev_err, ERR_EVENT_CAP); n = 0; }
  readout[rid].n_events = n;
}

// ----------------------------------------------------------------------------------------------
// Order the per-tile row runs by tile index.  Block 0 computes the exclusive scan of the
// tile counts (n_tiles is small), then every block copies its tiles' runs.
extern "C" __global__ void __launch_bounds__(1024)
mkp_scan_tiles(const uint32_t* __restrict__ tile_row_cnt, uint32_t n_tiles, uint32_t* __restrict__ tile_dst_off, uint32_t* __restrict__ total_rows) {
  __shared__ uint32_t carry_s;
  __shared__ uint32_t wtot[16];
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < n_tiles; b0 += 1024) {
    const uint32_t k = b0 + threadIdx.x;
    const uint32_t v = k < n_tiles ? tile_row_cnt[k] : 0u;
    const uint32_t inc = wave_incl_scan(v);
    if (lane_id() == 63) wtot[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w2 = 0; w2 < (threadIdx.x >> 6); w2++) woff += wtot[w2];
    if (k < n_tiles) tile_dst_off[k] = carry_s + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s += woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_rows = carry_s;
}

extern "C" __global__ void __launch_bounds__(256)
mkp_gather_rows(const uint32_t* __restrict__ tile_row_off, const uint32_t* __restrict__ tile_row_cnt, const uint32_t* __restrict__ tile_dst_off,
                uint32_t n_tiles, MkpRowsDev src, MkpRowsDev dst) {
  for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint32_t n = tile_row_cnt[t], so = tile_row_off[t], d0 = tile_dst_off[t];
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
      dst.pos[d0 + k] = src.pos[so + k]; dst.info[d0 + k] = src.info[so + k]; dst.code[d0 + k] = src.code[so + k];
      dst.n_valid[d0 + k] = src.n_valid[so + k]; dst.n_mod[d0 + k] = src.n_mod[so + k]; dst.n_can[d0 + k] = src.n_can[so + k];
      dst.n_other[d0 + k] = src.n_other[so + k]; dst.n_del[d0 + k] = src.n_del[so + k]; dst.n_fail[d0 + k] = src.n_fail[so + k];
      dst.n_diff[d0 + k] = src.n_diff[so + k]; dst.n_nocall[d0 + k] = src.n_nocall[so + k];
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Threshold estimation (thresholds.rs:121-159) without shipping the sample to the host: the argmax probabilities the
// mkp_sample_* kernels wrote stay in HBM.  The f32 bit pattern of a probability is its sort key (positive floats order like
// unsigned integers); patterns are below 2^30, so the canonical base of a value rides in the top two bits.
//   mkp_sample_accumulate  for the reads the host's sampling schedule took: append {base, pattern} keys to the resident sample
//                          and count the top 16 bits of every pattern per base (level-0 histogram: LDS-privatised window of the
//                          patterns probabilities actually have, global atomics for the rest)
//   mkp_sample_hist1       level-1 histogram: the low 16 bits of the keys of one base whose top 16 bits equal `prefix`
// Two levels give the exact order statistics percentile_linear_interp needs (a 2 x 16 bit radix select); both histograms are
// plain integer arrays that add across GPUs (the path's one collective).
#define MKP_HIST_LO 0x3800u     // top-16 patterns [0x3800, 0x4000) = values in [2^-15, 2) are counted in LDS
#define MKP_HIST_LDS 2048u
extern "C" __global__ void __launch_bounds__(256)
mkp_sample_accumulate(const MkpReadHdr* __restrict__ hdrs, const MkpReadOut* __restrict__ readout, const uint8_t* __restrict__ take, uint32_t n_reads,
                      const float* __restrict__ vals, const MkpEvent* __restrict__ events, uint32_t* __restrict__ store, unsigned long long store_cap,
                      unsigned long long* __restrict__ store_cursor, uint32_t* __restrict__ hist0, uint32_t* __restrict__ dev_err) {
  __shared__ uint32_t lh[4][MKP_HIST_LDS];
  for (uint32_t k = threadIdx.x; k < 4u * MKP_HIST_LDS; k += 256) (&lh[0][0])[k] = 0;
  __syncthreads();
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
  for (uint32_t r = wave; r < n_reads; r += n_waves) {
    if (!take[r]) continue;
    const MkpReadOut ro = readout[r];
    if (!ro.ok || !ro.n_events) continue;
    const uint32_t e0 = hdrs[r].event_off;
    for (uint32_t k0 = 0; k0 < ro.n_events; k0 += 64) {
      const uint32_t k = k0 + (uint32_t)lane;
      const bool v = k < ro.n_events;
      uint32_t key = 0;
      if (v) { const uint32_t bits = __float_as_uint(vals[e0 + k]), base = events[e0 + k].info & 3u; key = (base << 30) | (bits & 0x3fffffffu);
               const uint32_t top = (bits >> 16) & 0x3fffu;
               if (top - MKP_HIST_LO < MKP_HIST_LDS) atomicAdd(&lh[base][top - MKP_HIST_LO], 1u); else atomicAdd(&hist0[base * 65536u + top], 1u); }
      const unsigned long long b = __ballot(v);
      unsigned long long at = 0;
      if (lane == 0) at = atomicAdd(store_cursor, (unsigned long long)__popcll(b));
      at = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(at >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)at);
      const unsigned long long idx = at + (unsigned long long)__popcll(b & lanemask_lt());
      if (v) { if (idx < store_cap) store[idx] = key; else atomicOr(dev_err, ERR_EVENT_CAP); }
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < 4u * MKP_HIST_LDS; k += 256) { const uint32_t c = (&lh[0][0])[k];
    if (c) atomicAdd(&hist0[(k / MKP_HIST_LDS) * 65536u + MKP_HIST_LO + (k % MKP_HIST_LDS)], c);
    }
}

extern "C" __global__ void __launch_bounds__(256)
mkp_sample_hist1(const uint32_t* __restrict__ store, unsigned long long n, uint32_t base, uint32_t prefix, uint32_t* __restrict__ hist1) {
  const uint32_t want = (base << 14) | (prefix & 0x3fffu);   // key >> 16
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256u) {
    const uint32_t key = store[i];
    if ((key >> 16) == want) atomicAdd(&hist1[key & 0xffffu], 1u);
  }
}

// `modkit summary`: counts of the sampled calls of the reads the schedule took.  table[base][0 pass | 1 filtered][class] (class as in
// summary_info; filtered calls are counted under their argmax class), reads_with[base] = taken reads with a call on that base,
// reads_with[4] = taken reads with any call, reads_with[5] = OR of the reads' observed-code slot masks.
extern "C" __global__ void __launch_bounds__(256)
mkp_summary_accumulate(const MkpReadHdr* __restrict__ hdrs, const MkpReadOut* __restrict__ readout, const uint8_t* __restrict__ take,
    uint32_t n_reads,
                       const MkpEvent* __restrict__ events, unsigned long long* __restrict__ table /*[4][2][16]*/,
                           unsigned long long* __restrict__ reads_with /*[6]*/) {
  __shared__ uint32_t lt[128];
  __shared__ uint32_t lr[6];
  for (uint32_t k = threadIdx.x; k < 128; k += 256) lt[k] = 0;
  if (threadIdx.x < 6) lr[threadIdx.x] = 0;
  __syncthreads();
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
  for (uint32_t r = wave; r < n_reads; r += n_waves) {
    if (!take[r]) continue;
    const MkpReadOut ro = readout[r];
    if (!ro.ok || !ro.n_events) continue;
    const uint32_t e0 = hdrs[r].event_off;
    uint32_t bases = 0;
    for (uint32_t k0 = 0; k0 < ro.n_events; k0 += 64) {
      const uint32_t k = k0 + (uint32_t)lane;
      if (k < ro.n_events) {
        const uint32_t info = events[e0 + k].info, tb = info & 3u, thr = (info >> 4) & 15u, arg = (info >> 8) & 15u;
        atomicAdd(&lt[tb * 32u + (thr ? thr : 16u + arg)], 1u);
        bases |= 1u << tb;
      }
    }
    bases = wave_or(bases);
    if (lane == 0) { for (uint32_t b = 0; b < 4; b++) if (bases & (1u << b)) atomicAdd(&lr[b], 1u); atomicAdd(&lr[4], 1u);
      atomicOr(&lr[5], ro.obs[0] | ro.obs[1]); }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < 128; k += 256) if (lt[k]) atomicAdd(&table[k], (unsigned long long)lt[k]);
  if (threadIdx.x < 5 && lr[threadIdx.x]) atomicAdd(&reads_with[threadIdx.x], (unsigned long long)lr[threadIdx.x]);
  if (threadIdx.x == 5 && lr[5]) atomicOr(&reads_with[5], (unsigned long long)lr[5]);
}
extern "C" hipError_t mkp_launch_summary_accumulate(hipStream_t st, const MkpReadHdr* hdrs, const MkpReadOut* readout, const uint8_t* take,
    uint32_t n_reads, const MkpEvent* events,
                                                    unsigned long long* table, unsigned long long* reads_with) {
  if (!n_reads) return hipSuccess;
  const uint32_t grid = std::min<uint32_t>((n_reads + 3u) / 4u, 1024u);
  hipLaunchKernelGGL(mkp_summary_accumulate, dim3(grid), dim3(256), 0, st, hdrs, readout, take, n_reads, events, table, reads_with);
  return hipGetLastError();
}

extern "C" hipError_t mkp_launch_sample_accumulate(hipStream_t st, const MkpReadHdr* hdrs, const MkpReadOut* readout, const uint8_t* take,
    uint32_t n_reads, const float* vals,
                                                   const MkpEvent* events, uint32_t* store, unsigned long long store_cap,
                                                       unsigned long long* store_cursor, uint32_t* hist0, uint32_t* dev_err) {
  if (!n_reads) return hipSuccess;
  const uint32_t grid = std::min<uint32_t>((n_reads + 3u) / 4u, 1024u);
  hipLaunchKernelGGL(mkp_sample_accumulate, dim3(grid), dim3(256), 0, st, hdrs, readout, take, n_reads, vals, events, store, store_cap, store_cursor,
      hist0, dev_err);
  return hipGetLastError();
}
extern "C" hipError_t mkp_launch_sample_hist1(hipStream_t st, const uint32_t* store, unsigned long long n, uint32_t base, uint32_t prefix,
    uint32_t* hist1) {
  if (!n) return hipSuccess;
  const uint32_t grid = (uint32_t)std::min<unsigned long long>((n + 255u) / 256u, 4096ull);
  hipLaunchKernelGGL(mkp_sample_hist1, dim3(grid), dim3(256), 0, st, store, n, base, prefix, hist1);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------
// host-side launchers (called from mkp_api.cpp)
// read_ids = [SPARSE one tag | SPARSE two tags | FAST one tag | FAST two tags | all other reads], n_class = the five list lengths
extern "C" hipError_t mkp_launch_decode(hipStream_t st, const MkpReadHdr* hdrs, const uint32_t* read_ids, const uint32_t* n_class /* [7] */,
    const uint32_t* cigar, const uint8_t* seqs,
                                        const MkpTagRef* tagref, const uint32_t* ranks, const uint8_t* ml, const MkpLayout* layouts,
                                        const MkpRunParams* prm, MkpEvent* events, MkpReadOut* readout, uint32_t* dev_err,
                                        const uint8_t* bedmask, float* sample_vals) {
  const uint32_t waves_per_block = 4;
  const uint32_t* ids = read_ids;
  for (int cls = 0; cls < 7; cls++) {
    const uint32_t n = n_class[cls];
    if (n) {
      dim3 grid((n + waves_per_block - 1) / waves_per_block), block(64 * waves_per_block);
      if (cls >= 5) {   // duplex reads, listed once per group (never in sampling mode): SPARSE decode per group, then the merge
        if (prm->sample_mode) return hipErrorInvalidValue;
        if (cls == 5) hipLaunchKernelGGL(mkp_decode_sparse1, grid, block, 0, st, hdrs, n, cigar, seqs, tagref, ranks, ml, layouts, *prm, events,
            readout, dev_err, bedmask, sample_vals, ids);
        else hipLaunchKernelGGL(mkp_decode_sparse2, grid, block, 0, st, hdrs, n, cigar, seqs, tagref, ranks, ml, layouts, *prm, events, readout,
            dev_err, bedmask, sample_vals, ids);
        hipLaunchKernelGGL(mkp_merge_duplex, grid, block, 0, st, hdrs, ids, n, tagref, layouts, events, readout, prm->readout_b_off);
        ids += n;
        continue;
      }
#define MKP_DECODE_LAUNCH(K) hipLaunchKernelGGL(K, grid, block, 0, st, hdrs, n, cigar, seqs, tagref, ranks, ml, layouts, *prm, events, readout, dev_err, bedmask, sample_vals, ids)
      if (prm->sample_mode) { if (cls == 0) MKP_DECODE_LAUNCH(mkp_sample_sparse1); else if (cls == 1) MKP_DECODE_LAUNCH(mkp_sample_sparse2);
        else if (cls == 2) MKP_DECODE_LAUNCH(mkp_sample_fast1);
        else if (cls == 3) MKP_DECODE_LAUNCH(mkp_sample_fast2);
        else MKP_DECODE_LAUNCH(mkp_sample_reads);
        }
      else { if (cls == 0) MKP_DECODE_LAUNCH(mkp_decode_sparse1); else if (cls == 1) MKP_DECODE_LAUNCH(mkp_decode_sparse2);
        else if (cls == 2) MKP_DECODE_LAUNCH(mkp_decode_fast1);
        else if (cls == 3) MKP_DECODE_LAUNCH(mkp_decode_fast2);
        else MKP_DECODE_LAUNCH(mkp_decode_reads);
        }
    }
    ids += n;
  }
  return hipGetLastError();
}

// per device: both accumulate kernels may use the whole per-workgroup LDS budget the host planned for
extern "C" hipError_t mkp_pileup_set_lds(uint32_t accum_bytes) {
  for (const void* k : {(const void*)mkp_pileup_tiles, (const void*)mkp_pileup_tiles_keyed, (const void*)mkp_pileup_tiles_hemi}) {
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)accum_bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

extern "C" hipError_t mkp_launch_pileup(hipStream_t st, uint32_t lds_bytes, int focus_mode, const MkpReadHdr* hdrs, const uint32_t* cigar,
    const uint8_t* seqs,
                                        const MkpEvent* events, const MkpReadOut* readout, const MkpTile* tiles, uint32_t n_tiles,
                                            const MkpRunParams* prm_dev,
                                        const uint32_t* slotbm, const uint8_t* focus, const MkpCombo* combos, const MkpRowsDev* rows,
                                            uint32_t* row_cursor,
                                        uint32_t* tile_row_off, uint32_t* tile_row_cnt, const uint32_t* chunk_pfx, uint32_t* dev_err,
                                            uint32_t key_filter, uint32_t key_slot) {
  if (!n_tiles) return hipSuccess;
  const bool keyed = key_filter != MKP_NO_KEY_FILTER;
  const uint32_t key_arg = keyed ? ((key_filter & 0xffffu) | (key_slot << 16)) : 0u;
  const uint32_t grid = n_tiles;   // one workgroup per tile
  // ONE build of each accumulate kernel: what a one-shot shard pass launches is what a re-launch on the resident shard launches
#define MKP_PILEUP_LAUNCH(K) hipLaunchKernelGGL(K, dim3(grid), dim3(PILEUP_THREADS), lds_bytes, st, hdrs, cigar, seqs, events, readout, tiles, n_tiles, prm_dev, slotbm, focus, combos, rows->pos, \
                       row_cursor, tile_row_off, tile_row_cnt, reinterpret_cast<const uint2*>(chunk_pfx), dev_err, key_arg)
  if (focus_mode == 2) MKP_PILEUP_LAUNCH(mkp_pileup_tiles_hemi);   // pileup-hemi
  else if (focus_mode) return hipErrorInvalidValue;   // (focus runs are the slot pipeline's: mkp_launch_stream)
  else { if (keyed) MKP_PILEUP_LAUNCH(mkp_pileup_tiles_keyed); else MKP_PILEUP_LAUNCH(mkp_pileup_tiles); }
  return hipGetLastError();
}

extern "C" hipError_t mkp_launch_hemi_failed(hipStream_t st, const MkpReadHdr* hdrs, const uint32_t* cigar, const uint8_t* seqs, MkpEvent* events,
    MkpReadOut* readout, uint32_t n_reads,
                                             const uint32_t* slotbm, const uint32_t* iv_start, uint32_t n_iv, int32_t win_start, int32_t win_end,
                                                 uint32_t* dev_err) {
  if (!n_reads) return hipSuccess;
  hipLaunchKernelGGL(mkp_hemi_failed_reads, dim3((n_reads + 255u) / 256u), dim3(256), 0, st, hdrs, cigar, seqs, events, readout, n_reads, slotbm,
      iv_start, n_iv, win_start, win_end, dev_err);
  return hipGetLastError();
}

extern "C" hipError_t mkp_launch_gather(hipStream_t st, const uint32_t* tile_row_off, const uint32_t* tile_row_cnt, uint32_t* tile_dst_off,
                                        uint32_t n_tiles, uint32_t* total_rows, const MkpRowsDev* src, const MkpRowsDev* dst) {
  hipLaunchKernelGGL(mkp_scan_tiles, dim3(1), dim3(1024), 0, st, tile_row_cnt, n_tiles, tile_dst_off, total_rows);
  if (n_tiles) {
    uint32_t grid = n_tiles < 2048u ? n_tiles : 2048u;
    hipLaunchKernelGGL(mkp_gather_rows, dim3(grid), dim3(256), 0, st, tile_row_off, tile_row_cnt, tile_dst_off, n_tiles, *src, *dst);
  }
  return hipGetLastError();
}
