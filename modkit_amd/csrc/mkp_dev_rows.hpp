// Row emission from a tile's LDS tallies (FeatureVector::decode / add_tally_to_counts / combine_strand_features, pileup/mod.rs:283-561;
// DuplexFeatureVector::decode, duplex.rs:124-205), shared by the accumulate kernels of mkp_kernels.hip and mkp_slots.hip.
#pragma once
#include "mkp_dev_common.hpp"

struct RowAcc { uint32_t n_valid, n_mod, n_can, n_other, n_del, n_fail, n_diff, n_nocall; };

// packed tallies of a tile in LDS: [counter | observed-code slot][S], '+' tally in the low, '-' in the high 16 bits; column = tally slot
struct TileView {
  const uint32_t* pk;
  uint32_t W, n_counters;
  __device__ __forceinline__ uint32_t c(uint32_t s, uint32_t cid, uint32_t i) const { return (pk[cid * W + i] >> (16u * s)) & 0xffffu; }
  __device__ __forceinline__ int32_t o(uint32_t s, uint32_t sl, uint32_t i) const {
    return (int32_t)((pk[(n_counters + sl) * W + i] >> (16u * s)) & 0xffffu); }
};

// one (strand tally, primary base) row of FeatureVector::add_tally_to_counts (pileup/mod.rs:283-410);
// sl < 0: --combine-mods row (code = base letter).  Returns false when the reference emits nothing.
// FILL = false: only decide whether the row exists (the counting pass)
template <bool FILL, class TV>
__device__ __forceinline__ bool tally_row(const TV& tv, const MkpRunParams& prm, uint32_t s, uint32_t i, int sl, int pb, RowAcc* r) {
  const uint32_t ck = prm.can_of_pb[pb];
  if (ck == 0xffu) return false;
  uint32_t n_can = tv.c(s, MKP_C_CAN + ck, i), mods = 0;
  for (uint32_t t = 0; t < prm.n_slots; t++) if (prm.slots[t].pb == pb) mods += tv.c(s, prm.slots[t].cid, i);
  const uint32_t cov = n_can + mods;
  if (cov == 0) return false;
  uint32_t n_mod;
  if (sl >= 0) { if (tv.o(s, (uint32_t)sl, i) <= 0) return false; n_mod = tv.c(s, prm.slots[sl].cid, i); }
  else n_mod = mods;
  if (!FILL) return true;
  uint32_t total = 0;
  for (uint32_t k = 0; k < prm.n_counters; k++) if (k != MKP_C_DEL && k != MKP_C_FAIL) total += tv.c(s, k, i);
  const uint32_t nocall = tv.c(s, MKP_C_NC + pb, i);
  r->n_valid = cov; r->n_mod = n_mod; r->n_can = n_can; r->n_other = sl >= 0 ? mods - n_mod : 0u;
  r->n_del = tv.c(s, MKP_C_DEL, i); r->n_fail = tv.c(s, MKP_C_FAIL, i);
  r->n_diff = total - (nocall + cov); r->n_nocall = nocall;
  return true;
}

// Rows of one reference position p whose tallies sit in column i.  slot_of(q) = column of a nearby position q (the strand-
// combining partner: always a focus position of the same tile's halo'd range).
template <bool WRITE, class TV, class SlotOf>
__device__ __forceinline__ uint32_t rows_at(const TV& tv, const MkpRunParams& prm, const uint8_t* __restrict__ focus,
                                            const MkpCombo* combos, int32_t p, uint32_t i, const MkpRowsDev& rows,
                                            uint32_t wr, uint32_t fv /* this position's focus byte (3 when there is no focus) */, SlotOf slot_of,
                                                uint32_t key = 0) {
  const uint32_t rule = fv & 3u, combo = fv >> 2;
  if (!rule) return 0;
  uint32_t n = 0;
  auto put = [&](const RowAcc& r, uint32_t strand, uint32_t code, int motif) {
    if (WRITE) {
      uint32_t k = wr + n;
      rows.pos[k] = (uint32_t)p; rows.info[k] = strand | ((uint32_t)(motif + 1) << 8) | (key << 16); rows.code[k] = code;
      rows.n_valid[k] = r.n_valid; rows.n_mod[k] = r.n_mod; rows.n_can[k] = r.n_can; rows.n_other[k] = r.n_other;
      rows.n_del[k] = r.n_del; rows.n_fail[k] = r.n_fail; rows.n_diff[k] = r.n_diff; rows.n_nocall[k] = r.n_nocall;
    }
    n++;
  };
  static const char LETTER[4] = {'A', 'C', 'G', 'T'};
  if (!prm.combine_strands) {
    for (uint32_t s = 0; s < 2; s++) {
      if (!((rule >> s) & 1u)) continue;
      uint32_t n_ids = 0; const uint8_t* ids = nullptr;
      if (combo) { n_ids = s ? combos[combo].n_neg : combos[combo].n_pos; ids = s ? combos[combo].neg_ids : combos[combo].pos_ids; }
      RowAcc r;
      if (prm.numeric_mode == 1) {
        for (int pb = 0; pb < 4; pb++) if (tally_row<WRITE>(tv, prm, s, i, -1, pb, &r)) {
          if (n_ids) for (uint32_t k = 0; k < n_ids; k++) put(r, s, (uint32_t)LETTER[pb], ids[k]); else put(r, s, (uint32_t)LETTER[pb], -1);
        }
      } else {
        for (uint32_t oi = 0; oi < prm.n_slots; oi++) {
          const int sl = prm.slot_order[oi];
          if (tally_row<WRITE>(tv, prm, s, i, sl, prm.slots[sl].pb, &r)) {
            if (n_ids) for (uint32_t k = 0; k < n_ids; k++) put(r, s, prm.slots[sl].code_repr, ids[k]); else put(r, s, prm.slots[sl].code_repr, -1);
          }
        }
      }
    }
  } else {
    // combine_strand_features (pileup/mod.rs:469-561): only '+' motif positions produce rows
    if (!combo) return 0;
    const MkpCombo& cb = combos[combo];
    for (uint32_t m = 0; m < cb.n_pos; m++) {
      const int idx = cb.pos_ids[m];
      const int delta = cb.pos_delta[m];
      if (delta == -128) continue;  // not a palindrome / negative_strand_position() == None
      const int32_t qpos = p + delta;
      bool neg_ok = false; uint32_t iq = 0;
      if (delta != -127 && qpos >= prm.win_start && qpos < prm.win_end) {  // -127: mate position is in another interval
        uint32_t fq = focus[qpos - prm.win_start];
        if ((fq & 2u) && (fq >> 2)) {
          const MkpCombo& cq = combos[fq >> 2];
          for (uint32_t k = 0; k < cq.n_neg; k++) neg_ok |= (cq.neg_ids[k] == idx);
        }
        if (neg_ok) iq = slot_of(qpos);
      }
      const bool pos_ok = (rule & 1u) != 0;
      auto add = [](RowAcc& a, const RowAcc& b) { a.n_valid += b.n_valid; a.n_mod += b.n_mod; a.n_can += b.n_can; a.n_other += b.n_other;
                                                  a.n_del += b.n_del; a.n_fail += b.n_fail; a.n_diff += b.n_diff; a.n_nocall += b.n_nocall; };
      if (prm.numeric_mode == 1) {
        for (int pb = 0; pb < 4; pb++) {
          RowAcc acc = {0, 0, 0, 0, 0, 0, 0, 0}, r; bool any = false;
          if (pos_ok && tally_row<WRITE>(tv, prm, 0, i, -1, pb, &r)) { add(acc, r); any = true; }
          if (neg_ok && tally_row<WRITE>(tv, prm, 1, iq, -1, pb, &r)) { add(acc, r); any = true; }
          if (any) put(acc, 2, (uint32_t)LETTER[pb], idx);
        }
      } else {
        for (uint32_t oi = 0; oi < prm.n_slots; oi++) {
          const uint32_t code = prm.slots[prm.slot_order[oi]].code_repr;
          if (oi && prm.slots[prm.slot_order[oi - 1]].code_repr == code) continue;  // grouped by code (BTreeMap)
          RowAcc acc = {0, 0, 0, 0, 0, 0, 0, 0}, r; bool any = false;
          for (uint32_t oj = oi; oj < prm.n_slots && prm.slots[prm.slot_order[oj]].code_repr == code; oj++) {
            const int sl = prm.slot_order[oj];
            if (pos_ok && tally_row<WRITE>(tv, prm, 0, i, sl, prm.slots[sl].pb, &r)) { add(acc, r); any = true; }
          }
          for (uint32_t oj = oi; oj < prm.n_slots && prm.slots[prm.slot_order[oj]].code_repr == code; oj++) {
            const int sl = prm.slot_order[oj];
            if (neg_ok && tally_row<WRITE>(tv, prm, 1, iq, sl, prm.slots[sl].pb, &r)) { add(acc, r); any = true; }
          }
          if (any) put(acc, 2, code, idx);
        }
      }
    }
  }
  return n;
}

// pileup-hemi rows of one '+' motif position whose counters sit in column i (DuplexFeatureVector::decode, duplex.rs:124-205, in the
// writer's order: primary base, then pattern — writers.rs:196-207).  The pattern goes out as its two element indices
// (rows.code = a | b << 8), the primary base in rows.info; the host turns the elements into mod codes.
template <bool WRITE>
__device__ __forceinline__ uint32_t hemi_rows_at(const uint32_t* __restrict__ tal, uint32_t S, const MkpRunParams& prm, int32_t p, uint32_t i,
    const MkpRowsDev& rows, uint32_t wr) {
  uint32_t tot[4], n = 0;
  for (int pb = 0; pb < 4; pb++) {
    tot[pb] = 0;
    const uint32_t base = prm.hemi_pat_base[pb], nel = prm.hemi_nel[pb];
    if (base != 0xffu) for (uint32_t k = 0; k < nel * nel; k++) tot[pb] += tal[(base + k) * S + i];
  }
  for (int pb = 0; pb < 4; pb++) {
    if (!tot[pb]) continue;
    const uint32_t base = prm.hemi_pat_base[pb], nel = prm.hemi_nel[pb];
    for (uint32_t k = 0; k < nel * nel; k++) {
      const uint32_t cnt = tal[(base + k) * S + i];
      if (!cnt) continue;
      if (WRITE) {
        const uint32_t r = wr + n;
        rows.pos[r] = (uint32_t)p; rows.info[r] = (uint32_t)pb; rows.code[r] = (k / nel) | ((k % nel) << 8);
        rows.n_valid[r] = tot[pb]; rows.n_mod[r] = cnt; rows.n_can[r] = tal[base * S + i]; rows.n_other[r] = tot[pb] - cnt;
        rows.n_del[r] = tal[MKP_H_DEL * S + i]; rows.n_fail[r] = tal[(MKP_H_FAIL + pb) * S + i];
        rows.n_diff[r] = tot[0] + tot[1] + tot[2] + tot[3] - tot[pb]; rows.n_nocall[r] = tal[(MKP_H_NC + pb) * S + i];
      }
      n++;
    }
  }
  return n;
}

// ----------------------------------------------------------------------------------------------------------------------
// Row emission, row-major (round 6): MkpRunParams resolved once per workgroup into a small table (the "row program"), existence of a
// row decided from a handful of tallies, one thread per ROW filling and storing it (mkp_pileup_stream, mkp_pileup_tiles).
struct StreamProg {
  // row candidates per strand (or per motif when strands combine): observed-code slots in row order, or the four primary bases (--combine-mods)
  uint32_t n_groups;
  uint32_t totmask;        // counters that add up to a column's total (all but Delete and Filtered)
  uint32_t modmask[4];     // primary base -> the counters of its mod codes
  uint32_t code[16];       // group -> code of its rows
  // group -> [0:1] primary base, [2:6] observed-code slot + 1 (0: a --combine-mods row), [7:11] counter of the code, [12:16] counter of
  uint32_t info[16];
                           //          Canonical(base), [17] the base has one, [18] first group of its code (strand combining adds up the groups of a
                           //          code)
};

__device__ __forceinline__ uint32_t col_get(const uint32_t* __restrict__ tal, uint32_t S, uint32_t i, uint32_t s, uint32_t k) {
  return (tal[k * S + i] >> (16u * s)) & 0xffffu; }
__device__ __forceinline__ uint32_t col_sum(const uint32_t* __restrict__ tal, uint32_t S, uint32_t i, uint32_t s, uint32_t mask) {
  uint32_t t = 0;
  while (mask) { const uint32_t k = (uint32_t)__ffs((int)mask) - 1u; mask &= mask - 1u; t += col_get(tal, S, i, s, k); }
  return t;
}
// does (strand tally s, column i, group g) yield a row — add_tally_to_counts's early returns (pileup/mod.rs:283-410): the primary base has
// filtered coverage, and (per-code rows) the code was observed in a record over this column
__device__ __forceinline__ bool stream_row_exists(const uint32_t* __restrict__ tal, uint32_t S, uint32_t n_counters, const StreamProg& P, uint32_t s,
    uint32_t i, uint32_t g) {
  const uint32_t inf = P.info[g];
  if (!((inf >> 17) & 1u)) return false;
  const uint32_t cov = col_get(tal, S, i, s, (inf >> 12) & 31u) + col_sum(tal, S, i, s, P.modmask[inf & 3u]);
  if (!cov) return false;
  const uint32_t osl = (inf >> 2) & 31u;
  return !osl || col_get(tal, S, i, s, n_counters + osl - 1u) != 0u;
}
// the row itself, added into `r`
__device__ __forceinline__ void stream_row_add(const uint32_t* __restrict__ tal, uint32_t S, const StreamProg& P, uint32_t s, uint32_t i, uint32_t g,
    RowAcc& r) {
  const uint32_t inf = P.info[g], pb = inf & 3u;
  const uint32_t n_can = col_get(tal, S, i, s, (inf >> 12) & 31u), mods = col_sum(tal, S, i, s, P.modmask[pb]);
  const uint32_t n_mod = ((inf >> 2) & 31u) ? col_get(tal, S, i, s, (inf >> 7) & 31u) : mods;
  const uint32_t total = col_sum(tal, S, i, s, P.totmask), nocall = col_get(tal, S, i, s, MKP_C_NC + pb), cov = n_can + mods;
  r.n_valid += cov; r.n_mod += n_mod; r.n_can += n_can; r.n_other += mods - n_mod;
  r.n_del += col_get(tal, S, i, s, MKP_C_DEL); r.n_fail += col_get(tal, S, i, s, MKP_C_FAIL);
  r.n_diff += total - (nocall + cov); r.n_nocall += nocall;
}

// the row program, built by threads 0..15 of the workgroup (callers put a barrier before its first use)
__device__ __forceinline__ void rowprog_build(const MkpRunParams& prm, uint32_t n_counters, StreamProg& prog) {
  if (threadIdx.x >= 16u) return;
  const uint32_t g = threadIdx.x, combine_mods = prm.numeric_mode == 1 ? 1u : 0u;
  const uint32_t ng = combine_mods ? 4u : prm.n_slots;
  uint32_t code = 0, inf = 0;
  if (g < ng) {
    const uint32_t sl = combine_mods ? 0u : prm.slot_order[g], pb = combine_mods ? g : prm.slots[sl].pb, ck = prm.can_of_pb[pb];
    code = combine_mods ? (uint32_t)"ACGT"[g] : prm.slots[sl].code_repr;
    const bool first = combine_mods || g == 0u || prm.slots[prm.slot_order[g - 1u]].code_repr != code;
    inf = pb | ((combine_mods ? 0u : sl + 1u) << 2) | ((combine_mods ? 0u
        : (uint32_t)prm.slots[sl].cid) << 7) | (((MKP_C_CAN + ck) & 31u) << 12) | ((ck != 0xffu ? 1u : 0u) << 17) | ((first ? 1u : 0u) << 18);
  }
  prog.code[g] = code; prog.info[g] = inf;
  if (g < 4u) { uint32_t m = 0; for (uint32_t t = 0; t < prm.n_slots; t++) if (prm.slots[t].pb == g) m |= 1u << prm.slots[t].cid; prog.modmask[g] = m;
    }
  if (g == 0u) { prog.n_groups = ng; prog.totmask = ((1u << n_counters) - 1u) & ~((1u << MKP_C_DEL) | (1u << MKP_C_FAIL)); }
}

#define PILEUP_THREADS MKP_PILEUP_THREADS
#define PILEUP_WAVES (PILEUP_THREADS / 64)
#define PILEUP_WAVE_SCRATCH MKP_PILEUP_WAVE_SCRATCH

// Rows of a DENSE tile (mkp_pileup_tiles without focus positions: every position of the tile's range owns a column, both strands of every
// position are candidates, strands never combine): each thread takes a contiguous run of columns — rows keep position order — counts its
// rows, the block scans, thread 0 reserves the tile's run of the row buffer (mkp_scan_tiles + mkp_gather_rows order the runs afterwards),
// the threads scatter (column, strand, group) words into `rowmap` (LDS, `map_words` dwords: the accumulate phase's per-wave scratch, dead
// by now) and then every thread fills and stores whole rows.  Existence is evaluated twice (count, scatter) — a handful of LDS reads —
// instead of keeping a mask per column.  (Rounds 1-5: the interpreter of rows_at three times per column, 119 spilled registers.)
__device__ __forceinline__ void emit_dense_rows(const uint32_t* __restrict__ tal, uint32_t S, uint32_t n_counters, uint32_t n_tslots, int32_t T0h,
    const MkpTile& tl, uint32_t run, uint32_t key,
                                                const MkpRunParams& prm, StreamProg& prog, uint32_t* __restrict__ rowmap, uint32_t map_words,
                                                    uint32_t* __restrict__ rows_base,
                                                uint32_t* __restrict__ row_cursor, uint32_t* __restrict__ tile_row_off,
                                                    uint32_t* __restrict__ tile_row_cnt, uint32_t* __restrict__ dev_err,
                                                uint32_t* wave_tot, uint32_t* row_base_p, uint32_t* row_total_p) {
  const int lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  rowprog_build(prm, n_counters, prog);
  __syncthreads();
  const StreamProg& P = prog;
  // (strands combine at motif positions only: a run without focus positions has none)
  const uint32_t n_groups = prm.combine_strands ? 0u : P.n_groups;
  const uint32_t per = (n_tslots + PILEUP_THREADS - 1u) / PILEUP_THREADS;
  const uint32_t i0 = min(n_tslots, threadIdx.x * per), i1 = min(n_tslots, i0 + per);
  auto in_rows = [&](uint32_t i) { const int32_t p = T0h + (int32_t)i; return p >= tl.r0 && p < tl.r1; };
  uint32_t cnt = 0;
  for (uint32_t i = i0; i < i1; i++) {
    if (!in_rows(i)) continue;
    for (uint32_t s = 0; s < 2; s++) for (uint32_t g = 0; g < n_groups; g++) cnt += stream_row_exists(tal, S, n_counters, P, s, i, g) ? 1u : 0u;
  }
  const uint32_t inc2 = wave_incl_scan(cnt);
  if (lane == 63) wave_tot[wave] = inc2;
  __syncthreads();
  uint32_t off = inc2 - cnt, tile_rows = 0;
  for (uint32_t w2 = 0; w2 < PILEUP_WAVES; w2++) { const uint32_t t = wave_tot[w2]; if (w2 < wave) off += t; tile_rows += t; }
  if (threadIdx.x == 0) {
    uint32_t s = tile_rows;
    const uint32_t base = s ? atomicAdd(row_cursor, s) : 0u;
    if (base + s > prm.row_capacity) { atomicOr(dev_err, ERR_ROW_CAP); s = 0; }
    *row_base_p = base; *row_total_p = s; tile_row_off[run] = base; tile_row_cnt[run] = s;
  }
  MkpRowsDev rows;
  { const size_t cap = prm.row_capacity; uint32_t* q = rows_base;
    rows.pos = q; rows.info = q + cap; rows.code = q + 2 * cap; rows.n_valid = q + 3 * cap; rows.n_mod = q + 4 * cap; rows.n_can = q + 5 * cap;
      rows.n_other = q + 6 * cap;
    rows.n_del = q + 7 * cap; rows.n_fail = q + 8 * cap; rows.n_diff = q + 9 * cap; rows.n_nocall = q + 10 * cap; }
  for (uint32_t r0 = 0; r0 < tile_rows; r0 += map_words) {
    if (r0) __syncthreads();   // the round before has read the map
    if (cnt && off < r0 + map_words && off + cnt > r0) {
      uint32_t r = off;
      for (uint32_t i = i0; i < i1; i++) {
        if (!in_rows(i)) continue;
        for (uint32_t s = 0; s < 2; s++) for (uint32_t g = 0; g < n_groups; g++) {
          if (!stream_row_exists(tal, S, n_counters, P, s, i, g)) continue;
          if (r >= r0 && r < r0 + map_words) rowmap[r - r0] = i | (g << 13) | (s << 17);
          r++;
        }
      }
    }
    __syncthreads();
    const uint32_t total = *row_total_p, n_here = min(total, r0 + map_words) > r0 ? min(total, r0 + map_words) - r0 : 0u;
    for (uint32_t rr = threadIdx.x; rr < n_here; rr += PILEUP_THREADS) {
      const uint32_t e = rowmap[rr], si = e & 8191u, g = (e >> 13) & 15u, s = (e >> 17) & 1u;
      RowAcc acc = {0, 0, 0, 0, 0, 0, 0, 0};
      stream_row_add(tal, S, P, s, si, g, acc);
      const size_t at = (size_t)*row_base_p + r0 + rr;
      rows.pos[at] = (uint32_t)(T0h + (int32_t)si); rows.info[at] = s | (key << 16); rows.code[at] = P.code[g];   // (no motif: info[8:15] = 0)
      rows.n_valid[at] = acc.n_valid; rows.n_mod[at] = acc.n_mod; rows.n_can[at] = acc.n_can; rows.n_other[at] = acc.n_other;
      rows.n_del[at] = acc.n_del; rows.n_fail[at] = acc.n_fail; rows.n_diff[at] = acc.n_diff; rows.n_nocall[at] = acc.n_nocall;
    }
  }
}

// LDS byte addresses as integers: a tally update is then `lane base + 256*window + row*4*S`, two VALU instructions
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ void lds_add(uint32_t a, uint32_t v) {
  __hip_atomic_fetch_add((lds_u32*)(uintptr_t)a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// first index in the position-sorted event list `ev[0..n)` whose pos is >= key: 64-way probes, two dependent
// loads for up to 4096 events instead of a 12-step bisection
__device__ __forceinline__ uint32_t event_lower_bound(const MkpEvent* __restrict__ ev, uint32_t n, int32_t key) {
  const uint32_t lane = (uint32_t)lane_id();
  uint32_t lo = 0, span = n;
  while (span > 64) {
    const uint32_t stride = (span + 63u) >> 6;
    const uint32_t k = lo + lane * stride;
    const bool valid = k < lo + span;
    const int32_t p = valid ? (int32_t)ev[k].pos : 0x7fffffff;
    const uint32_t c = (uint32_t)__popcll(__ballot(valid && p < key));
    if (c == 0) return lo;
    const uint32_t nlo = lo + (c - 1u) * stride + 1u;
    const uint32_t nhi = min(lo + c * stride, lo + span);
    lo = nlo; span = nhi - nlo;
  }
  const bool valid = lane < span;
  const int32_t p = valid ? (int32_t)ev[lo + lane].pos : 0x7fffffff;
  return lo + (uint32_t)__popcll(__ballot(valid && p < key));
}

// Position <-> tally column of a tile.  Without focus positions every position of the tile's (halo'd) range owns a column.
// With them only the focus positions do: a bitmap of the range plus its running popcount (both in LDS) give
// rank(p) = number of focus positions below p, which is the column of p when p is one, and the first column at or after p
// when it is not (what the ends of a read span, a deletion or a CIGAR op need).
template <bool FOCUS> struct SlotMap {
  const uint32_t* bm; const uint32_t* pfx; const int32_t* fpos; int32_t lbase, T0h;
  __device__ __forceinline__ uint32_t rank(int32_t p) const {
    if (!FOCUS) return (uint32_t)(p - T0h);
    const uint32_t lb = (uint32_t)(p - lbase), w = lb >> 5;
    return pfx[w] + (uint32_t)__popc(bm[w] & ((1u << (lb & 31u)) - 1u));
  }
  __device__ __forceinline__ bool is_slot(int32_t p) const { if (!FOCUS) return true; const uint32_t lb = (uint32_t)(p - lbase);
    return (bm[lb >> 5] >> (lb & 31u)) & 1u; }
  __device__ __forceinline__ int32_t pos_of(uint32_t c) const { return FOCUS ? fpos[c] : T0h + (int32_t)c; }
};

// Row emission of one tile from its LDS tallies (the tail of mkp_pileup_tiles): count the rows of every slot, reserve the tile's
// run in the row buffer with one atomic, write.  Slot order = position order, so a block scan of the per-slot counts keeps it.
// Row runs in genome order without a second pass: a run's place in the row buffer is the number of rows of the runs before it, found by
// a decoupled look-back over one 64-bit word per run (bits 62-63: 1 = this run's own count is known, 2 = the count of everything up to
// and including it; low bits: the value).  Run numbers are TICKETS drawn from an atomic counter when a workgroup starts (round 5; round 4
// took blockIdx and relied on dispatch order), so the runs a workgroup waits for belong to workgroups that are already running, whatever
// the dispatch order and whoever else shares the device.  The look-back is made by a whole wave, 64 words per step (round 4: one thread,
// one dependent global load per run before it — with 512 workgroups in flight that walk was most of a tile's tail).
__device__ __forceinline__ uint32_t lookback_reserve_wave(unsigned long long* __restrict__ state, uint32_t run, uint32_t s) {
  const int lane = lane_id();
  if (lane == 0) __hip_atomic_store(&state[run], (1ull << 62) | (unsigned long long)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t excl = 0;
  for (long long hi = (long long)run - 1; hi >= 0;) {
    const long long idx = hi - lane;
    // (before the first run: everything so far = 0)
    const unsigned long long v = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
    const uint32_t flag = (uint32_t)(v >> 62);
    const unsigned long long m2 = __ballot(flag == 2u), m0 = __ballot(flag == 0u);
    const uint32_t first2 = m2 ? (uint32_t)__builtin_ctzll(m2) : 64u, first0 = m0 ? (uint32_t)__builtin_ctzll(m0) : 64u;
    // the words nearest to this run, up to the first inclusive one or to the first that is not written yet
    const uint32_t take = first0 < first2 ? first0 : (first2 < 64u ? first2 + 1u : 64u);
    const uint32_t part = wave_incl_scan((uint32_t)lane < take ? (uint32_t)v : 0u);
    excl += (uint32_t)__builtin_amdgcn_readlane((int)part, 63);
    if (first2 < first0) break;          // reached a run that knows everything before it
    hi -= take;
    if (first0 < 64u) __builtin_amdgcn_s_sleep(2);   // the next word is still being counted
  }
  if (lane == 0) __hip_atomic_store(&state[run], (2ull << 62) | (unsigned long long)(excl + s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

template <bool FOCUS, bool HEMI, class SM, bool ORDERED = false>
__device__ __forceinline__ void emit_tile_rows(const uint32_t* __restrict__ tal, SM sm, uint32_t n_tslots, MkpTile tl,
    uint32_t tix /* row-run index: key pass * tiles + tile */, uint32_t key, const MkpRunParams* __restrict__ prmp,
                                            const uint8_t* __restrict__ focus, const MkpCombo* combos_l, uint32_t* __restrict__ rows_base,
                                                uint32_t* __restrict__ row_cursor,
                                            uint32_t* __restrict__ tile_row_off, uint32_t* __restrict__ tile_row_cnt, uint32_t* __restrict__ dev_err,
                                            uint32_t* wave_tot, uint32_t* row_base_p, uint32_t* scan_carry_p,
                                                uint32_t n_runs = 0 /* ORDERED: runs of the launch sequence */) {
  const MkpRunParams& prm = *prmp;
  const int lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  MkpRowsDev rows;
  { const size_t cap = prm.row_capacity; uint32_t* p = rows_base;
    rows.pos = p; rows.info = p + cap; rows.code = p + 2 * cap; rows.n_valid = p + 3 * cap; rows.n_mod = p + 4 * cap; rows.n_can = p + 5 * cap;
      rows.n_other = p + 6 * cap;
    rows.n_del = p + 7 * cap; rows.n_fail = p + 8 * cap; rows.n_diff = p + 9 * cap; rows.n_nocall = p + 10 * cap; }
  TileView tv; tv.pk = tal; tv.W = prm.slot_cap; tv.n_counters = prm.n_counters;
  auto slot_of = [&](int32_t q) { return sm.rank(q); };
  if (FOCUS) {
    // focus tiles hold at most one slot per thread (the host planner caps them at MKP_PILEUP_THREADS): every thread counts its
    // slot's rows once, the block scans, thread 0 reserves the tile's run, and the same threads write — position, focus byte and
    // count stay in registers
    const uint32_t i = threadIdx.x;
    uint32_t cnt = 0, fv = 0; int32_t p = 0;
    if (i < n_tslots) {
      p = sm.pos_of(i);
      if (p >= tl.r0 && p < tl.r1) {
        if (HEMI) cnt = hemi_rows_at<false>(tal, prm.slot_cap, prm, p, i, rows, 0);
        else { fv = prm.has_focus ? (uint32_t)focus[p - prm.win_start] : 3u;
          cnt = rows_at<false>(tv, prm, focus, combos_l, p, i, rows, 0, fv, slot_of); }
      }
    }
    const uint32_t inc2 = wave_incl_scan(cnt);
    if (lane == 63) wave_tot[wave] = inc2;
    __syncthreads();
    if (ORDERED ? threadIdx.x < 64u : threadIdx.x == 0) {
      uint32_t s = 0;
      for (uint32_t w2 = 0; w2 < PILEUP_WAVES; w2++) s += wave_tot[w2];
      uint32_t base;
      // tile_row_off = the runs' look-back words (two dwords each); n_runs = number of runs (a kernel argument), row_cursor[1] = total rows, written
      // by the last run
      if (ORDERED) {
#ifdef MKP_DEBUG
        // ablation: no look-back (rows in completion order)
        if (prm.debug_skip & 4096u) { uint32_t b0 = 0; if (threadIdx.x == 0) b0 = atomicAdd(row_cursor + 1, s);
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0); } else
#endif
        {
        base = lookback_reserve_wave(reinterpret_cast<unsigned long long*>(tile_row_off), tix, s);
        if (threadIdx.x == 0 && tix + 1u == n_runs) row_cursor[1] = base + s;
        }
      } else base = s ? atomicAdd(row_cursor, s) : 0u;
      if (threadIdx.x == 0) {
        if (base + s > prm.row_capacity) { atomicOr(dev_err, ERR_ROW_CAP); s = 0; }
        *row_base_p = base; if (!ORDERED) { tile_row_off[tix] = base; tile_row_cnt[tix] = s; } *scan_carry_p = s ? 0u : 0xffffffffu;
      }
    }
    __syncthreads();
    if (*scan_carry_p != 0xffffffffu && cnt) {
      uint32_t woff = *row_base_p + inc2 - cnt;
      for (uint32_t w2 = 0; w2 < wave; w2++) woff += wave_tot[w2];
      if (HEMI) hemi_rows_at<true>(tal, prm.slot_cap, prm, p, i, rows, woff);
        else rows_at<true>(tv, prm, focus, combos_l, p, i, rows, woff, fv, slot_of, key);
    }
    return;
  }
  // pass 1: rows per slot, summed over the tile
  uint32_t mine = 0;
  for (uint32_t i = threadIdx.x; i < n_tslots; i += PILEUP_THREADS) {
    const int32_t p = sm.pos_of(i);
    if (p < tl.r0 || p >= tl.r1) continue;
    if (HEMI) { mine += hemi_rows_at<false>(tal, prm.slot_cap, prm, p, i, rows, 0); continue; }
    const uint32_t fv = prm.has_focus ? (uint32_t)focus[p - prm.win_start] : 3u;
    mine += rows_at<false>(tv, prm, focus, combos_l, p, i, rows, 0, fv, slot_of);
  }
  { const uint32_t inc2 = wave_incl_scan(mine); if (lane == 63) wave_tot[wave] = inc2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t s = 0;
    for (uint32_t w2 = 0; w2 < PILEUP_WAVES; w2++) s += wave_tot[w2];
    uint32_t base = s ? atomicAdd(row_cursor, s) : 0u;
    if (base + s > prm.row_capacity) { atomicOr(dev_err, ERR_ROW_CAP); s = 0; }
    *row_base_p = base; tile_row_off[tix] = base; tile_row_cnt[tix] = s; *scan_carry_p = s ? 0u : 0xffffffffu;
  }
  __syncthreads();
  // pass 2: write, 1024 slots at a time
  if (*scan_carry_p != 0xffffffffu) {
    const uint32_t row_base = *row_base_p;
    for (uint32_t i0 = 0; i0 < n_tslots; i0 += PILEUP_THREADS) {
      const uint32_t i = i0 + threadIdx.x;
      uint32_t cnt = 0, fv = 0; int32_t p = 0;
      if (i < n_tslots) {
        p = sm.pos_of(i);
        if (p >= tl.r0 && p < tl.r1) {
          if (HEMI) cnt = hemi_rows_at<false>(tal, prm.slot_cap, prm, p, i, rows, 0);
          else { fv = prm.has_focus ? (uint32_t)focus[p - prm.win_start] : 3u;
            cnt = rows_at<false>(tv, prm, focus, combos_l, p, i, rows, 0, fv, slot_of); }
        }
      }
      const uint32_t inc2 = wave_incl_scan(cnt);
      __syncthreads();   // wave_tot / scan_carry of the previous round are consumed
      if (lane == 63) wave_tot[wave] = inc2;
      __syncthreads();
      uint32_t woff = *scan_carry_p;
      for (uint32_t w2 = 0; w2 < wave; w2++) woff += wave_tot[w2];
      if (cnt) { if (HEMI) hemi_rows_at<true>(tal, prm.slot_cap, prm, p, i, rows, row_base + woff + inc2 - cnt);
        else rows_at<true>(tv, prm, focus, combos_l, p, i, rows, row_base + woff + inc2 - cnt, fv, slot_of, key);
        }
      __syncthreads();
      if (threadIdx.x == PILEUP_THREADS - 1) *scan_carry_p = woff + inc2;
    }
  }
}
