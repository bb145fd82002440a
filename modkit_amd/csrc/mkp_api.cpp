// libmkpileup C ABI (include/mkpileup.h): context, shard residency in HBM, kernel launches,
// row read-back.  No CPU fallback lives here: the only way rows are produced is the three HIP
// kernels of mkp_kernels.hip; without a gfx950 device every compute entry point fails with
// MKP_E_DEVICE.
#include <atomic>
#include <climits>
#include <memory>
#include <thread>

#include "mkp_ctx.hpp"
#include "mkp_ingest_host.hpp"

using namespace mkp;

extern "C" {
hipError_t mkp_launch_decode(hipStream_t, const MkpReadHdr*, const uint32_t* /*read ids by class*/, const uint32_t* /*n_class[7]*/, const uint32_t*,
    const uint8_t*,
    const MkpTagRef*, const uint32_t*,
                             const uint8_t*, const MkpLayout*, const MkpRunParams*, MkpEvent*, MkpReadOut*, uint32_t*, const uint8_t*, float*);
hipError_t mkp_pileup_set_lds(uint32_t accum_bytes);
hipError_t mkp_launch_pileup(hipStream_t, uint32_t /*LDS bytes*/, int /*focus mode*/, const MkpReadHdr*, const uint32_t*, const uint8_t*,
    const MkpEvent*,
    const MkpReadOut*, const MkpTile*, uint32_t,
                             const MkpRunParams* /*device*/, const uint32_t* /*slot bitmap*/, const uint8_t* /*focus bytes*/, const MkpCombo*,
                                 const MkpRowsDev*, uint32_t* /*row cursor*/,
                             uint32_t* /*tile row offsets*/, uint32_t* /*tile row counts*/, const uint32_t* /*chunk offsets*/,
                                 uint32_t* /*error bits*/, uint32_t /*key filter*/, uint32_t /*key pass*/);
hipError_t mkp_launch_slots(hipStream_t, const MkpWork* /*fused reads: long | short*/, uint32_t, uint32_t, const MkpReadHdr*,
    const uint32_t* /*cover read ids*/,
    uint32_t, const uint32_t*, const uint8_t*, const MkpTagRef*,
                            const uint32_t*, const uint8_t*, const MkpLayout*, const MkpFusedDesc*, const MkpRunParams*,
                                const uint32_t* /*slot positions*/, uint8_t* /*feature stream*/, MkpVisit*, MkpEvent*, MkpReadOut*, uint32_t*);
hipError_t mkp_stream_set_lds(uint32_t bytes);
hipError_t mkp_launch_dup_restore(hipStream_t, MkpReadHdr*, const MkpDupCons*, uint32_t);
hipError_t mkp_launch_dup_events(hipStream_t, MkpReadHdr*, const uint32_t*, const uint8_t*, MkpEvent*, MkpReadOut*, MkpDupCons*, const MkpDupSeg*,
    uint32_t, uint32_t* /*error bits*/);
hipError_t mkp_launch_stream(hipStream_t, uint32_t /*LDS bytes*/, const MkpVisit*, const uint8_t*, const MkpEvent*, const MkpSTile*, uint32_t,
    const MkpRunParams* /*device*/, const uint32_t* /*slot positions*/,
                             const uint8_t* /*focus bytes*/, const MkpCombo*, const MkpRowsDev*, uint32_t* /*row cursor*/, uint32_t*, uint32_t*,
                                 uint32_t* /*error bits*/, uint32_t /*key filter*/, uint32_t /*key pass*/, uint32_t /*motif combos*/,
                                 uint32_t /*row runs of the launch sequence*/,
                             uint32_t /*slot capacity of a tile*/, uint32_t /*tally words per slot*/);
hipError_t mkp_launch_gather(hipStream_t, const uint32_t*, const uint32_t*, uint32_t*, uint32_t, uint32_t*, const MkpRowsDev*, const MkpRowsDev*);
hipError_t mkp_launch_hemi_failed(hipStream_t, const MkpReadHdr*, const uint32_t*, const uint8_t*, MkpEvent*, MkpReadOut*, uint32_t,
    const uint32_t* /*slot bitmap*/,
    const uint32_t* /*interval starts*/,
                                  uint32_t, int32_t, int32_t, uint32_t* /*error bits*/);
}

namespace {

MkpRowsDev carve_rows(DevBuf& b, uint64_t cap) {
  b.ensure(cap * 44 + 64);
  MkpRowsDev r; uint32_t* p = b.as<uint32_t>();
  r.pos = p; r.info = p + cap; r.code = p + 2 * cap; r.n_valid = p + 3 * cap; r.n_mod = p + 4 * cap; r.n_can = p + 5 * cap; r.n_other = p + 6 * cap;
  r.n_del = p + 7 * cap; r.n_fail = p + 8 * cap; r.n_diff = p + 9 * cap; r.n_nocall = p + 10 * cap;
  return r;
}

// reads by decode kernel: [SPARSE one tag | SPARSE two tags | FAST one tag | FAST two tags | everything else].
// FAST = one (strand, base) group, no code listed twice, at most two tags.  SPARSE = FAST with explicit ('?') tags only and,
// for two tags, identical rank lists (`C+h?,d..;C+m?,d..` as basecallers write them): the calls are located per call, not per base.
// With `duplex` (never for the threshold sampler): [.. | duplex one tag per group | duplex two]: reads whose layout has two groups on
// different bases (`C+h?;C+m?;G-h?;G-m?`), every tag explicit and the tags of a group sharing one rank list; such a read is listed
// twice (bit 31 = its second group), each listing decoded by a SPARSE wave, and mkp_merge_duplex interleaves the two event lists.
template <class F> void host_parallel(size_t n, size_t grain, F f);
// ids reordered by key(id) DESCENDING, equal keys keeping their order (what std::stable_sort with `>` gives): three 11-bit counting passes
// — the planner sorts a shard's 200 000 reads by length three times, and comparison sorts were a third of its time
template <class Key> void stable_sort_desc(std::vector<uint32_t>& ids, Key key) {
  const size_t n = ids.size(); if (n < 2) return;
  if (n < 2048) { std::stable_sort(ids.begin(), ids.end(), [&](uint32_t x, uint32_t y) { return key(x) > key(y); }); return; }
  std::vector<uint32_t> k(n), tmp(n), ktmp(n);
  for (size_t i = 0; i < n; i++) k[i] = ~key(ids[i]);   // ascending in the complement = descending in the key
  for (int pass = 0; pass < 3; pass++) {
    const int sh = 11 * pass; size_t cnt[2049] = {0};
    for (size_t i = 0; i < n; i++) cnt[((k[i] >> sh) & 2047u) + 1]++;
    bool trivial = false; for (size_t b = 1; b <= 2048; b++) if (cnt[b] == n) trivial = true;
    if (trivial) continue;
    for (size_t b = 0; b < 2048; b++) cnt[b + 1] += cnt[b];
    for (size_t i = 0; i < n; i++) { const size_t at = cnt[(k[i] >> sh) & 2047u]++; tmp[at] = ids[i]; ktmp[at] = k[i]; }
    ids.swap(tmp); k.swap(ktmp);
  }
}
inline void sort_u32(std::vector<uint32_t>& v) {   // ascending, three 11-bit counting passes
  const size_t n = v.size(); if (n < 4096) { std::sort(v.begin(), v.end()); return; }
  std::vector<uint32_t> tmp(n);
  for (int pass = 0; pass < 3; pass++) {
    const int sh = 11 * pass; const uint32_t mask = pass == 2 ? 1023u : 2047u; size_t cnt[2049] = {0};
    for (size_t i = 0; i < n; i++) cnt[((v[i] >> sh) & mask) + 1]++;
    for (size_t b = 0; b < 2048; b++) cnt[b + 1] += cnt[b];
    for (size_t i = 0; i < n; i++) tmp[cnt[(v[i] >> sh) & mask]++] = v[i];
    v.swap(tmp);
  }
}
void class_ids(const ShardHost& S, const LayoutTables& T, std::vector<uint32_t>* ids, uint32_t n_class[7], bool duplex) {
  auto class_of = [&](size_t i) -> int {
    const MkpReadHdr& h = S.hdr[i];
    auto same = [&](uint32_t t0, uint32_t t1) { const MkpTagRef &a = S.tagref[h.tag_off + t0], &b = S.tagref[h.tag_off + t1];
        // compared on the device while the lists were written (only neighbours are ever asked about)
        if (S.dev_packed) return t1 == t0 + 1 && b.pad != 0;
        return a.n == b.n && (a.n == 0 || memcmp(&S.ranks[a.rank_off], &S.ranks[b.rank_off], 4 * (size_t)a.n) == 0); };
    if ((h.flags & MKP_RF_BAD) || !h.n_tags || h.layout >= T.dev.size()) return 4;
    const MkpLayout& L = T.dev[h.layout];
    bool explicit_tags = true;
    for (uint32_t t = 0; t < h.n_tags; t++) if (L.tags[t].mode != 0) explicit_tags = false;
    if (duplex && L.fast == 2) {
      const uint32_t nA = L.pad;
      if (explicit_tags && (nA != 2 || same(0, 1)) && (h.n_tags - nA != 2 || same(nA, nA + 1))) return (nA == 1 && h.n_tags - nA == 1) ? 5 : 6;
      return 4;
    }
    if (L.fast == 1 && h.n_tags <= 2) {
      const bool sparse = explicit_tags && (h.n_tags == 1 || same(0, 1));
      return (sparse ? 0 : 2) + (int)(h.n_tags - 1);
    }
    return 4;
  };
  // classified on all cores (the rank-list comparisons touch every call of a two-tag read); then one list per class,
  // one wave decodes one read start to end: the longest reads are launched first so that they do not form the kernel's tail
  std::vector<uint8_t> cl(S.hdr.size());
  host_parallel(S.hdr.size(), 4096, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) cl[i] = (uint8_t)class_of(i); });
  std::vector<uint32_t> cls[7];
  for (size_t i = 0; i < cl.size(); i++) cls[cl[i]].push_back((uint32_t)i);
  host_parallel(7, 1, [&](size_t lo, size_t hi) { for (size_t c = lo; c < hi; c++) stable_sort_desc(cls[c], [&](uint32_t x) { return S.hdr[x].l_seq;
    }); });
  ids->clear();
  for (int c = 0; c < 5; c++) { n_class[c] = (uint32_t)cls[c].size(); ids->insert(ids->end(), cls[c].begin(), cls[c].end()); }
  for (int c = 5; c < 7; c++) { n_class[c] = 2u * (uint32_t)cls[c].size(); for (uint32_t r : cls[c]) { ids->push_back(r);
      ids->push_back(r | 0x80000000u); } }
}

// MkpFusedDesc of a layout whose tags form one explicit-mode group (decode class SPARSE): the walk of
// MultipleThresholdModCaller::call (threshold_mod_caller.rs:28-63) over a call's map, resolved from the layout's caller tables
MkpFusedDesc fused_desc(const MkpLayout& D) {
  MkpFusedDesc f; memset(&f, 0, sizeof(f));
  if (D.fast != 1 || D.n_tags == 0 || D.n_tags > 2) return f;
  const uint32_t b0 = D.tags[0].fb & 3u, sg0 = D.tags[0].neg & 1u;
  const MkpGroupDesc& G = D.groups[sg0 * 4 + b0];
  uint32_t hit = 0; for (uint32_t t = 0; t < D.n_tags; t++) hit |= 1u << (D.tagmap[t][b0] & 15u);
  const uint32_t pv = G.pat[hit < 20u ? hit : 0u], n_post = std::min<uint32_t>((pv >> 3) & 7u, MKP_KMAX);
  uint32_t ob = 0;
  for (uint32_t i = 0; i < n_post; i++) {
    const uint32_t kq = (pv >> (16 + 2 * i)) & 3u;
    ob |= 1u << ((G.slots >> (8 * kq)) & 0xffu);
    f.it_cid |= ((G.cids >> (8 * kq)) & 0xffu) << (8 * i); f.it_thr[i] = G.thr_mod[kq];
    for (uint32_t t = 0; t < D.n_tags; t++) for (uint32_t k = 0; k < D.tags[t].n_codes; k++)
      if (((D.tagmap[t][b0] >> (4 + 4 * k)) & 15u) == kq) f.it_src |= (t | (k << 1)) << (4 * i);
  }
  f.misc = b0 | (sg0 << 2) | (n_post << 3) | (MKP_G_CIDCAN(G.misc) << 8) | (ob << 16);
  f.nc = (uint32_t)D.tags[0].n_codes | ((D.n_tags > 1 ? (uint32_t)D.tags[1].n_codes : 0u) << 8);
  f.thr_can = G.thr_can;
  // the thresholds of the integer caller: least T with T / 2048 >= threshold (exact in double: a float times 2^11)
  auto i_of = [](float thr) -> int32_t { if (!(thr == thr)) return INT32_MAX; const double t = std::ceil((double)thr * 2048.0); return (int32_t)std::max(-1073741824.0,
      std::min(1073741824.0, t)); };
  for (uint32_t i = 0; i < MKP_KMAX; i++) f.i_thr[i] = i_of(f.it_thr[i]);
  f.i_can = i_of(f.thr_can);
  f.misc |= 1u << 6;
  const int x = MKP_G_COLL(G.misc); const uint32_t n_pre = pv & 7u;   // collapse_redistribute's inputs (only looked at in collapse runs)
  bool col_exact = true;
  for (uint32_t i = 0; i < n_pre && i < MKP_KMAX; i++) if ((int)((pv >> (8 + 2 * i)) & 3u) == x) {
    f.col = 1u; f.n_other = (float)n_pre;
    for (uint32_t t = 0; t < D.n_tags; t++) for (uint32_t k = 0; k < D.tags[t].n_codes; k++)
      if ((int)((D.tagmap[t][b0] >> (4 + 4 * k)) & 15u) == x) f.col |= (t | (k << 1)) << 1;
    // (a share over three codes is rounded: the f32 walk stays)
    if (n_pre == 1 || n_pre == 2 || n_pre == 4) f.col |= (n_pre == 1 ? 0u : n_pre == 2 ? 1u : 2u) << 5; else col_exact = false;
  }
  if (col_exact) f.misc |= 1u << 7;
  return f;
}

template <class V> void upload(DevBuf& b, const V& v) {
  const size_t bytes = v.size() * sizeof(v[0]);
  b.ensure(std::max<size_t>(bytes, 16));
  if (!v.empty()) h2d_copy(b.p, v.data(), bytes);   // (through the library's page-locked staging: mkp_ctx.hpp)
}

template <class F> void host_parallel(size_t n, size_t grain, F f) {   // f(lo, hi) over [0, n) in `grain`-sized pieces on the host pool
  const size_t pieces = (n + grain - 1) / grain;
  HostPool::get().parallel(pieces, [&](size_t i) { f(i * grain, std::min(n, (i + 1) * grain)); });
}

// htslib's pileup engine (bam_plp_push; `set_max_depth` = bam_plp_set_maxcnt, pileup/mod.rs:755-759) refuses a record when it starts
// where the last buffered record started and the buffer — the records that end at or behind that position, plus the list's tail node —
// holds more than max_depth entries (`iter->pos == b->core.pos && iter->mp->cnt > iter->maxcnt`); the first record of a start is always
// taken, so depth alone drops nothing.  Every interval of the reference's grid is a fetch with an iterator of its own over the records
// that overlap it.  Dropping per (record, interval) is not reproduced on the device; what is decided here, exactly, is whether ANY
// record would be dropped: an interval [a, e) and a start b shared by two or more of its records with
//     #{records of the interval: start <= b and end >= b}  >  max_depth        (end > a where b <= a: the fetch's own condition).
// If there is none, htslib keeps every record at any depth and the device result is exact (up to the tallies' 65 535).  The population
// is htslib's: every record passing BAM_DEF_MASK (kept reads and supplementary ones), by reference span (ref-skips included).
void depth_guard(const ShardHost& S, uint32_t max_depth, const std::vector<uint32_t>& iv_starts) {
  const size_t n = S.hdr.size() + S.extra_spans.size();
  if (n <= max_depth) return;
  std::vector<std::pair<uint32_t, uint32_t>> sp; sp.reserve(n);   // (start, end), starts ascending (alignment starts and ends are non-negative)
  for (auto& h : S.hdr) sp.push_back({(uint32_t)h.ref_start, (uint32_t)std::max(h.ref_end, h.ref_start + 1)});
  for (auto& x : S.extra_spans) sp.push_back({(uint32_t)x.first, (uint32_t)std::max(x.second, x.first + 1)});
  if (!std::is_sorted(sp.begin(), sp.end(), [](auto& x, auto& y) { return x.first < y.first; }))
    std::stable_sort(sp.begin(), sp.end(), [](auto& x, auto& y) { return x.first < y.first; });
  std::vector<uint32_t> en(n); for (size_t i = 0; i < n; i++) en[i] = sp[i].second;
  sort_u32(en);
  size_t j = 0, cur = 0, best = 0;
  for (size_t i = 0; i < n; i++) { while (j < n && en[j] <= sp[i].first) { j++; cur--; } cur++; best = std::max(best, cur); }
  if (best > 65535) throw Error(MKP_E_UNSUPPORTED,
      "more than 65535 reads over one position: columns this deep are outside the device path (16-bit packed tallies)");
  auto refuse = [&](uint32_t b, size_t held) {
    throw Error(MKP_E_UNSUPPORTED, "htslib would drop records here: " + std::to_string(held) + " buffered records at position " + std::to_string(b) +
        ", where several records start, exceed max_depth (" + std::to_string(max_depth) + "); bam_plp_push's maxcnt read dropping is not reproduced"); };
  auto ends_below = [&](uint32_t v) { return (size_t)(std::lower_bound(en.begin(), en.end(), v) - en.begin()); };   // #{end < v}
  // interval starts of the shard (none given: the shard is one interval)
  std::vector<uint32_t> A; A.push_back((uint32_t)std::max(S.win_start, 0));
  for (size_t k = 1; k < iv_starts.size(); k++) if (iv_starts[k] > A.back()) A.push_back(iv_starts[k]);
  // (1) starts b inside an interval (a <= b): the interval's fetch holds every record with start <= b <= end
  for (size_t i = 0; i < n;) {
    size_t r = i; while (r < n && sp[r].first == sp[i].first) r++;
    const uint32_t b = sp[i].first;
    if (r - i >= 2 && b >= A[0]) {
      const bool on_start = std::binary_search(A.begin(), A.end(), b);   // b == a: records that end AT a are not fetched
      const size_t held = r - (on_start ? ends_below(b + 1) : ends_below(b));
      if (held > max_depth) refuse(b, held);
    }
    i = r;
  }
  // (2) starts in front of an interval's start a: among the records with start < a < end
  for (uint32_t a : A) {
    const size_t before = (size_t)(std::lower_bound(sp.begin(), sp.end(), a, [](auto& x, uint32_t v) { return x.first < v; }) - sp.begin());
    if (before - std::min(before, ends_below(a + 1)) <= max_depth) continue;
    size_t held = 0;
    for (size_t i = 0; i < before;) {
      size_t r = i, m = 0; while (r < before && sp[r].first == sp[i].first) { m += sp[r].second > a; r++; }
      held += m;
      if (m >= 2 && held > max_depth) refuse(sp[i].first, held);
      i = r;
    }
  }
}

// BGZF inflate on the device: mkp_inflate_wave4, one wave per block — speculative token decode, a scalar walk that only marks the chain,
// all output of a pass placed at once; 4 KiB ring, sixteen waves per CU (round 5; it replaced the five kernels of rounds 2-4 at every
// launch size).  MKP_INFLATE_KERNEL=wave4_8k|wave4_2k picks another ring size (A/B runs; --stats names it).
}  // namespace
extern "C" hipError_t mkp_launch_inflate_wave4(hipStream_t, const uint8_t*, const void* /*MkpBgzfBlock[]*/, uint32_t, uint8_t*, uint32_t*, int);
hipError_t mkp_launch_inflate_auto(hipStream_t st, const uint8_t* in, const void* blks, uint32_t n, uint8_t* out, uint32_t* status) {
  static const char* force = getenv("MKP_INFLATE_KERNEL");
  if (force && !strcmp(force, "wave4_8k")) return mkp_launch_inflate_wave4(st, in, blks, n, out, status, 8);
  if (force && !strcmp(force, "wave4_2k")) return mkp_launch_inflate_wave4(st, in, blks, n, out, status, 2);
  return mkp_launch_inflate_wave4(st, in, blks, n, out, status, 4);
}
namespace {
hipError_t launch_inflate(hipStream_t st, const uint8_t* in, const void* blks, uint32_t n, uint8_t* out, uint32_t* status) {
  return mkp_launch_inflate_auto(st, in, blks, n, out, status); }

// slot bitmap of a focus window (bit p - win_start + margin set where position p owns a tally column), its running popcount per word and —
// slot pipeline — the slot positions.  Depends on the focus bytes only.
void window_slots(mkp_ctx* c, bool hemi, bool stream, std::vector<uint32_t>& slotbm, std::vector<uint32_t>& wpfx, std::vector<uint32_t>& slot_pos_h) {
  const ShardHost& S = c->shard;
  const int64_t win = (int64_t)S.win_end - (int64_t)S.win_start;
  const size_t nbits = (size_t)win + 2 * MKP_SLOTBM_MARGIN, nwords = (nbits + 31) / 32 + 2;
  slotbm.assign(nwords, 0);
  const uint8_t* fz = c->focus.data();
  // pileup-hemi: only the positions with a positive-strand motif hit own a column (positions_to_motifs.get(pos), duplex.rs:289-296)
  uint8_t hemi_ok[64]; for (size_t k = 0; k < 64; k++) hemi_ok[k] = (k < c->combos.size() && c->combos[k].n_pos > 0) ? 1 : 0;
  // pieces are multiples of 32 positions and the margin is 64: no two pieces share a word
  host_parallel((size_t)win, (size_t)1 << 20, [&](size_t lo, size_t hi) {
    for (size_t p = lo; p < hi; p++) if (hemi ? ((fz[p] & 1u) && hemi_ok[fz[p] >> 2]) : (fz[p] & 3u)) { const size_t b = p + MKP_SLOTBM_MARGIN;
        slotbm[b >> 5] |= 1u << (b & 31); }
  });
  wpfx.assign(nwords + 1, 0);
  {   // running popcount over the bitmap words: block sums on all cores, a short serial pass over the blocks, then the blocks again
    const size_t blk = (size_t)1 << 16, nblk = (nwords + blk - 1) / blk; std::vector<uint32_t> bsum(nblk + 1, 0);
    host_parallel(nblk, 1, [&](size_t lo, size_t hi) { for (size_t b = lo; b < hi; b++) { uint32_t t = 0;
        for (size_t w = b * blk; w < std::min(nwords, (b + 1) * blk); w++) t += (uint32_t)__builtin_popcount(slotbm[w]);
        bsum[b + 1] = t; } });
    for (size_t b = 0; b < nblk; b++) bsum[b + 1] += bsum[b];
    host_parallel(nblk, 1, [&](size_t lo, size_t hi) { for (size_t b = lo; b < hi; b++) { uint32_t run = bsum[b];
        for (size_t w = b * blk; w < std::min(nwords, (b + 1) * blk); w++) { wpfx[w] = run;
          run += (uint32_t)__builtin_popcount(slotbm[w]); } } });
    wpfx[nwords] = bsum[nblk];
  }
  slot_pos_h.clear();
  if (stream) {
    slot_pos_h.resize(wpfx[nwords]);
    host_parallel(nwords, (size_t)1 << 15, [&](size_t lo, size_t hi) {
      for (size_t w = lo; w < hi; w++) { uint32_t at = wpfx[w];
          for (uint32_t bits = slotbm[w]; bits; bits &= bits - 1u) slot_pos_h[at++] = (uint32_t)((int64_t)(w * 32 + (size_t)__builtin_ctz(bits)) - MKP_SLOTBM_MARGIN + S.win_start);
          }
    });
  }
}
bool stream_pipeline(const mkp_ctx* c, bool hemi) { return c->has_focus && !hemi; }

// derive tile geometry, tile read ranges and the run parameters; upload everything
void make_resident(mkp_ctx* c) {
  auto t0 = std::chrono::steady_clock::now();
  ShardHost& S = c->shard;
  const bool trace = getenv("MKP_TRACE_PLAN") != nullptr;   // host planning stages on stderr
  auto lap = [&, last = t0](const char* what) mutable { if (trace) { auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[mkpileup plan] %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count()); last = now; } };
  // hazard: the reference's ReadCache is keyed by read NAME (read_cache.rs:28-35); two kept records with
  // one name in one interval share a cache entry there: the later one is answered from the earlier one's calls.  Found here, planned
  // behind the tile plan (it needs the focus positions), reproduced by mkp_dup_events (mkp_slots.hip).
  std::vector<std::vector<uint32_t>> dup_groups;   // records of one name (and partition key) that meet inside an interval: planned below (plan_dups)
  { const size_t nn = S.name_hash.size();
    // every thread takes the names whose hash falls into its sixteenth and looks for a repeat in an open-addressing table of its own
    // (value = the first record carrying the name); repeats are rare, so they are only collected here and judged below
    std::mutex dmu; std::vector<std::pair<uint32_t, uint32_t>> dups;   // (first record with the name, a later one)
    host_parallel(16, 1, [&](size_t lo, size_t hi) { for (size_t part = lo; part < hi; part++) {
      size_t mine = 0; for (size_t i = 0; i < nn; i++) if ((S.name_hash[i] >> 60) == part) mine++;
      if (mine < 2) continue;
      size_t cap = 64; while (cap < 2 * mine) cap <<= 1;
      std::vector<uint64_t> tab(cap, 0), key(cap, 0); std::vector<uint32_t> who(cap, 0); std::vector<uint8_t> used(cap, 0);
      for (size_t i = 0; i < nn; i++) { const uint64_t hh = S.name_hash[i]; if ((hh >> 60) != part) continue;
        const uint64_t kk = i < S.hdr.size() ? S.hdr[i].flags >> MKP_RF_KEY_SHIFT : 0u;   // per partition key: tallies of different keys never meet
        size_t at = (size_t)((hh * 0x9e3779b97f4a7c15ull) >> 20) & (cap - 1);
        for (;;) { if (!used[at]) { used[at] = 1; tab[at] = hh; key[at] = kk; who[at] = (uint32_t)i; break; }
                   if (tab[at] == hh && key[at] == kk) { std::lock_guard<std::mutex> g(dmu); dups.push_back({who[at], (uint32_t)i}); break;
                     } at = (at + 1) & (cap - 1); } }
    } });
    // The reference keys its read cache by NAME, one cache per interval (read_cache.rs:28-35, pileup/mod.rs:718-760): two kept records
    // with one name meet only if they overlap a common interval — then the later one is answered from the earlier one's calls, which the
    // device path does not reproduce.  Mates / split alignments that lie in different intervals never share a cache and are simply two reads.
    // (No interval grid given: the shard is one interval.)
    // Every pair of records of one name is judged (three records A, B, C: the table above names (A, B) and (A, C); B and C can meet too).
    // A record that lies wholly outside the shard window — a halo record of the fetch — belongs to none of its intervals.
    std::map<uint32_t, std::vector<uint32_t>> groups;
    for (auto& d : dups) { if (d.first >= S.hdr.size() || d.second >= S.hdr.size()) continue; auto& g = groups[d.first];
      if (g.empty()) g.push_back(d.first);
      g.push_back(d.second); }
    auto iv_range = [&](const MkpReadHdr& h, int64_t* a, int64_t* b) -> bool {   // false: outside the window
      const int64_t s0 = h.ref_start, e0 = std::max<int64_t>(h.ref_end, (int64_t)h.ref_start + 1);
      if (e0 <= (int64_t)S.win_start || s0 >= (int64_t)S.win_end) return false;
      if (c->iv_starts.empty()) { *a = 0; *b = 0; return true; }
      auto idx = [&](int64_t p) {
        return (int64_t)(std::upper_bound(c->iv_starts.begin(), c->iv_starts.end(), (uint32_t)std::max<int64_t>(p, 0)) - c->iv_starts.begin()) - 1; };
      *a = std::max<int64_t>(idx(std::max<int64_t>(s0, S.win_start)), 0); *b = std::max<int64_t>(idx(std::min<int64_t>(e0, S.win_end) - 1), 0);
      return true;
    };
    for (auto& kv : groups) {
      std::vector<uint32_t> g = kv.second; std::sort(g.begin(), g.end()); g.erase(std::unique(g.begin(), g.end()), g.end());
      std::vector<std::pair<int64_t, int64_t>> rg; rg.reserve(g.size());
      for (uint32_t r : g) { int64_t a, b; if (iv_range(S.hdr[r], &a, &b)) rg.push_back({a, b}); }
      bool meet = false;
      for (size_t x = 0; x < rg.size() && !meet; x++) for (size_t y = x + 1; y < rg.size(); y++) if (rg[x].first <= rg[y].second
          && rg[y].first <= rg[x].second) {
        meet = true; break; }
      if (!meet) continue;
      // pileup-hemi keys its DuplexReadCache by name as well (read_cache.rs:368-468); that form is not reproduced
      if (c->hemi) throw Error(MKP_E_UNSUPPORTED,
          "two primary records share a read name inside one interval (unmarked duplicates, or mates / split reads that overlap the same interval); pileup-hemi answers the later record from the earlier one's calls (its per-interval cache is keyed by name) and this is not reproduced on the device");
      dup_groups.push_back(std::move(g));
    }
  }
  lap("duplicate-name check");
  // caller tables over the layouts this shard's reads use (not whatever the packer has interned before)
  { std::vector<uint8_t> used(c->packer.layouts.size(), 0);
    for (auto& h : S.hdr) if (!(h.flags & MKP_RF_BAD) && h.n_tags && h.layout < used.size()) used[h.layout] = 1;
      c->tables.build(c->packer.layouts, c->caller, &used); }
  MkpRunParams& P = c->prm; memset(&P, 0, sizeof(P));
  P.win_start = S.win_start; P.win_end = S.win_end;
  P.n_counters = c->tables.n_counters; P.n_slots = (uint32_t)c->tables.st.slots.size(); P.n_pb = (uint32_t)c->tables.st.can_pbs.size();
  P.numeric_mode = c->caller.numeric_mode; P.combine_strands = c->caller.combine_strands; P.edge_filter = c->caller.edge;
    P.edge_start = c->caller.edge_start;
  P.edge_end = c->caller.edge_end; P.edge_inverted = c->caller.edge_inverted; P.force_allow = c->caller.force_allow;
    P.max_depth = c->caller.max_depth;
  P.has_focus = c->has_focus; P.n_combos = (uint32_t)c->combos.size();
#ifdef MKP_DEBUG
  if (const char* dbg = getenv("MKP_DEBUG_SKIP")) P.debug_skip = (uint32_t)strtoul(dbg, nullptr, 0);
#endif
  for (int b = 0; b < 4; b++) { P.can_of_pb[b] = 0xff; P.pb_of_can[b] = 0; }
  for (size_t k = 0; k < c->tables.st.can_pbs.size(); k++) { P.can_of_pb[c->tables.st.can_pbs[k]] = (uint8_t)k;
    P.pb_of_can[k] = (uint8_t)c->tables.st.can_pbs[k]; }
  std::vector<int> order(P.n_slots); for (uint32_t i = 0; i < P.n_slots; i++) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { const MkpSlot &x = c->tables.st.slots[(size_t)a],
      &y = c->tables.st.slots[(size_t)b]; return x.code_repr != y.code_repr ? x.code_repr < y.code_repr : x.pb < y.pb; });
  for (uint32_t i = 0; i < P.n_slots; i++) { P.slot_order[i] = (uint8_t)order[i]; P.slots[i] = c->tables.st.slots[i]; }
  lap("caller tables + params");
  if (P.combine_strands && !P.has_focus) throw Error(MKP_E_INVALID, "combine_strands needs motif focus positions");
  if (c->hemi) {
    // pileup-hemi counters: a pattern block of (1 + codes)^2 counters per primary base that has calls, elements in DuplexModCodeRepr
    // order (Canonical < Code(char) < ChEbi(id) = 0 < the code_repr encoding's numeric order)
    if (!c->has_focus) throw Error(MKP_E_INVALID, "pileup-hemi needs the focus positions of a palindromic motif");
    if (!c->partition_tags.empty()) throw Error(MKP_E_INVALID, "pileup-hemi has no partition tags");
    P.hemi = 1; P.hemi_off = c->hemi_off;
    memset(P.hemi_el, 0xff, sizeof(P.hemi_el)); memset(c->hemi_codes, 0, sizeof(c->hemi_codes));
    uint32_t next = MKP_H_PAT;
    for (int b = 0; b < 4; b++) { P.hemi_pat_base[b] = 0xff; P.hemi_nel[b] = 0; }
    for (size_t k = 0; k < c->tables.st.can_pbs.size(); k++) {
      const int pb = c->tables.st.can_pbs[k];
      std::vector<int> mine; for (size_t i = 0; i < c->tables.st.slots.size(); i++) if (c->tables.st.slots[i].pb == pb) mine.push_back((int)i);
      std::sort(mine.begin(), mine.end(), [&](int x, int y) {
        return c->tables.st.slots[(size_t)x].code_repr < c->tables.st.slots[(size_t)y].code_repr; });
      const bool comb = P.numeric_mode == 1;   // DuplexModCall::into_combined: every modified element becomes the base's any-mod code
      const uint32_t nel = comb ? (mine.empty() ? 1u : 2u) : 1u + (uint32_t)mine.size();
      if (nel > MKP_KMAX + 1) throw Error(MKP_E_UNSUPPORTED, "more mod codes on one base than a pileup-hemi pattern block holds");
      P.hemi_pat_base[pb] = (uint8_t)next; P.hemi_nel[pb] = (uint8_t)nel; next += nel * nel;
      P.hemi_el[MKP_C_CAN + k] = 0;
      for (size_t r = 0; r < mine.size(); r++) {
        const MkpSlot& sl = c->tables.st.slots[(size_t)mine[r]];
        P.hemi_el[sl.cid] = (uint8_t)(comb ? 1u : 1u + r);
        c->hemi_codes[pb][comb ? 1u : 1u + r] = comb ? (uint32_t)"ACGT"[pb] : sl.code_repr;
      }
    }
    if (next > MKP_H_MAX_COUNTERS) throw Error(MKP_E_UNSUPPORTED, "too many pileup-hemi pattern counters for one LDS tile");
    P.hemi_counters = next;
    // a record whose tags fail leaves one NoCall per interval it crosses (mkp_hemi_failed_reads), written into its own event slice:
    // make every slice at least that long
    if (c->hemi_iv.empty()) c->hemi_iv.push_back((uint32_t)S.win_start);
    if (!std::is_sorted(c->hemi_iv.begin(), c->hemi_iv.end())) throw Error(MKP_E_INVALID, "interval starts must ascend");
    for (auto& h : S.hdr) {
      auto iv_of = [&](int64_t p) {
        return (int64_t)(std::upper_bound(c->hemi_iv.begin(), c->hemi_iv.end(), (uint32_t)std::max<int64_t>(p, 0)) - c->hemi_iv.begin()); };
      const int64_t need = iv_of((int64_t)h.ref_end - 1) - iv_of(h.ref_start) + 1;
      h.event_cap = (uint32_t)std::max<int64_t>(h.event_cap, need);
    }
  }
  // decode kernel classes; a duplex read decoded one group per wave needs room for both groups' lists behind the merged one
  std::vector<uint32_t> class_list; class_ids(S, c->tables, &class_list, c->n_class, true);
  lap("decode classes");
  // SPARSE reads with two tags: combine_checked's test (mod_bam.rs:629-656) that a call's probabilities over both tags do not add up
  // to more than 1.01 — every term is (2q + 1) / 512, so the f32 sum is exact and `> 1.01f` is "the numerators reach 518" — is made
  // here over the ML bytes, on all cores, and handed to the slot decoder as a header flag
  host_parallel(c->n_class[1], 2048, [&](size_t lo, size_t hi) {
    for (size_t k = lo; k < hi; k++) {
      MkpReadHdr& h = S.hdr[class_list[c->n_class[0] + k]];
      if (S.dev_packed) { if (S.dev_sum2[class_list[c->n_class[0] + k]]) h.flags |= MKP_RF_SUMERR; else h.flags &= ~MKP_RF_SUMERR; continue; }
      const MkpLayout& L = c->tables.dev[h.layout];
      const MkpTagRef &t0 = S.tagref[h.tag_off], &t1 = S.tagref[h.tag_off + 1];
      const uint32_t nc0 = L.tags[0].n_codes, nc1 = L.tags[1].n_codes;
      bool bad = false;
      for (uint32_t j = 0; j < t0.n && !bad; j++) {
        uint32_t num = 0;
        for (uint32_t i = 0; i < nc0; i++) num += 2u * S.ml[t0.ml_off + j * nc0 + i] + 1u;
        for (uint32_t i = 0; i < nc1; i++) num += 2u * S.ml[t1.ml_off + j * nc1 + i] + 1u;
        bad = num >= 518u;
      }
      if (bad) h.flags |= MKP_RF_SUMERR; else h.flags &= ~MKP_RF_SUMERR;
    }
  });
  lap("probability-sum test");
  { size_t at = 0; for (int k = 0; k < 5; k++) at += c->n_class[k];
    for (; at < class_list.size(); at += 2) {
      MkpReadHdr& h = S.hdr[class_list[at]];
      const uint64_t need = 2ull * ((uint64_t)S.tagref[h.tag_off].n + S.tagref[h.tag_off + c->tables.dev[h.layout].pad].n);
      if (need > 0xffffffffull) throw Error(MKP_E_UNSUPPORTED, "more than 2^32 call events in one read");
      h.event_cap = std::max(h.event_cap, (uint32_t)need);
    } }
  { uint64_t off = 0;   // event slices laid out again (capacities may have grown above)
    for (auto& h : S.hdr) {
      if (off > 0xfffffff0ull - h.event_cap) throw Error(MKP_E_UNSUPPORTED, "shard exceeds 4 Gi call events; use smaller shards");
        h.event_off = (uint32_t)off; off += h.event_cap; }
    S.n_events_cap = off; }
  P.readout_b_off = (uint32_t)S.hdr.size();
  lap("decode classes + event slices");
  const size_t n = S.hdr.size();
  for (size_t i = 1; i < n; i++) if (S.hdr[i].ref_start < S.hdr[i - 1].ref_start) throw Error(MKP_E_INVALID, "records must be coordinate sorted");
  depth_guard(S, c->caller.max_depth, c->iv_starts);
  lap("sortedness + depth guard");

  // ---- tile plan.  The accumulate kernel runs two 1024-thread workgroups per CU, each holding one tile in LDS: 76 KiB per
  // workgroup including ~3.5 KiB of static LDS leaves slack for the allocation granule (at 80 KiB each one GPU box ran them one
  // per CU and the kernel took 1.9x as long).  A tile = a run of reference positions; its tally columns ("slots") are all of
  // its positions, or — when the run has focus positions — only those, so a --cpg tile spans ~50x more reference.
  const uint32_t words_per_slot = c->hemi ? P.hemi_counters : P.n_counters + P.n_slots;
  const uint32_t budget_words = (76u * 1024u - 3584u) / 4u;
  const int64_t win = (int64_t)S.win_end - (int64_t)S.win_start;
  std::vector<MkpTile> tiles; std::vector<uint32_t> slotbm; uint32_t Scap = 0, Wcap = 0;
  // focus runs take the slot pipeline (mkp_slots.hip); pileup-hemi keeps the tile walk
  const bool stream = stream_pipeline(c, c->hemi);
  std::vector<uint32_t> slot_pos_h, wpfx; std::vector<MkpSTile> stiles; bool preplanned = false;
  c->slot_mode = stream; P.slot_stream = stream ? 1u : 0u; c->cov_bytes = 0;
  auto max_slots_for = [&](uint32_t W) { uint32_t best = 0; for (uint32_t s = 64; s <= 4160; s += 64) if (MKP_PILEUP_LDS_WORDS(words_per_slot, s,
      W) <= budget_words) best = s;
  return best; };
  if (!c->has_focus) {
    uint32_t Smax = max_slots_for(0);
    if (Smax < 128) throw Error(MKP_E_UNSUPPORTED, "too many counters for one LDS tile");
    uint32_t T = Smax - 2 * MKP_HALO;
    if (c->cfg.tile_positions) T = std::min(T, std::max<uint32_t>(32u, c->cfg.tile_positions));
    Scap = (T + 2 * MKP_HALO + 63u) & ~63u;
    for (int64_t r0 = S.win_start; r0 < S.win_end; r0 += T) tiles.push_back({(int32_t)r0, (int32_t)std::min<int64_t>(r0 + T, S.win_end), 0, 0});
  } else {
    // slot bitmap + running popcount + slot positions: made ahead of the reads when the driver asked for it (mkp_internal_shard_preplan)
    const size_t nbits = (size_t)win + 2 * MKP_SLOTBM_MARGIN, nwords = (nbits + 31) / 32 + 2;
    preplanned = c->wplan.valid && stream && !c->hemi && c->wplan.slotbm.size() == nwords;
    if (preplanned) { slotbm.swap(c->wplan.slotbm); wpfx.swap(c->wplan.wpfx); slot_pos_h.swap(c->wplan.slot_pos); c->wplan.valid = false; }
    else window_slots(c, c->hemi, stream, slotbm, wpfx, slot_pos_h);
    auto rank = [&](int64_t p) { const size_t b = (size_t)(p - S.win_start + MKP_SLOTBM_MARGIN);
        return wpfx[b >> 5] + (uint32_t)__builtin_popcount(slotbm[b >> 5] & ((1u << (b & 31)) - 1u)); };
    if (stream) {
      // ---- slot pipeline plan (mkp_slots.hip): global slot numbering, per-read slot ranges and feature-stream offsets, tiles of slots
      const int64_t lo_clamp = (int64_t)S.win_start - MKP_SLOTBM_MARGIN, hi_clamp = (int64_t)S.win_end + MKP_SLOTBM_MARGIN;
      auto crank = [&](int64_t p) { return rank(std::min(std::max(p, lo_clamp), hi_clamp)); };
      const uint32_t total = wpfx[nwords];
      host_parallel(S.hdr.size(), 8192, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) { MkpReadHdr& h = S.hdr[i];
          const uint32_t a = crank(h.ref_start), b = std::max(a, crank(h.ref_end)); h.gs0 = a; h.n_sl = b - a; h.pad = 0; } });
      uint64_t off = 0;
      for (auto& h : S.hdr) {   // (the stream offsets are a running sum: serial, but nothing else is left in the loop)
        h.cov_off = (uint32_t)off;
        off += ((uint64_t)h.n_sl + 3u) & ~3ull;
        if (off > 0xffffff00ull) throw Error(MKP_E_UNSUPPORTED, "shard exceeds 4 GiB of per-read focus positions; use smaller shards");
      }
      c->cov_bytes = off;
      // tile = a run of slots: rows for [g0, g1), tally columns for the slots within MKP_HALO positions of them (strand combining)
      // LDS of mkp_pileup_stream: tallies + per slot its position and an emission word, + the row map (one thread per slot decides the rows)
      const uint32_t Smax = std::min<uint32_t>(MKP_PILEUP_THREADS, ((budget_words - MKP_STREAM_ROWMAP_WORDS) / (words_per_slot + 2u)) & ~63u);
      if (Smax < 128) throw Error(MKP_E_UNSUPPORTED, "too many counters for one LDS tile");
      // as many slots per tile as LDS and the one-thread-per-slot emission allow, down to ~1024 tiles for small windows: a read is visited once per
      // tile it crosses and the visits are most of the kernel (C3, round 6: 448 / 640 / 704 / 896 / 992 slots: 0.153 / 0.139 / 0.128 / 0.125 / 0.116
      // ms)
      uint32_t Te = std::min<uint32_t>(Smax - 2 * MKP_HALO, std::max<uint32_t>(256u, (total / 1024u + 63u) & ~63u));
      // tests: many small tiles
      if (c->cfg.tile_positions) Te = std::max<uint32_t>(32u, std::min<uint32_t>(Smax - 2 * MKP_HALO, c->cfg.tile_positions / 4u));
      if (const char* e = getenv("MKP_STREAM_TILE")) Te = std::max<uint32_t>(32u,
          std::min<uint32_t>(Smax - 2 * MKP_HALO, (uint32_t)strtoul(e, nullptr, 10)));
          // experiments
      uint32_t most = 0;
      for (uint32_t g0 = 0; g0 < total; g0 += Te) {
        MkpSTile t; t.g0 = g0; t.g1 = std::min(total, g0 + Te);
        t.r0 = (int32_t)slot_pos_h[t.g0]; t.r1 = (int32_t)slot_pos_h[t.g1 - 1u] + 1;
        t.gh0 = crank((int64_t)t.r0 - MKP_HALO); t.gh1 = crank((int64_t)t.r1 + MKP_HALO); t.first = t.last = 0;
        stiles.push_back(t); most = std::max(most, t.gh1 - t.gh0);
      }
      if (most > Smax) throw Error(MKP_E_UNSUPPORTED, "internal: slot tile does not fit the LDS budget");
      Scap = std::max<uint32_t>(64u, (most + 63u) & ~63u); Wcap = 0;
    } else {
    // span: enough tiles to balance 512 persistent workgroups, bounded by the bitmap the tile keeps in LDS
    const uint32_t span_max = 32768 - 128;
    // A read is visited once per tile it crosses: longer tiles mean fewer visits (10 kb reads: 1.64 per read at 16 kb, 1.41 at 24 kb),
    // fewer tiles mean a coarser balance over the 512 resident workgroups.  Measured on C3: 2 600 tiles of 24 kb beat 4 100 of 16 kb by 9 %
    // and 7 900 of 8 kb by 31 %.  Small windows keep >= 16 kb tiles (a handful of workgroups of little work each).
    uint32_t span = (uint32_t)std::min<int64_t>(span_max, std::max<int64_t>(16384, ((win / 2600) + 63) & ~63ll));
    if (c->cfg.tile_positions) span = std::max<uint32_t>(64u, std::min(span_max & ~63u, c->cfg.tile_positions & ~63u));
        // explicit tile span (tests: many small tiles; experiments: larger ones)
    Wcap = (span + 2 * MKP_HALO + 31 + 31) / 32 + 2;
    const uint32_t Smax = std::min<uint32_t>(max_slots_for(Wcap), MKP_PILEUP_THREADS);   // row emission of a focus tile: one thread per slot
    if (Smax < 128) throw Error(MKP_E_UNSUPPORTED, "too many counters for one LDS tile");
    uint32_t most = 0;
    for (int64_t r0 = S.win_start; r0 < S.win_end;) {
      int64_t r1 = std::min<int64_t>(r0 + span, S.win_end);
      auto halo_slots = [&](int64_t a, int64_t b) { return rank(std::min<int64_t>(b + MKP_HALO,
          (int64_t)S.win_end + MKP_SLOTBM_MARGIN)) - rank(std::max<int64_t>(a - MKP_HALO, (int64_t)S.win_start - MKP_SLOTBM_MARGIN)); };
      while (halo_slots(r0, r1) > Smax && r1 - r0 > 32) r1 = r0 + std::max<int64_t>(32, (r1 - r0) / 2);   // dense focus: shorter tile
      if (halo_slots(r0, r1) > Smax) throw Error(MKP_E_UNSUPPORTED, "internal: focus tile does not fit the LDS budget");
      if (rank(r1) > rank(r0)) { tiles.push_back({(int32_t)r0, (int32_t)r1, 0, 0}); most = std::max(most, halo_slots(r0, r1));
          }   // a tile without focus positions emits nothing
      r0 = r1;
    }
    Scap = std::max<uint32_t>(64u, (most + 63u) & ~63u);
    }
  }
  lap("slot bitmap + tiles");
  // tile -> [first, last) candidate reads (coordinate sorted; the prefix-max of ends bounds the first candidate)
  {
    std::vector<int32_t> pmax(n); int32_t m = INT32_MIN;
    for (size_t i = 0; i < n; i++) { m = std::max(m, S.hdr[i].ref_end); pmax[i] = m; }
    size_t first = 0, last = 0; std::vector<MkpTile> kept;
    for (auto& tl : tiles) {
      const int64_t lo = (int64_t)tl.r0 - MKP_HALO, hi = (int64_t)tl.r1 + MKP_HALO;
      while (first < n && pmax[first] <= lo) first++;
      if (last < first) last = first;
      while (last < n && S.hdr[last].ref_start < hi) last++;
      bool any = false; for (size_t i = first; i < last && !any; i++) any = S.hdr[i].ref_end > lo;
      if (any) { tl.first = (uint32_t)first; tl.last = (uint32_t)last; kept.push_back(tl); }
    }
    tiles.swap(kept);
  }
  // slot tiles: reads are coordinate sorted, so their first slots ascend; the prefix-max of their slot ends bounds the first candidate
  if (stream) {
    std::vector<uint32_t> pmax(n); uint32_t m = 0;
    for (size_t i = 0; i < n; i++) { m = std::max(m, S.hdr[i].gs0 + S.hdr[i].n_sl); pmax[i] = m; }
    size_t first = 0, last = 0; std::vector<MkpSTile> kept;
    for (auto& tl : stiles) {
      while (first < n && pmax[first] <= tl.gh0) first++;
      if (last < first) last = first;
      while (last < n && S.hdr[last].gs0 < tl.gh1) last++;
      bool any = false; for (size_t i = first; i < last && !any; i++) any = S.hdr[i].gs0 + S.hdr[i].n_sl > tl.gh0 && S.hdr[i].n_sl > 0;
      if (any) { tl.first = (uint32_t)first; tl.last = (uint32_t)last; kept.push_back(tl); }
    }
    stiles.swap(kept);
  }
  // ---- records sharing a name inside an interval (dup_groups, found above): who answers for the name in which interval.
  // The reference asks its per-interval cache about a record the first time the record shows up in a focus column of the interval as
  // something other than a reference skip (add_mod_codes_for_record is called before the deletion check, pileup/mod.rs:783-835); columns
  // ascend, records inside a column come in file order.  The first record asked OWNS the name in that interval: its tags are parsed, its
  // calls (by reference position and read base), its codes and its failure answer for every record of the name (read_cache.rs:232-355).
  std::vector<MkpDupCons> dup_cons; std::vector<MkpDupSeg> dup_segs; std::vector<uint8_t> dup_member(n, 0);
  if (!dup_groups.empty()) {
    auto cigar_of = [&](uint32_t r) {
      const MkpReadHdr& h = S.hdr[r]; std::vector<uint32_t> cg(h.n_cigar);
      if (!S.dev_packed) { for (uint32_t k = 0; k < h.n_cigar; k++) cg[k] = S.cigar[h.cigar_off + k]; }
      else if (h.n_cigar) { hip_check(hipSetDevice(c->device), "hipSetDevice");
        d2h_copy(cg.data(), c->d_cigar.as<uint32_t>() + h.cigar_off, (size_t)h.n_cigar * 4, c->stream); }
      return cg;
    };
    const int64_t W0 = S.win_start, W1 = S.win_end;
    auto iv_bounds = [&](int64_t k, int64_t* a, int64_t* b) {   // interval k of the shard's grid, clipped to the window
      if (c->iv_starts.empty()) { *a = W0; *b = W1; return; }
      *a = std::max<int64_t>(W0, k == 0 ? W0 : (int64_t)c->iv_starts[(size_t)k]);
        *b = (size_t)k + 1 < c->iv_starts.size() ? std::min<int64_t>(W1, (int64_t)c->iv_starts[(size_t)k + 1]) : W1;
    };
    auto iv_index = [&](int64_t p) -> int64_t { if (c->iv_starts.empty()) return 0;
      return std::max<int64_t>((int64_t)(std::upper_bound(c->iv_starts.begin(), c->iv_starts.end(),
          (uint32_t)std::max<int64_t>(p, 0)) - c->iv_starts.begin()) - 1, 0);
        };
    uint64_t ev_off = S.n_events_cap;
    for (auto& g : dup_groups) {
      struct Mem { uint32_t r; int64_t beg, end; std::vector<std::pair<int64_t, int64_t>> skips; };
      std::vector<Mem> ms;
      for (uint32_t r : g) {
        const MkpReadHdr& h = S.hdr[r]; Mem m; m.r = r; m.beg = h.ref_start; m.end = std::max<int64_t>(h.ref_end, (int64_t)h.ref_start + 1);
        if (m.end <= W0 || m.beg >= W1) continue;   // a halo record of the fetch: in none of the shard's intervals
        int64_t p = h.ref_start;
        for (uint32_t w : cigar_of(r)) { const uint32_t op = w & 15u, len = w >> 4; if (op == 3u) m.skips.push_back({p, p + len});
          if (op == 0u || op == 2u || op == 3u || op == 7u || op == 8u) p += len;
          }
        ms.push_back(std::move(m)); dup_member[r] = 1;
      }
      // first column of [lo, hi) in which member m is asked about: a focus position that is not inside one of its reference skips
      auto first_col = [&](const Mem& m, int64_t lo, int64_t hi) -> int64_t {
        lo = std::max(lo, m.beg); hi = std::min(hi, m.end);
        auto in_skip = [&](int64_t p, int64_t* e) {
          for (auto& sk : m.skips) if (p >= sk.first && p < sk.second) { *e = sk.second; return true; }
          return false; };
        if (!c->has_focus) { int64_t p = lo, e; while (p < hi && in_skip(p, &e)) p = e; return p < hi ? p : -1; }
        for (size_t k = (size_t)(std::lower_bound(slot_pos_h.begin(), slot_pos_h.end(),
            (uint32_t)std::max<int64_t>(lo, 0)) - slot_pos_h.begin()); k < slot_pos_h.size() && (int64_t)slot_pos_h[k] < hi; k++) {
          int64_t e; if (!in_skip((int64_t)slot_pos_h[k], &e)) return (int64_t)slot_pos_h[k]; }
        return -1;
      };
      int64_t k_lo = INT64_MAX, k_hi = -1;
      for (auto& m : ms) { k_lo = std::min(k_lo, iv_index(std::max(m.beg, W0))); k_hi = std::max(k_hi, iv_index(std::min(m.end, W1) - 1)); }
      std::vector<std::vector<MkpDupSeg>> segs_of(ms.size());
      for (int64_t k = k_lo; k <= k_hi; k++) {
        int64_t a, b; iv_bounds(k, &a, &b);
        int64_t best_p = -1; size_t best = 0; std::vector<uint8_t> present(ms.size(), 0);
        for (size_t x = 0; x < ms.size(); x++) { const int64_t fc = first_col(ms[x], a, b); if (fc < 0) continue; present[x] = 1;
          if (best_p < 0 || fc < best_p) { best_p = fc; best = x; } }   // (members are in file order: the first of equal columns stays)
        if (best_p < 0) continue;
        for (size_t x = 0; x < ms.size(); x++) if (present[x]) {
          auto& sv = segs_of[x]; const uint32_t owner = ms[best].r;
          if (!sv.empty() && sv.back().owner == owner && sv.back().p_hi == (int32_t)a) sv.back().p_hi = (int32_t)b;
          else sv.push_back({(int32_t)a, (int32_t)b, S.hdr[owner].event_off, owner});
        }
      }
      for (size_t x = 0; x < ms.size(); x++) {
        bool foreign = false; for (auto& sg : segs_of[x]) if (sg.owner != ms[x].r) foreign = true;
        if (!foreign) continue;   // asked about first wherever it shows up: an ordinary record (whose events others may read)
        MkpDupCons dc; memset(&dc, 0, sizeof(dc)); dc.rid = ms[x].r; dc.seg_off = (uint32_t)dup_segs.size(); dc.n_seg = (uint32_t)segs_of[x].size();
          dc.own_off = S.hdr[ms[x].r].event_off;
        uint64_t cap = 0; for (auto& sg : segs_of[x]) { cap += S.hdr[sg.owner].event_cap; dup_segs.push_back(sg); }
        if (ev_off + cap > 0xfffffff0ull) throw Error(MKP_E_UNSUPPORTED, "shard exceeds 4 Gi call events; use smaller shards");
        dc.eff_off = (uint32_t)ev_off; dc.eff_cap = (uint32_t)cap; ev_off += cap;
        dup_cons.push_back(dc);
      }
    }
    S.n_events_cap = ev_off;
    lap("records sharing a name: owners per interval");
  }
  c->n_dup_cons = (uint32_t)dup_cons.size();
  // reads by kernel on the slot pipeline: the fused slot decoder takes the SPARSE classes when no edge filter is set (it never locates
  // calls off the focus positions, which the edge filter's "any call left" test would need); every other read is decoded into events
  // by its class kernel and then covered
  c->read_ids_dec_off = 0; for (auto& x : c->n_slot_class) x = 0;
  std::vector<uint32_t> slot_ids;
  if (stream) {
    const bool fused = !P.edge_filter && !(getenv("MKP_FUSED") && !strcmp(getenv("MKP_FUSED"), "0"));
    uint32_t dup_forced[2] = {0, 0};
    std::vector<uint8_t> is_fused(n, 0);
    if (fused) {
      // (records sharing a name inside an interval keep their event decoder: their events are what the others are answered from; they move
      //  behind the fused reads of the class list, where the decode launch starts)
      if (!dup_groups.empty()) {
        std::vector<uint32_t> keep, f0, f1; const uint32_t n0 = c->n_class[0], n01 = c->n_class[0] + c->n_class[1]; uint32_t k0 = 0;
        for (uint32_t k = 0; k < n01; k++) { const uint32_t r = class_list[k]; if (dup_member[r]) (k < n0 ? f0 : f1).push_back(r); else {
            keep.push_back(r); if (k < n0) k0++; } }
        std::vector<uint32_t> nl(keep); nl.insert(nl.end(), f0.begin(), f0.end()); nl.insert(nl.end(), f1.begin(), f1.end());
          nl.insert(nl.end(), class_list.begin() + n01, class_list.end());
        class_list.swap(nl); c->n_class[0] = k0; c->n_class[1] = (uint32_t)keep.size() - k0; dup_forced[0] = (uint32_t)f0.size();
          dup_forced[1] = (uint32_t)f1.size();
      }
      const uint32_t nf = c->n_class[0] + c->n_class[1];
      for (uint32_t k = 0; k < nf; k++) is_fused[class_list[k]] = 1;
      // one list for both SPARSE classes, longest first; the reads of more than one base window (mkp_decode_slots_long) lead it
      slot_ids.resize(nf);
      auto longer = [&](uint32_t x, uint32_t y) { return S.hdr[x].l_seq > S.hdr[y].l_seq; };
      std::merge(class_list.begin(), class_list.begin() + c->n_class[0], class_list.begin() + c->n_class[0], class_list.begin() + nf,
          slot_ids.begin(), longer);
      uint32_t n_long = 0; while (n_long < nf && S.hdr[slot_ids[n_long]].l_seq > MKP_SLOT_WB) n_long++;
      c->n_slot_class[0] = n_long; c->n_slot_class[1] = nf - n_long;
      c->read_ids_dec_off = nf; c->n_class[0] = dup_forced[0]; c->n_class[1] = dup_forced[1];
    }
    std::vector<uint32_t> rest; rest.reserve(n);
    for (size_t i = 0; i < n; i++) if (!is_fused[i]) rest.push_back((uint32_t)i);
    stable_sort_desc(rest, [&](uint32_t x) { return S.hdr[x].n_sl; });
    c->n_slot_class[2] = (uint32_t)rest.size();
    slot_ids.insert(slot_ids.end(), rest.begin(), rest.end());
  }
  P.slot_cap = Scap; P.focus_words = Wcap;
  c->lds_bytes = stream ? ((words_per_slot + 2u) * Scap + MKP_STREAM_ROWMAP_WORDS) * 4u : MKP_PILEUP_LDS_WORDS(words_per_slot, Scap, Wcap) * 4u;
  c->n_tiles = stream ? (uint32_t)stiles.size() : (uint32_t)tiles.size();
  c->n_slots_total = 0;
  if (c->has_focus) { for (size_t w = 0; w < slotbm.size(); w++) c->n_slots_total += (uint64_t)__builtin_popcount(slotbm[w]); }
  lap("candidate reads per tile");
  c->stats.pack_ms += ms_since(t0);
  auto t1 = std::chrono::steady_clock::now();
  hip_check(hipSetDevice(c->device), "hipSetDevice");
  upload(c->d_hdr, S.hdr);
  if (!S.dev_packed) { upload(c->d_cigar, S.cigar); upload(c->d_chunk, S.chunk_pfx); upload(c->d_seq, S.seq); upload(c->d_tagref, S.tagref);
    upload(c->d_ranks, S.ranks); upload(c->d_ml, S.ml); }
  // (device ingest: those arrays were written in HBM by mkp_ingest_pack and handed over by mkp_internal_shard_attach)
  lap("upload: packed reads");
  upload(c->d_layouts, c->tables.dev); upload(c->d_tiles, tiles);
  upload(c->d_read_ids, class_list);
  if (c->has_focus && preplanned) { /* focus bytes, combos, slot bitmap and slot positions went up with the pre-plan */ }
  else if (c->has_focus) { upload(c->d_focus, c->focus); upload(c->d_combos, c->combos); upload(c->d_slotbm, slotbm); } else { c->d_focus.ensure(16);
    c->d_combos.ensure(64);
      c->d_slotbm.ensure(16); }
  c->d_events.ensure(std::max<uint64_t>(S.n_events_cap, 1) * sizeof(MkpEvent));
  if (c->n_dup_cons) { upload(c->d_dupcons, dup_cons); upload(c->d_dupsegs, dup_segs); }
  c->d_readout.ensure(std::max<size_t>(2 * S.hdr.size(), 1) * sizeof(MkpReadOut));   // second half: second-group summaries of duplex reads
  c->d_misc.ensure(64);
  lap("upload: focus + event buffers");
  if (stream) {
    if (!preplanned) upload(c->d_slot_pos, slot_pos_h);
    upload(c->d_stiles, stiles);
    {   // the fused decoder's work records, in launch order; the cover kernel keeps a read-id list
      const uint32_t nf = c->n_slot_class[0] + c->n_slot_class[1];
      std::vector<MkpWork> work(nf);
      host_parallel(nf, 8192, [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; k++) {
          const MkpReadHdr& h = S.hdr[slot_ids[k]]; MkpWork& w = work[k]; memset(&w, 0, sizeof(w));
          w.ref_start = h.ref_start; w.l_seq = h.l_seq; w.n_cigar = h.n_cigar; w.cigar_off = h.cigar_off; w.seq_off = h.seq_off; w.flags = h.flags;
            w.gs0 = h.gs0;
              w.n_sl = h.n_sl;
          w.cov_off = h.cov_off; w.n_tags = h.n_tags; w.layout = h.layout; w.rid = slot_ids[k];
          if (!(h.flags & MKP_RF_BAD) && h.n_tags) { const MkpTagRef& t0 = S.tagref[h.tag_off]; w.rank_off = t0.rank_off; w.n_calls = t0.n;
            w.ml_off0 = t0.ml_off;
              if (h.n_tags > 1) w.ml_off1 = S.tagref[h.tag_off + 1].ml_off; }
        }
      });
      upload(c->d_work, work);
      std::vector<uint32_t> cover(slot_ids.begin() + nf, slot_ids.end()); upload(c->d_slot_ids, cover);
    }
    { std::vector<MkpFusedDesc> fd(c->tables.dev.size()); for (size_t i = 0; i < fd.size(); i++) fd[i] = fused_desc(c->tables.dev[i]);
      upload(c->d_fdesc, fd); }
    c->d_cov.ensure(c->cov_bytes + 256); c->d_visits.ensure(std::max<size_t>(S.hdr.size(), 1) * sizeof(MkpVisit));
    hip_check(mkp_stream_set_lds(c->lds_bytes), "hipFuncSetAttribute(max dynamic LDS, stream)");
  }
  if (c->hemi) upload(c->d_hemi_iv, c->hemi_iv);
  // partition keys present in this shard: one accumulate pass each
  c->key_passes.clear();
  if (c->partition_tags.empty()) c->key_passes.push_back(MKP_NO_KEY_FILTER);
  else { std::vector<uint8_t> seen(c->key_names.size(), 0); for (auto& h : S.hdr) { const uint32_t k = h.flags >> MKP_RF_KEY_SHIFT;
      if (k < seen.size()) seen[k] = 1;
      } for (uint32_t k = 0; k < seen.size(); k++) if (seen[k]) c->key_passes.push_back(k); if (c->key_passes.empty()) c->key_passes.push_back(0); }
  hip_check(mkp_pileup_set_lds(c->lds_bytes), "hipFuncSetAttribute(max dynamic LDS)");
  hip_check(hipDeviceSynchronize(), "upload sync");
  lap("upload: slot plan + sync");
  c->stats.h2d_ms = ms_since(t1);
  c->resident = true; c->resident_hemi = c->hemi;
  // algorithmic bytes
  uint64_t b_reads = 0; for (auto& h : S.hdr) b_reads += 16 + 4ull * h.n_cigar + (h.l_seq + 1) / 2;
  c->stats.n_reads = S.hdr.size(); c->stats.n_tiles = c->n_tiles; c->stats.n_positions = (uint64_t)win;
  // + 8*events added after the run
  c->stats.alg_bytes_decode = b_reads + (S.dev_packed ? S.dev_n_ranks : S.ranks.size()) * 2ull + (S.dev_packed ? S.dev_n_ml : S.ml.size());
  c->stats.alg_bytes_pileup = b_reads;                                          // + 8*events + 44*rows added after the run
  c->stats.slot_pipeline = stream ? 1u : 0u; c->stats.stream_bytes = 0; c->stats.alg_bytes_agg_survey = 0;
  if (stream) {
    // slot pipeline: the decode side also reads the slot positions of every read's span and writes one feature byte per slot and a
    // 32-byte visit record per read; the aggregation kernel reads exactly those (SEQ and CIGAR are read once per pass)
    uint64_t n_rs = 0; for (auto& h : S.hdr) n_rs += h.n_sl;
    c->stats.stream_bytes = n_rs;
    c->stats.alg_bytes_decode += 4ull * n_rs + n_rs + 32ull * S.hdr.size();
    c->stats.alg_bytes_pileup = n_rs + 32ull * S.hdr.size();               // + 44*rows added after the run
    c->stats.alg_bytes_agg_survey = 8ull * n_rs;                            // 8 B per coverage event at a candidate position (+ call events, + 44*rows)
  }
}

void run_kernels(mkp_ctx* c, bool time_kernels) {
  MkpRunParams& P = c->prm;
  if (c->row_cap == 0) {
    // focus runs: usually one strand rule per focus position and one row per observed code; otherwise two strands per position
    uint64_t guess = c->hemi ? c->n_slots_total * 3 + 4096 : c->has_focus ? c->n_slots_total * std::max<uint32_t>(1u,
        P.numeric_mode == 1 ? P.n_pb : P.n_slots) + 4096 : (uint64_t)c->stats.n_positions * 2 + 1024;
    c->row_cap = std::max<uint64_t>(1u << 16, std::min<uint64_t>(guess * c->key_passes.size(), 1ull << 28));
  }
  const bool trace = time_kernels && getenv("MKP_TRACE_PLAN") != nullptr;
  auto t_rk = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) { if (trace) { auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[mkpileup plan]   kernels: %-20s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_rk).count()); t_rk = now;
    } };
  const uint32_t n_runs = c->n_tiles * (uint32_t)c->key_passes.size();   // row runs: one per (key pass, tile), ordered by key then genome
  // (slot pipeline: row_off holds the runs' 64-bit look-back words behind a 64-byte header — the pass's cursor / total / error words — so that
  //  ONE memset readies a pass: between two kernels of a 1 ms step every extra fill or copy is a 5-10 us launch of its own)
  c->d_tile_row_off.ensure(64 + (size_t)(n_runs + 1) * 8); c->d_tile_row_cnt.ensure((size_t)(n_runs + 1) * 4);
    c->d_tile_dst.ensure((size_t)(n_runs + 1) * 4);
  for (;;) {
    P.row_capacity = (uint32_t)c->row_cap;
    c->rows_src = carve_rows(c->d_rows_src, c->row_cap); c->rows_dst = carve_rows(c->d_rows_dst, c->row_cap);
    lap("row buffers");
    // (trace runs: what the stream still had queued shows up here, not in the kernels' sync)
    if (trace) { hip_check(hipStreamSynchronize(c->stream), "sync"); lap("stream drained"); }
    // [0] row cursor / tile ticket, [1] total rows, [2] error bits; the look-back words (slot pipeline) or row offsets start at +16 dwords
    uint32_t* misc = c->d_tile_row_off.as<uint32_t>();
    uint32_t* row_off = misc + 16;
    // rows leave mkp_pileup_stream in genome order (look-back over the runs): the words start at zero, the number of runs is a kernel argument
    hip_check(hipMemsetAsync(misc, 0, c->slot_mode ? 64 + (size_t)(n_runs + 1) * 8 : 64, c->stream), "memset");
    c->d_prm.ensure(sizeof(MkpRunParams));
    // (re-launches on a resident shard: unchanged)
    if (c->prm_uploaded.size() != sizeof(MkpRunParams) || c->prm_uploaded_to != c->d_prm.p
        || memcmp(c->prm_uploaded.data(), &P, sizeof(MkpRunParams)) != 0) {
      c->prm_uploaded.assign(reinterpret_cast<const uint8_t*>(&P), reinterpret_cast<const uint8_t*>(&P) + sizeof(MkpRunParams));
        c->prm_uploaded_to = c->d_prm.p;
      hip_check(hipMemcpyAsync(c->d_prm.p, c->prm_uploaded.data(), sizeof(MkpRunParams), hipMemcpyHostToDevice, c->stream), "params H2D");
    }
    if (time_kernels) hip_check(hipEventRecord(c->ev[0], c->stream), "event");
    if (c->n_dup_cons) hip_check(mkp_launch_dup_restore(c->stream, c->d_hdr.as<MkpReadHdr>(), c->d_dupcons.as<MkpDupCons>(), c->n_dup_cons),
        "dup restore launch");
    hip_check(mkp_launch_decode(c->stream, c->d_hdr.as<MkpReadHdr>(), c->d_read_ids.as<uint32_t>() + c->read_ids_dec_off, c->n_class,
        c->d_cigar.as<uint32_t>(),
        c->d_seq.as<uint8_t>(), c->d_tagref.as<MkpTagRef>(),
                                c->d_ranks.as<uint32_t>(), c->d_ml.as<uint8_t>(), c->d_layouts.as<MkpLayout>(), &P, c->d_events.as<MkpEvent>(),
                                    c->d_readout.as<MkpReadOut>(), misc + 2, c->d_focus.as<uint8_t>(), nullptr), "decode launch");
    if (c->hemi) hip_check(mkp_launch_hemi_failed(c->stream, c->d_hdr.as<MkpReadHdr>(), c->d_cigar.as<uint32_t>(), c->d_seq.as<uint8_t>(),
        c->d_events.as<MkpEvent>(),
        c->d_readout.as<MkpReadOut>(),
                                                  (uint32_t)c->shard.hdr.size(), c->d_slotbm.as<uint32_t>(), c->d_hemi_iv.as<uint32_t>(),
                                                      (uint32_t)c->hemi_iv.size(), P.win_start, P.win_end, misc + 2), "hemi failed-reads launch");
    // records answered from another record of their name: their event lists are rebuilt from the owners' (behind every event decoder, in front of
    // whatever consumes events: mkp_cover_reads / mkp_pileup_tiles)
    if (c->n_dup_cons) hip_check(mkp_launch_dup_events(c->stream, c->d_hdr.as<MkpReadHdr>(), c->d_cigar.as<uint32_t>(), c->d_seq.as<uint8_t>(),
        c->d_events.as<MkpEvent>(), c->d_readout.as<MkpReadOut>(),
                                                       c->d_dupcons.as<MkpDupCons>(), c->d_dupsegs.as<MkpDupSeg>(), c->n_dup_cons, misc + 2),
                                                           "dup events launch");
    if (c->slot_mode) hip_check(mkp_launch_slots(c->stream, c->d_work.as<MkpWork>(), c->n_slot_class[0], c->n_slot_class[1],
        c->d_hdr.as<MkpReadHdr>(),
        c->d_slot_ids.as<uint32_t>(), c->n_slot_class[2],
                                              c->d_cigar.as<uint32_t>(), c->d_seq.as<uint8_t>(), c->d_tagref.as<MkpTagRef>(),
                                              c->d_ranks.as<uint32_t>(), c->d_ml.as<uint8_t>(), c->d_layouts.as<MkpLayout>(),
                                                  c->d_fdesc.as<MkpFusedDesc>(), &P, c->d_slot_pos.as<uint32_t>(), c->d_cov.as<uint8_t>(),
                                                  c->d_visits.as<MkpVisit>(),
                                              c->d_events.as<MkpEvent>(), c->d_readout.as<MkpReadOut>(), misc + 2), "slot decode launch");
    if (time_kernels) hip_check(hipEventRecord(c->ev[1], c->stream), "event");
    for (uint32_t kp = 0; kp < c->key_passes.size(); kp++)   // one pass per partition key present (a single unfiltered pass without --partition-tag)
      if (c->slot_mode) hip_check(mkp_launch_stream(c->stream, c->lds_bytes, c->d_visits.as<MkpVisit>(), c->d_cov.as<uint8_t>(),
          c->d_events.as<MkpEvent>(),
          c->d_stiles.as<MkpSTile>(), c->n_tiles, c->d_prm.as<MkpRunParams>(),
                                                 c->d_slot_pos.as<uint32_t>(), c->d_focus.as<uint8_t>(), c->d_combos.as<MkpCombo>(), &c->rows_src,
                                                     misc, row_off, c->d_tile_row_cnt.as<uint32_t>(), misc + 2,
                                                 c->key_passes[kp], kp, (uint32_t)c->combos.size(), n_runs, P.slot_cap, P.n_counters + P.n_slots),
                                                     "stream pileup launch");
      else hip_check(mkp_launch_pileup(c->stream, c->lds_bytes, c->hemi ? 2 : c->has_focus ? 1 : 0, c->d_hdr.as<MkpReadHdr>(),
          c->d_cigar.as<uint32_t>(),
          c->d_seq.as<uint8_t>(), c->d_events.as<MkpEvent>(), c->d_readout.as<MkpReadOut>(),
                                  c->d_tiles.as<MkpTile>(), c->n_tiles, c->d_prm.as<MkpRunParams>(), c->d_slotbm.as<uint32_t>(),
                                      c->d_focus.as<uint8_t>(), c->d_combos.as<MkpCombo>(), &c->rows_src, misc,
                                  row_off, c->d_tile_row_cnt.as<uint32_t>(), c->d_chunk.as<uint32_t>(), misc + 2, c->key_passes[kp], kp),
                                      "pileup launch");
    if (time_kernels) hip_check(hipEventRecord(c->ev[2], c->stream), "event");
    if (c->slot_mode) c->rows_dst = c->rows_src;   // already in genome order
    else hip_check(mkp_launch_gather(c->stream, row_off, c->d_tile_row_cnt.as<uint32_t>(), c->d_tile_dst.as<uint32_t>(), n_runs, misc + 1,
        &c->rows_src, &c->rows_dst), "gather launch");
    if (time_kernels && !c->slot_mode) hip_check(hipEventRecord(c->ev[3], c->stream), "event");
    lap("launches");
    if (trace && time_kernels) { hip_check(hipEventSynchronize(c->ev[0]), "sync"); lap("first event reached");
      hip_check(hipEventSynchronize(c->ev[2]), "sync"); lap("kernels done (event)"); }
    if (!c->h_words && hipHostMalloc(reinterpret_cast<void**>(&c->h_words), 64, hipHostMallocDefault) != hipSuccess) { c->h_words = nullptr;
      throw Error(MKP_E_NOMEM, "hipHostMalloc failed"); }
    uint32_t* h = c->h_words;
    hip_check(hipMemcpyAsync(h, misc, 16, hipMemcpyDeviceToHost, c->stream), "D2H");
    hip_check(hipStreamSynchronize(c->stream), "kernel sync");
    lap("sync");
    if (h[2] & 2u) { c->row_cap *= 2; if (c->row_cap > (1ull << 31)) throw Error(MKP_E_NOMEM, "row buffer would exceed 2^31 rows"); continue; }
    if (h[2] & 1u) throw Error(MKP_E_DEVICE, "internal: event segment overflow");
    if (h[2] & ERR_DUP_MIXED) throw Error(MKP_E_UNSUPPORTED,
        "records sharing a read name inside one interval are answered from the first one's calls (the reference's per-interval cache is keyed by name); here the records one of them is answered from in different intervals disagree in status or observed mod codes, which is not reproduced on the device");
    c->stats.n_rows = h[1];
    if (time_kernels) {
      float a = 0, b = 0, d = 0; hip_check(hipEventElapsedTime(&a, c->ev[0], c->ev[1]), "event");
        hip_check(hipEventElapsedTime(&b, c->ev[1], c->ev[2]), "event");
      if (!c->slot_mode) hip_check(hipEventElapsedTime(&d, c->ev[2], c->ev[3]), "event");   // (the slot pipeline has no gather pass)
      c->stats.decode_kernel_ms = a; c->stats.pileup_kernel_ms = b; c->stats.rows_kernel_ms = 0; c->stats.gather_kernel_ms = d;
        c->stats.kernel_ms = a + b + d;
    }
    return;
  }
}

// row columns and per-read outcome counts, device -> host
void fetch_row_columns(mkp_ctx* c) {
  const uint64_t n = c->stats.n_rows;
  const uint32_t* src[11] = {c->rows_dst.pos, c->rows_dst.info, c->rows_dst.code, c->rows_dst.n_valid, c->rows_dst.n_mod, c->rows_dst.n_can,
      c->rows_dst.n_other,
                             c->rows_dst.n_del, c->rows_dst.n_fail, c->rows_dst.n_diff, c->rows_dst.n_nocall};
  c->h_rows.ensure(std::max<uint64_t>(n, 1));
  if (n) { for (int k = 0; k < 11; k++) hip_check(hipMemcpyAsync(c->h_rows.col[k], src[k], n * 4, hipMemcpyDeviceToHost, c->stream), "rows D2H");
    hip_check(hipStreamSynchronize(c->stream), "rows D2H sync"); }
  std::vector<MkpReadOut> ro(c->shard.hdr.size());
  d2h_copy(ro.data(), c->d_readout.p, ro.size() * sizeof(MkpReadOut), c->stream);
  c->n_ok = 0; c->n_bad = 0; uint64_t ev = 0;
  for (auto& r : ro) { if (r.ok) { c->n_ok++; ev += r.n_events; } else c->n_bad++; }
  c->stats.n_events = ev;
}

void fetch_hemi_rows(mkp_ctx* c, mkp_hemi_rows* out) {
  auto t0 = std::chrono::steady_clock::now();
  fetch_row_columns(c);
  const uint64_t n = c->stats.n_rows;
  c->h_hemi_base.resize(n); c->h_hemi_pat[0].resize(n); c->h_hemi_pat[1].resize(n);
  for (uint64_t i = 0; i < n; i++) {   // rows.info = primary base, rows.code = pattern elements a | b << 8
    const uint32_t pb = c->h_rows.col[1][i] & 3u, a = c->h_rows.col[2][i] & 0xffu, b = (c->h_rows.col[2][i] >> 8) & 0xffu;
    c->h_hemi_base[i] = (uint8_t)"ACGT"[pb];
    c->h_hemi_pat[0][i] = a <= MKP_KMAX + 1 ? c->hemi_codes[pb][a] : 0u; c->h_hemi_pat[1][i] = b <= MKP_KMAX + 1 ? c->hemi_codes[pb][b] : 0u;
  }
  c->stats.d2h_ms = ms_since(t0);
  if (out) {
    out->n_rows = n; out->pos = c->h_rows.col[0]; out->primary_base = c->h_hemi_base.data(); out->pattern_pos = c->h_hemi_pat[0].data();
        out->pattern_neg = c->h_hemi_pat[1].data();
    out->n_valid = c->h_rows.col[3]; out->count = c->h_rows.col[4]; out->n_canonical = c->h_rows.col[5]; out->n_other_pattern = c->h_rows.col[6];
    out->n_delete = c->h_rows.col[7]; out->n_fail = c->h_rows.col[8]; out->n_diff = c->h_rows.col[9]; out->n_nocall = c->h_rows.col[10];
    out->processed_records = c->n_ok; out->skipped_records = c->n_bad;
  }
}

void fetch_rows(mkp_ctx* c, mkp_rows* out) {
  auto t0 = std::chrono::steady_clock::now();
  fetch_row_columns(c);
  const uint64_t n = c->stats.n_rows;
  c->h_strand.resize(n); c->h_motif.resize(n); c->h_key.resize(n);
  host_parallel(n, (size_t)1 << 17, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) { const uint32_t inf = c->h_rows.col[1][i];
      c->h_strand[i] = "+-."[inf & 3u]; c->h_motif[i] = (int32_t)((inf >> 8) & 0xffu) - 1;
      c->h_key[i] = inf >> 16; } });
  c->key_name_ptrs.clear(); for (auto& k : c->key_names) c->key_name_ptrs.push_back(k.c_str());
  c->stats.d2h_ms = ms_since(t0);
  if (out) {
    out->n_rows = n; out->pos = c->h_rows.col[0]; out->strand = c->h_strand.data(); out->code_repr = c->h_rows.col[2];
      out->motif_idx = c->h_motif.data();
    out->n_valid = c->h_rows.col[3]; out->n_mod = c->h_rows.col[4]; out->n_canonical = c->h_rows.col[5]; out->n_other = c->h_rows.col[6];
    out->n_delete = c->h_rows.col[7]; out->n_fail = c->h_rows.col[8]; out->n_diff = c->h_rows.col[9]; out->n_nocall = c->h_rows.col[10];
    out->processed_records = c->n_ok; out->skipped_records = c->n_bad;
    out->partition_key = c->h_key.data(); out->n_partition_keys = (uint32_t)c->key_name_ptrs.size();
      out->partition_key_names = c->key_name_ptrs.data();
  }
}

// get_stringable_aux (util.rs:670-688): the value of an aux field as the text the reference partitions by
bool aux_stringable(const mkp_record& r, const char* tag, std::string* out) {
  if (!r.data || r.l_data <= 0) return false;
  const size_t fixed = (size_t)r.l_qname + 4 * (size_t)r.n_cigar + ((size_t)std::max(r.l_qseq, 0) + 1) / 2 + (size_t)std::max(r.l_qseq, 0);
  if (fixed > (size_t)r.l_data) return false;
  const uint8_t* a = r.data + fixed; const uint8_t* e = r.data + r.l_data;
  auto width = [](uint8_t ty) -> int { switch (ty) { case 'A': case 'c': case 'C': return 1; case 's': case 'S': return 2;
      case 'i': case 'I': case 'f': return 4;
      case 'd': return 8; default: return -1; } };
  while (a + 3 <= e) {
    const uint8_t ty = a[2]; const uint8_t* v = a + 3; const uint8_t* nx;
    if (ty == 'Z' || ty == 'H') { const uint8_t* z = v; while (z < e && *z) z++; if (z >= e) return false; nx = z + 1; }
    else if (ty == 'B') { if (v + 5 > e) return false; const int w = width(v[0]); uint32_t cnt; memcpy(&cnt, v + 1, 4);
        if (w < 0 || (uint64_t)cnt * (uint64_t)w > (uint64_t)(e - v - 5)) return false; nx = v + 5 + (size_t)cnt * (size_t)w; }
    else { const int w = width(ty); if (w < 0 || v + w > e) return false; nx = v + w; }
    if (a[0] == (uint8_t)tag[0] && a[1] == (uint8_t)tag[1]) {   // first occurrence (bam_aux_get)
      char buf[64];
      switch (ty) {
        case 'Z': case 'H': out->assign((const char*)v, (size_t)(nx - 1 - v)); return true;
        case 'A': out->assign(1, (char)v[0]); return true;
        case 'c': snprintf(buf, sizeof(buf), "%d", (int)(int8_t)v[0]); break;
        case 'C': snprintf(buf, sizeof(buf), "%u", (unsigned)v[0]); break;
        case 's': { int16_t x; memcpy(&x, v, 2); snprintf(buf, sizeof(buf), "%d", (int)x); break; }
        case 'S': { uint16_t x; memcpy(&x, v, 2); snprintf(buf, sizeof(buf), "%u", (unsigned)x); break; }
        case 'i': { int32_t x; memcpy(&x, v, 4); snprintf(buf, sizeof(buf), "%d", x); break; }
        case 'I': { uint32_t x; memcpy(&x, v, 4); snprintf(buf, sizeof(buf), "%u", x); break; }
        default: return false;   // floats print through Rust's Display (shortest round-trip form): not restated; arrays are not stringable
      }
      *out = buf; return true;
    }
    a = nx;
  }
  return false;
}

template <class F> int guarded(mkp_ctx* c, F f) {
  try { f(); return MKP_OK; }
  catch (const Error& e) { if (c) c->err = e.what(); return e.status; }
  catch (const std::bad_alloc&) { if (c) c->err = "out of host memory"; return MKP_E_NOMEM; }
  catch (const std::exception& e) { if (c) c->err = e.what(); return MKP_E_INVALID; }
}

}  // namespace

extern "C" {

const char* mkp_version(void) { return "libmkpileup 0.1 (gfx950)"; }
unsigned mkp_host_threads(void) { return HostPool::get().size(); }
// test hook (tests/test_host_deflate.py; not part of include/mkpileup.h): the host DEFLATE decoder alone, 1 = decoded, 0 = declined
int mkp_internal_host_inflate(const uint8_t* src, size_t clen, uint8_t* dst, size_t dlen) {
  std::vector<uint8_t> padded(clen + 8, 0); if (clen) memcpy(padded.data(), src, clen);
  return hostinf::inflate(padded.data(), clen, dst, dlen) ? 1 : 0;
}

// test hook (tests/test_host_deflate.py): the CRC-32 every inflate path checks a block's bytes with (mkp_crc32.hpp)
uint32_t mkp_internal_crc32(const uint8_t* p, size_t n) { return crc32_of(p, n); }

int mkp_ctx_create(const mkp_config* cfg, mkp_ctx** out) {
  if (!out) return MKP_E_INVALID;
  *out = nullptr;
  mkp_ctx* c = new (std::nothrow) mkp_ctx();
  if (!c) return MKP_E_NOMEM;
  memset(&c->cfg, 0, sizeof(c->cfg)); if (cfg) c->cfg = *cfg;
  memset(&c->stats, 0, sizeof(c->stats));
  c->device = c->cfg.device;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { delete c; return MKP_E_DEVICE; }
  if (c->device < 0 || c->device >= n) { delete c; return MKP_E_DEVICE; }
  // the context's stream runs the short kernels a caller waits for — sampling rounds, the pileup pass — often beside an ingest that fills the
  // chip for tens of milliseconds: it goes first when workgroup slots free up
  { int plo = 0, phi = 0;
    if (hipSetDevice(c->device) != hipSuccess || hipDeviceGetStreamPriorityRange(&plo, &phi) != hipSuccess
        || pooled_stream_create(&c->stream, c->device, hipStreamDefault, phi) != hipSuccess) {
      delete c; return MKP_E_DEVICE; } c->stream_prio = phi; }
  // timing events between the kernels of a pass: no system-scope fence when they are recorded (its cache write-back and invalidation sat
  // between the decoder and the kernel that reads what it just wrote; nothing on the host looks at device memory through these events)
  for (auto& e : c->ev) if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) { delete c; return MKP_E_DEVICE; }
  c->caller = CallerCfg();
  *out = c;
  return MKP_OK;
}

void mkp_ctx_destroy(mkp_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (DevBuf* b : {&c->d_vals, &c->d_hdr, &c->d_cigar, &c->d_seq, &c->d_tagref, &c->d_ranks, &c->d_ml, &c->d_layouts, &c->d_events, &c->d_readout,
      &c->d_focus, &c->d_combos, &c->d_tiles,
                    &c->d_slotbm, &c->d_prm, &c->d_read_ids, &c->d_chunk, &c->d_store, &c->d_hist0, &c->d_hist1, &c->d_sample_cursor, &c->d_take,
                        &c->d_tile_row_off, &c->d_tile_row_cnt, &c->d_tile_dst, &c->d_misc, &c->d_rows_src, &c->d_rows_dst, &c->d_hemi_iv,
                        &c->d_slot_pos, &c->d_cov, &c->d_visits, &c->d_stiles, &c->d_slot_ids, &c->d_fdesc, &c->d_work, &c->d_zin, &c->d_zout,
                        &c->d_zblk, &c->d_zstat, &c->d_summary, &c->d_bedmask, &c->d_hist64}) b->release();
  mkp_internal_ingest_destroy(c->ingest); c->ingest = nullptr;
  c->h_rows.release();
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  if (c->h_words) (void)hipHostFree(c->h_words);
  pooled_stream_release(c->stream, c->device, hipStreamDefault, c->stream_prio);   // (drained, then parked for the next context: mkp_ctx.hpp)
  delete c;
}

const char* mkp_last_error(const mkp_ctx* c) { return c ? c->err.c_str() : "no context (is a gfx950 device visible?)"; }

int mkp_set_caller(mkp_ctx* c, const mkp_caller* k) {
  if (!c || !k) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (k->numeric_mode > 2) throw Error(MKP_E_INVALID, "numeric_mode must be 0, 1 or 2");
    CallerCfg cc; cc.default_threshold = k->default_threshold;
    for (int b = 0; b < 4; b++) { cc.per_base[b] = k->per_base_threshold[b]; cc.has_per_base[b] = k->has_per_base[b] != 0; }
    for (uint32_t i = 0; i < k->n_per_mod; i++) cc.per_mod[k->per_mod[i].code_repr] = k->per_mod[i].threshold;
    cc.numeric_mode = k->numeric_mode; cc.collapse_code = k->collapse_code; cc.edge = k->edge_filter != 0; cc.edge_start = k->edge_start;
      cc.edge_end = k->edge_end;
    cc.edge_inverted = k->edge_inverted != 0; cc.force_allow = k->force_allow_implicit != 0; cc.combine_strands = k->combine_strands != 0;
    cc.max_depth = k->max_depth ? k->max_depth : 8000;
    c->caller = cc; c->caller_set = true; c->resident = false;
  });
}

int mkp_set_partition_tags(mkp_ctx* c, const char* const* tags, uint32_t n) {
  if (!c || (!tags && n)) return MKP_E_INVALID;
  return guarded(c, [&]() {
    std::vector<std::string> v;
    for (uint32_t i = 0; i < n; i++) { if (!tags[i] || strlen(tags[i]) != 2) throw Error(MKP_E_INVALID, "SAM tags are two characters");
        if (std::find(v.begin(), v.end(), tags[i]) != v.end()) throw Error(MKP_E_INVALID, "partition tag given twice"); v.push_back(tags[i]); }
    c->partition_tags = v; c->resident = false;
  });
}

int mkp_shard_begin(mkp_ctx* c, const mkp_shard* s) {
  if (!c || !s) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (s->end <= s->start) throw Error(MKP_E_INVALID, "empty shard window");
    if ((uint64_t)s->end > 0x7fffffffull) throw Error(MKP_E_UNSUPPORTED, "reference coordinates beyond 2^31");
    c->shard.clear(); c->shard.tid = s->tid; c->shard.win_start = (int32_t)s->start; c->shard.win_end = (int32_t)s->end;
    c->has_focus = s->focus != nullptr;
    if (c->has_focus) {
      { const size_t nf = (size_t)(s->end - s->start); c->focus.resize(nf); const uint8_t* src = s->focus; uint8_t* dst = c->focus.data();
        host_parallel(nf, (size_t)4 << 20, [&](size_t lo, size_t hi) { memcpy(dst + lo, src + lo, hi - lo); }); }
      if (s->n_combos > 64) throw Error(MKP_E_UNSUPPORTED, "more than 64 motif combos");
      c->combos.assign(s->combos, s->combos + s->n_combos);
      if (c->combos.empty()) { mkp_motif_combo z; memset(&z, 0, sizeof(z)); c->combos.push_back(z); }
    } else { c->focus.clear(); c->combos.clear(); }
    c->shard_open = true; c->resident = false; c->row_cap = 0; c->iv_starts.clear(); c->wplan.valid = false;
    c->key_names.assign(1, "ungrouped");
    memset(&c->stats, 0, sizeof(c->stats));
  });
}

// Ahead of the reads (the device ingest of this shard is still running): everything of the plan that depends on the window alone.
}   // extern "C"
int mkp_internal_shard_preplan(mkp_ctx* c) {
  if (!c) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!c->shard_open) throw Error(MKP_E_INVALID, "mkp_shard_begin first");
    c->wplan.valid = false;
    if (!stream_pipeline(c, false)) return;
    window_slots(c, false, true, c->wplan.slotbm, c->wplan.wpfx, c->wplan.slot_pos);
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    upload(c->d_focus, c->focus); upload(c->d_combos, c->combos); upload(c->d_slotbm, c->wplan.slotbm); upload(c->d_slot_pos, c->wplan.slot_pos);
    c->wplan.valid = true;
    // a fresh context's first shard: page-locking the row arena costs ~25 ms per 100 MB — here it hides behind the ingest (the caller is
    // about to wait for it); two rows per focus position is what a two-code run can produce at most per strand rule
    if (c->h_rows.cap == 0 && !c->wplan.slot_pos.empty()) c->h_rows.ensure(std::min<size_t>(c->wplan.slot_pos.size() * 2, (size_t)1 << 26));
  });
}
extern "C" {

// The reference's interval grid inside the open shard (ascending interval starts; the last interval ends with the window): only read by the
// duplicate-name rule of the planner — records with one name matter only when they overlap a common interval.  Without it the shard counts
// as one interval.
int mkp_shard_set_intervals(mkp_ctx* c, const uint32_t* starts, uint32_t n) {
  if (!c || (!starts && n)) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!c->shard_open) throw Error(MKP_E_INVALID, "mkp_shard_begin first");
    c->iv_starts.assign(starts, starts + n);
    if (!std::is_sorted(c->iv_starts.begin(), c->iv_starts.end())) throw Error(MKP_E_INVALID, "interval starts must ascend");
  });
}

int mkp_shard_add_records(mkp_ctx* c, const mkp_record* recs, uint32_t n) {
  if (!c || (!recs && n)) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!c->shard_open) throw Error(MKP_E_INVALID, "mkp_shard_begin first");
    auto t0 = std::chrono::steady_clock::now();
    const int32_t tid = c->shard.tid;
    const size_t hdr_before = c->shard.hdr.size();
    pack_records(c->packer, c->shard, recs, n, [tid](const mkp_record& r) { return r.tid == tid && Packer::keep(r); });
    // supplementary records are not tallied but htslib buffers them (BAM_DEF_MASK lets 0x800 through): their spans count for the max-depth guard
    for (uint32_t i = 0; i < n; i++) {
      const mkp_record& r = recs[i];
      if (r.tid != tid || !(r.flag & 2048) || (r.flag & (4 | 256 | 512 | 1024)) || !r.n_cigar || !r.data
          || (uint64_t)r.l_qname + 4ull * r.n_cigar > (uint64_t)std::max(r.l_data,
          0)) continue;
      int64_t len = 0; for (uint32_t k = 0; k < r.n_cigar; k++) { uint32_t w; memcpy(&w, r.data + r.l_qname + 4 * (size_t)k, 4);
          if ((0x18du >> (w & 15u)) & 1u) len += w >> 4; }
      c->shard.extra_spans.push_back({r.pos, (int32_t)std::min<int64_t>((int64_t)r.pos + std::max<int64_t>(len, 1), INT32_MAX)});
    }
    // PartitionKey per kept record (parse_tags_from_record, pileup/mod.rs:626-643): values joined by '_', "missing" for an absent tag
    if (!c->partition_tags.empty()) {
      size_t at = hdr_before;
      for (uint32_t i = 0; i < n; i++) {
        const mkp_record& r = recs[i];
        if (!(r.tid == tid && Packer::keep(r))) continue;
        if (at >= c->shard.hdr.size()) throw Error(MKP_E_INVALID, "internal: packer dropped a kept record");
        std::string key; bool any = false;
        for (size_t t = 0; t < c->partition_tags.size(); t++) { std::string v; const bool got = aux_stringable(r, c->partition_tags[t].c_str(), &v);
          any |= got;
            if (t) key += '_'; key += got ? v : std::string("missing"); }
        uint32_t id = 0;
        if (any) { auto it = std::find(c->key_names.begin() + 1, c->key_names.end(), key); id = (uint32_t)(it - c->key_names.begin());
            if (it == c->key_names.end()) {
              if (c->key_names.size() >= 65535) throw Error(MKP_E_UNSUPPORTED, "more than 65534 partition keys in one shard");
            c->key_names.push_back(key); } }
        c->shard.hdr[at].flags |= id << MKP_RF_KEY_SHIFT;
        at++;
      }
      if (at != c->shard.hdr.size()) throw Error(MKP_E_INVALID, "internal: packer added records that were not kept");
    }
    c->resident = false; c->row_cap = 0;   // the HBM copy and the tile plan belong to the previous record set
    c->stats.pack_ms += ms_since(t0);
  });
}

}   // extern "C"
namespace {
// the shard's own layout ids -> the context's table (shared with the threshold sampler); once per shard
void adopt_layouts(mkp_ctx* c, DevShard* sh) {
  if (sh->layouts_adopted) return;
  const std::vector<uint16_t> map = c->packer.adopt(sh->layouts);
  for (auto* hv : {&sh->S.hdr, &sh->S.so_hdr}) for (auto& h : *hv) if (h.n_tags && !(h.flags & MKP_RF_BAD)) {
    if (h.layout >= map.size()) throw Error(MKP_E_DEVICE, "internal: device ingest layout id out of range");
    h.layout = map[h.layout]; }
  sh->layouts_adopted = true;
}
}  // namespace
int mkp_internal_sample_bind(mkp_ctx* c, DevShard* sh) {
  if (!c || !sh) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!sh->bound) adopt_layouts(c, sh);
    std::swap(c->shard, sh->S);
    std::swap(c->d_cigar, sh->d_cigar); std::swap(c->d_chunk, sh->d_chunk); std::swap(c->d_seq, sh->d_seq); std::swap(c->d_tagref, sh->d_tagref);
      std::swap(c->d_ranks, sh->d_ranks); std::swap(c->d_ml, sh->d_ml);
    sh->bound = !sh->bound;
    c->shard_open = sh->bound; c->resident = false; c->wplan.valid = false;
    if (sh->bound) for (DevBuf* b : {&c->d_cigar, &c->d_chunk, &c->d_seq, &c->d_tagref, &c->d_ranks, &c->d_ml}) b->ensure(16);
  });
}
// Device ingest hand-over (mkp_ingest_host.cpp): the open shard takes the records the device packed.  Their big arrays are swapped into
// the context's device buffers (what those held goes back with `sh`), the digest becomes the host shard, layout ids are mapped into the
// context's table (shared with the threshold sampler).
int mkp_internal_shard_attach(mkp_ctx* c, DevShard* sh) {
  if (!c || !sh) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!c->shard_open) throw Error(MKP_E_INVALID, "mkp_shard_begin first");
    if (!c->partition_tags.empty()) throw Error(MKP_E_INVALID, "internal: device ingest does not read partition tags");
    if (!c->shard.hdr.empty()) throw Error(MKP_E_INVALID, "internal: the shard already holds host-packed records");
    auto t0 = std::chrono::steady_clock::now();
    if (sh->bound) throw Error(MKP_E_INVALID, "internal: the shard is still bound for sampling");
    adopt_layouts(c, sh);
    const int32_t tid = c->shard.tid, ws = c->shard.win_start, we = c->shard.win_end;
    c->shard = std::move(sh->S); c->shard.tid = tid; c->shard.win_start = ws; c->shard.win_end = we; c->shard.dev_packed = true;
    std::swap(c->d_cigar, sh->d_cigar); std::swap(c->d_chunk, sh->d_chunk); std::swap(c->d_seq, sh->d_seq); std::swap(c->d_tagref, sh->d_tagref);
      std::swap(c->d_ranks, sh->d_ranks);
        std::swap(c->d_ml, sh->d_ml);
    // (an empty shard: the kernels still take valid pointers)
    for (DevBuf* b : {&c->d_cigar, &c->d_chunk, &c->d_seq, &c->d_tagref, &c->d_ranks, &c->d_ml}) b->ensure(16);
    c->resident = false; c->row_cap = 0;
    c->stats.pack_ms += ms_since(t0);
  });
}
extern "C" {

int mkp_shard_run(mkp_ctx* c, mkp_rows* out) {
  if (!c) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!c->shard_open) throw Error(MKP_E_INVALID, "mkp_shard_begin first");
    if (!c->caller_set) throw Error(MKP_E_INVALID, "mkp_set_caller first");
    c->hemi = false;
    const bool trace = getenv("MKP_TRACE_PLAN") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&, last = t0](const char* what) mutable { if (trace) { auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mkpileup plan] %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count()); last = now; } };
    if (!c->resident || c->resident_hemi) { c->row_cap = 0; make_resident(c); }
    lap("run: make_resident");
    run_kernels(c, true);
    lap("run: kernels");
    fetch_rows(c, out);
    lap("run: fetch rows");
    if (c->slot_mode) {   // n_events = call features + events of the reads the event decoders took
      c->stats.alg_bytes_pileup += 44ull * c->stats.n_rows;
      c->stats.alg_bytes_agg_survey += 44ull * c->stats.n_rows;
    } else {
      c->stats.alg_bytes_decode += 8ull * c->stats.n_events;
      c->stats.alg_bytes_pileup += 8ull * c->stats.n_events + 44ull * c->stats.n_rows;   // rows leave the accumulate kernel directly
    }
    c->stats.alg_bytes_rows = 0;
  });
}

// process_region_batch (src/pileup/mod.rs:684-716) as ONE call: the intervals of a MultiChromCoordinates with the records the caller's
// reader fetched for them.  Runs of intervals that follow each other on one contig become one resident shard (one pack, one plan, one
// launch sequence, one read-back); the rows come back per interval.
int mkp_batch_run(mkp_ctx* c, const mkp_shard* ivs, uint32_t n_ivs, const mkp_record* recs, uint32_t n_recs, mkp_rows* out) {
  if (!c || (!ivs && n_ivs) || (!recs && n_recs) || (!out && n_ivs)) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!c->caller_set) throw Error(MKP_E_INVALID, "mkp_set_caller first");
    for (auto& v : c->batch_cols) v.clear();
    c->batch_strand.clear(); c->batch_motif.clear(); c->batch_key.clear();
    struct Slice { uint64_t lo, hi, processed, skipped; }; std::vector<Slice> slice(n_ivs, Slice{0, 0, 0, 0});
    mkp_stats acc; memset(&acc, 0, sizeof(acc));
    for (uint32_t i0 = 0; i0 < n_ivs;) {
      // a group: intervals i0 .. i1-1 follow each other on one contig with the same kind of focus (with partition keys every interval runs alone:
      // rows come grouped by key)
      uint32_t i1 = i0 + 1;
      while (c->partition_tags.empty() && i1 < n_ivs && ivs[i1].tid == ivs[i0].tid && ivs[i1].start == ivs[i1 - 1].end
          && (ivs[i1].focus != nullptr) == (ivs[i0].focus != nullptr) &&
             ivs[i1].combos == ivs[i0].combos && ivs[i1].n_combos == ivs[i0].n_combos && (uint64_t)ivs[i1].end - ivs[i0].start <= (1ull << 27)) i1++;
      mkp_shard sh = ivs[i0]; sh.end = ivs[i1 - 1].end;
      std::vector<uint8_t> focus;
      if (sh.focus && i1 > i0 + 1) { focus.resize((size_t)(sh.end - sh.start));
        for (uint32_t k = i0; k < i1; k++) memcpy(focus.data() + (ivs[k].start - sh.start), ivs[k].focus, (size_t)(ivs[k].end - ivs[k].start));
        sh.focus = focus.data(); }
      int rc = mkp_shard_begin(c, &sh); if (rc != MKP_OK) throw Error(rc, c->err);
      c->iv_starts.clear(); for (uint32_t k = i0; k < i1; k++) c->iv_starts.push_back(ivs[k].start);   // the duplicate-name rule is per interval
      // the group's records: its contig, starting before the window's end (the packer drops the other contigs itself; a record that ends
      // before the window only costs its packing)
      std::vector<mkp_record> mine; mine.reserve(n_recs);
      const int64_t hi = (int64_t)sh.end + MKP_HALO;
      for (uint32_t r = 0; r < n_recs; r++) if (recs[r].tid == sh.tid && (int64_t)recs[r].pos < hi) mine.push_back(recs[r]);
      rc = mkp_shard_add_records(c, mine.data(), (uint32_t)mine.size()); if (rc != MKP_OK) throw Error(rc, c->err);
      mkp_rows rows; memset(&rows, 0, sizeof(rows));
      rc = mkp_shard_run(c, &rows); if (rc != MKP_OK) throw Error(rc, c->err);
      // rows are in genome order (by key first with partition tags: then the group is one interval): cut at the interval ends
      const uint64_t base = c->batch_cols[0].size();
      const uint32_t* src[11] = {rows.pos, nullptr, rows.code_repr, rows.n_valid, rows.n_mod, rows.n_canonical, rows.n_other, rows.n_delete,
          rows.n_fail, rows.n_diff, rows.n_nocall};
      for (int k = 0; k < 11; k++) if (src[k]) c->batch_cols[k].insert(c->batch_cols[k].end(), src[k], src[k] + rows.n_rows);
      c->batch_strand.insert(c->batch_strand.end(), rows.strand, rows.strand + rows.n_rows);
        c->batch_motif.insert(c->batch_motif.end(), rows.motif_idx, rows.motif_idx + rows.n_rows);
      c->batch_key.insert(c->batch_key.end(), rows.partition_key, rows.partition_key + rows.n_rows);
      uint64_t at = 0;
      for (uint32_t k = i0; k < i1; k++) {
        uint64_t e = at; if (i1 == i0 + 1) e = rows.n_rows; else while (e < rows.n_rows && rows.pos[e] < ivs[k].end) e++;
        slice[k] = {base + at, base + e, k == i0 ? rows.processed_records : 0, k == i0 ? rows.skipped_records : 0}; at = e;
      }
      acc.pack_ms += c->stats.pack_ms; acc.h2d_ms += c->stats.h2d_ms; acc.kernel_ms += c->stats.kernel_ms; acc.d2h_ms += c->stats.d2h_ms;
        acc.n_rows += c->stats.n_rows; acc.n_reads += c->stats.n_reads;
      i0 = i1;
    }
    c->key_name_ptrs.clear(); for (auto& k : c->key_names) c->key_name_ptrs.push_back(k.c_str());
    for (uint32_t k = 0; k < n_ivs; k++) {
      mkp_rows& o = out[k]; memset(&o, 0, sizeof(o)); const Slice& sl = slice[k]; const uint64_t lo = sl.lo;
      o.n_rows = sl.hi - sl.lo; o.pos = c->batch_cols[0].data() + lo; o.strand = c->batch_strand.data() + lo;
        o.code_repr = c->batch_cols[2].data() + lo; o.motif_idx = c->batch_motif.data() + lo;
      o.n_valid = c->batch_cols[3].data() + lo; o.n_mod = c->batch_cols[4].data() + lo; o.n_canonical = c->batch_cols[5].data() + lo;
        o.n_other = c->batch_cols[6].data() + lo;
      o.n_delete = c->batch_cols[7].data() + lo; o.n_fail = c->batch_cols[8].data() + lo; o.n_diff = c->batch_cols[9].data() + lo;
        o.n_nocall = c->batch_cols[10].data() + lo;
      o.processed_records = sl.processed; o.skipped_records = sl.skipped;
      o.partition_key = c->batch_key.data() + lo; o.n_partition_keys = (uint32_t)c->key_name_ptrs.size();
        o.partition_key_names = c->key_name_ptrs.data();
    }
    // the batch's sums (the other fields: its last group)
    c->stats.pack_ms = acc.pack_ms; c->stats.h2d_ms = acc.h2d_ms; c->stats.kernel_ms = acc.kernel_ms; c->stats.d2h_ms = acc.d2h_ms;
  });
}

int mkp_hemi_shard_run(mkp_ctx* c, int32_t partner_offset, const uint32_t* interval_starts, uint32_t n_intervals, mkp_hemi_rows* out) {
  if (!c || (n_intervals && !interval_starts)) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!c->shard_open) throw Error(MKP_E_INVALID, "mkp_shard_begin first");
    if (!c->caller_set) throw Error(MKP_E_INVALID, "mkp_set_caller first");
    if (partner_offset < -MKP_HALO || partner_offset > MKP_HALO) throw Error(MKP_E_UNSUPPORTED, "motif longer than the tile halo");
    std::vector<uint32_t> iv(interval_starts, interval_starts + n_intervals);
    if (iv.empty()) iv.push_back((uint32_t)std::max(c->shard.win_start, 0));
    const bool same = c->resident && c->resident_hemi && c->hemi_off == partner_offset && c->hemi_iv == iv;
    c->hemi = true; c->hemi_off = partner_offset;
    if (!same) { c->hemi_iv = std::move(iv); c->row_cap = 0; make_resident(c); }
    run_kernels(c, true);
    fetch_hemi_rows(c, out);
    c->stats.alg_bytes_decode += 8ull * c->stats.n_events;
    c->stats.alg_bytes_pileup += 8ull * c->stats.n_events + 44ull * c->stats.n_rows;
    c->stats.alg_bytes_rows = 0;
  });
}

int mkp_shard_rerun(mkp_ctx* c, uint32_t iters, mkp_rows* out) {
  if (!c) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!c->resident) throw Error(MKP_E_INVALID, "no resident shard: call mkp_shard_run once first");
    if (c->resident_hemi && out) throw Error(MKP_E_INVALID,
        "the resident shard ran as pileup-hemi: pass out = NULL here and read rows with mkp_hemi_shard_run");
    double d = 0, p = 0, g = 0;
    for (uint32_t i = 0; i < iters; i++) { run_kernels(c, true); d += c->stats.decode_kernel_ms; p += c->stats.pileup_kernel_ms;
      g += c->stats.gather_kernel_ms; }
    if (iters) { c->stats.decode_kernel_ms = d / iters; c->stats.pileup_kernel_ms = p / iters; c->stats.rows_kernel_ms = 0;
      c->stats.gather_kernel_ms = g / iters;
        c->stats.kernel_ms = (d + p + g) / iters; }
    if (out) fetch_rows(c, out);
  });
}

uint32_t mkp_abi_version(void) { return MKP_ABI_VERSION; }
size_t mkp_run_report_size(void) { return sizeof(mkp_run_report); }

int mkp_get_stats(const mkp_ctx* c, mkp_stats* out) { if (!c || !out) return MKP_E_INVALID; *out = c->stats; return MKP_OK; }

int mkp_percentile(const float* xs, uint64_t n, float q, float* out) {  // percentile_linear_interp (thresholds.rs:17-38)
  if (!xs || !out || n < 2 || !(q >= 0.0f) || q > 1.0f) return MKP_E_THRESHOLD;
      // negative / NaN quantiles would index out of bounds (Rust's `as usize` saturates; here they are refused)
  if (q == 1.0f) { *out = xs[n - 1]; return MKP_OK; }
  float l = (float)(n - 1), lq = l * q, left = floorf(lq); uint64_t right = std::min<uint64_t>((uint64_t)ceilf(lq), n - 1);
  float g = lq - truncf(lq), a = xs[(uint64_t)left] * (1.0f - g), b = xs[right] * g;
  *out = a + b;
  return MKP_OK;
}

int mkp_bgzf_inflate(mkp_ctx* c, const uint8_t* bgzf, uint64_t n_bytes, const uint8_t** out, uint64_t* out_len, double* kernel_ms) {
  if (!c || (!bgzf && n_bytes) || !out || !out_len) return MKP_E_INVALID;
  return guarded(c, [&]() {
    struct Blk { unsigned long long in_off, out_off; uint32_t in_len, out_len; };   // == MkpBgzfBlock (mkp_inflate_wave4.hip)
    std::vector<Blk> blks; std::vector<uint32_t> crcs; uint64_t o = 0, total = 0;
    while (o < n_bytes) {   // header walk, as in load_bam (mkp_bam.hpp): gzip magic, FEXTRA with a BC subfield holding BSIZE
      if (o + 18 > n_bytes || bgzf[o] != 31 || bgzf[o + 1] != 139 || bgzf[o + 2] != 8 || !(bgzf[o + 3] & 4)) throw Error(MKP_E_IO, "not BGZF");
      uint16_t xlen; memcpy(&xlen, bgzf + o + 10, 2);
      uint64_t x = o + 12; const uint64_t xe = x + xlen; uint32_t bsize = 0; bool found = false;
      if (xe > n_bytes) throw Error(MKP_E_IO, "bad BGZF block");
      while (x + 4 <= xe) { uint16_t sl; memcpy(&sl, bgzf + x + 2, 2); if (bgzf[x] == 'B' && bgzf[x + 1] == 'C' && sl == 2 && x + 6 <= xe) {
          uint16_t b;
          memcpy(&b, bgzf + x + 4, 2); bsize = (uint32_t)b + 1; found = true; } x += 4 + (uint64_t)sl; }
      if (!found || o + bsize > n_bytes || bsize < (uint32_t)xlen + 20u) throw Error(MKP_E_IO, "bad BGZF block");
      uint32_t crc, isize; memcpy(&crc, bgzf + o + bsize - 8, 4); memcpy(&isize, bgzf + o + bsize - 4, 4);
      if (isize > 65536u) throw Error(MKP_E_IO, "BGZF block inflates to more than 64 KiB");
      blks.push_back({o + 12 + xlen, total, bsize - xlen - 20, isize}); crcs.push_back(crc);
      total += isize; o += bsize;
    }
    if (blks.size() > 0xffffffffull) throw Error(MKP_E_UNSUPPORTED, "too many BGZF blocks");
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    c->d_zin.ensure(std::max<uint64_t>(n_bytes, 16)); c->d_zout.ensure(std::max<uint64_t>(total, 16));
      c->d_zblk.ensure(std::max<size_t>(blks.size(), 1) * sizeof(Blk));
        c->d_zstat.ensure(std::max<size_t>(blks.size(), 1) * 4);
    h2d_copy(c->d_zin.p, bgzf, n_bytes);   // (caller memory and a temporary: through the library's page-locked staging, mkp_ctx.hpp)
    h2d_copy(c->d_zblk.p, blks.data(), blks.size() * sizeof(Blk));
    hip_check(hipMemsetAsync(c->d_zstat.p, 0xff, std::max<size_t>(blks.size(), 1) * 4, c->stream), "memset");
    hip_check(hipEventRecord(c->ev[0], c->stream), "event");
    hip_check(launch_inflate(c->stream, c->d_zin.as<uint8_t>(), c->d_zblk.p, (uint32_t)blks.size(), c->d_zout.as<uint8_t>(),
        c->d_zstat.as<uint32_t>()),
        "inflate launch");
    hip_check(hipEventRecord(c->ev[1], c->stream), "event");
    std::vector<uint32_t> st(blks.size());
    c->h_inflated.resize(total);
    d2h_copy(st.data(), c->d_zstat.p, blks.size() * 4, c->stream);
    d2h_copy(c->h_inflated.data(), c->d_zout.p, total, c->stream);
    hip_check(hipStreamSynchronize(c->stream), "inflate sync");
    if (kernel_ms) { float ms = 0; hip_check(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]), "event"); *kernel_ms = ms; }
    std::atomic<long> bad{-1};
    host_parallel(blks.size(), 64, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; i++) {
        const bool ok = st[i] == 0 && (uint32_t)crc32(crc32(0L, Z_NULL, 0), c->h_inflated.data() + blks[i].out_off, blks[i].out_len) == crcs[i];
        if (!ok) { long exp = -1; bad.compare_exchange_strong(exp, (long)i); }
      }
    });
    if (bad.load() >= 0) throw Error(MKP_E_IO,
        "corrupt BGZF data: block " + std::to_string(bad.load()) + " (decoder status " + std::to_string(st[(size_t)bad.load()]) + ")");
    *out = c->h_inflated.data(); *out_len = total;
  });
}

// ---- --device-inflate: the fetch path's inflate stage on the GPU (BamSource::dev_inflate).  Own stream and buffers: it runs on the
// prefetch thread while the shard in hand uses the context's stream.
struct PinnedBuf {   // page-locked host staging: pageable copies of a window's 270 MB ran at under 3 GB/s
  void* p = nullptr; size_t cap = 0;
  void ensure(size_t n) { if (n <= cap) return; release(); const size_t want = n + n / 8 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr;
      throw Error(MKP_E_NOMEM, "hipHostMalloc failed"); } cap = want; }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};
struct mkp_dev_inflater { int device = 0; hipStream_t stream = nullptr; DevBuf zin, zout, zblk, zstat; PinnedBuf pin_in, pin_out; std::mutex mu; };
}   // extern "C"
mkp_dev_inflater* mkp_internal_inflater_create(int device) {
  std::unique_ptr<mkp_dev_inflater> d(new mkp_dev_inflater()); d->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
  return d.release();
}
void mkp_internal_inflater_destroy(mkp_dev_inflater* d) {
  if (!d) return;
  (void)hipSetDevice(d->device); d->zin.release(); d->zout.release(); d->zblk.release(); d->zstat.release(); d->pin_in.release();
    d->pin_out.release();
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}
bool mkp_internal_device_inflate(void* user, const InflateJob& j) {
  mkp_dev_inflater* d = (mkp_dev_inflater*)user;
  if (!d || j.n_blks == 0 || j.n_blks > 0xffffffffull) return false;
  std::lock_guard<std::mutex> g(d->mu);
  try {
    auto ok = [](hipError_t e) { if (e != hipSuccess) throw Error(MKP_E_DEVICE, hipGetErrorString(e)); };
    ok(hipSetDevice(d->device));
    d->zin.ensure(j.comp_len + 16); d->zout.ensure(j.dtotal + 16); d->zblk.ensure(j.n_blks * sizeof(InflateBlk)); d->zstat.ensure(j.n_blks * 4);
    d->pin_in.ensure(j.comp_len + j.n_blks * sizeof(InflateBlk)); d->pin_out.ensure(j.dtotal + j.n_blks * 4);
    // file pages -> pinned (all cores, behind the foreground work), one H2D; inflate; one D2H into pinned; pinned -> the window's buffer
    uint8_t* pin = (uint8_t*)d->pin_in.p; const size_t piece = (size_t)4 << 20;
    HostPool::get().parallel((j.comp_len + piece - 1) / piece, [&](size_t i) { const size_t lo = i * piece, n = std::min(piece, j.comp_len - lo);
      memcpy(pin + lo, j.comp + lo, n); });
    memcpy(pin + j.comp_len, j.blks, j.n_blks * sizeof(InflateBlk));
    ok(hipMemcpyAsync(d->zin.p, pin, j.comp_len, hipMemcpyHostToDevice, d->stream));
    ok(hipMemcpyAsync(d->zblk.p, pin + j.comp_len, j.n_blks * sizeof(InflateBlk), hipMemcpyHostToDevice, d->stream));
    ok(hipMemsetAsync(d->zstat.p, 0xff, j.n_blks * 4, d->stream));
    ok(launch_inflate(d->stream, d->zin.as<uint8_t>(), d->zblk.p, (uint32_t)j.n_blks, d->zout.as<uint8_t>(), d->zstat.as<uint32_t>()));
    uint8_t* pout = (uint8_t*)d->pin_out.p;
    ok(hipMemcpyAsync(pout, d->zout.p, j.dtotal, hipMemcpyDeviceToHost, d->stream));
    ok(hipMemcpyAsync(pout + j.dtotal, d->zstat.p, j.n_blks * 4, hipMemcpyDeviceToHost, d->stream));
    ok(hipStreamSynchronize(d->stream));
    const uint32_t* st = (const uint32_t*)(pout + j.dtotal);   // (dtotal is a sum of block sizes; the status words may sit unaligned)
    // the host decoder takes the window and names the error
    for (size_t i = 0; i < j.n_blks; i++) { uint32_t v; memcpy(&v, (const uint8_t*)st + 4 * i, 4); if (v != 0) return false; }
    HostPool::get().parallel((j.dtotal + piece - 1) / piece, [&](size_t i) { const size_t lo = i * piece, n = std::min(piece, j.dtotal - lo);
      memcpy(j.dst + lo, pout + lo, n); });
    // the blocks' CRC32s (trailer word behind each payload), as the host decoder checks them: a mismatch hands the window to the host path,
    // which names the error
    std::atomic<bool> crc_bad{false};
    HostPool::get().parallel((j.n_blks + 63) / 64, [&](size_t g) { for (size_t i = g * 64; i < std::min(j.n_blks, (g + 1) * 64); i++) {
        const InflateBlk& b = j.blks[i];
        uint32_t want; memcpy(&want, j.comp + b.in_off + b.in_len, 4); if (crc32_of(j.dst + b.out_off, b.out_len) != want) crc_bad = true; } });
    if (crc_bad) return false;
    return true;
  } catch (const Error&) { return false; }
}
extern "C" {

int mkp_host_mm_ranks(const char* mm, uint32_t l_seq, uint32_t n_ml, mkp_host_tag* tags, uint32_t tags_cap, uint32_t* ranks, uint32_t ranks_cap) {
  if (!mm || (!tags && tags_cap) || (!ranks && ranks_cap)) return MKP_E_INVALID;
  try {
    // a synthetic one-record shard: qname "r", one M op, all-A SEQ, aux = MM:Z + ML:B:C (n_ml zero bytes)
    std::vector<uint8_t> data; data.push_back('r'); data.push_back(0);
    const uint32_t cg = (l_seq << 4) | 0u; data.insert(data.end(), (const uint8_t*)&cg, (const uint8_t*)&cg + 4);
    data.insert(data.end(), (l_seq + 1) / 2, 0x11); data.insert(data.end(), l_seq, 0xff);
    data.push_back('M'); data.push_back('M'); data.push_back('Z'); data.insert(data.end(), mm, mm + strlen(mm) + 1);
    data.push_back('M'); data.push_back('L'); data.push_back('B'); data.push_back('C');
      data.insert(data.end(), (const uint8_t*)&n_ml, (const uint8_t*)&n_ml + 4);
        data.insert(data.end(), n_ml, 0);
    mkp_record r; memset(&r, 0, sizeof(r)); r.tid = 0; r.pos = 0; r.l_qname = 2; r.n_cigar = 1; r.l_qseq = (int32_t)l_seq;
      r.l_data = (int32_t)data.size();
        r.data = data.data();
    Packer pk; ShardHost S; S.tid = 0; pk.add(r, S);
    if (S.hdr.empty() || (S.hdr[0].flags & MKP_RF_BAD)) return MKP_E_INVALID;
    const LayoutHost& L = pk.layouts[S.hdr[0].layout];
    if (L.tags.size() > tags_cap || S.ranks.size() > ranks_cap) return MKP_E_NOMEM;
    for (size_t t = 0; t < L.tags.size(); t++) {
      mkp_host_tag& o = tags[t]; memset(&o, 0, sizeof(o));
      o.base = L.tags[t].fb; o.negative_strand = L.tags[t].neg; o.mode = L.tags[t].mode; o.n_codes = (uint8_t)L.tags[t].codes.size();
      for (size_t i = 0; i < L.tags[t].codes.size() && i < 4; i++) o.codes[i] = L.tags[t].codes[i];
      o.rank_off = S.tagref[t].rank_off; o.n_ranks = S.tagref[t].n;
    }
    for (size_t i = 0; i < S.ranks.size(); i++) ranks[i] = S.ranks[i];
    return (int)L.tags.size();
  } catch (const Error& e) { return e.status; } catch (...) { return MKP_E_INVALID; }
}

int mkp_host_map_order(const uint32_t* code_reprs, uint32_t n, uint32_t* order_out) {
  if (!code_reprs || !order_out || n > 14) return MKP_E_INVALID;
  try { FxOrder m; for (uint32_t i = 0; i < n; i++) m.insert(code_reprs[i], (int)i); auto c = m.codes();
    for (size_t i = 0; i < c.size(); i++) order_out[i] = c[i];
      return (int)c.size(); }
  catch (...) { return MKP_E_INVALID; }
}

}  // extern "C"

// ---- threshold sampling on the device (decode kernels in sampling mode; the values stay in HBM)
extern "C" {
hipError_t mkp_launch_sample_accumulate(hipStream_t, const MkpReadHdr*, const MkpReadOut*, const uint8_t*, uint32_t, const float*, const MkpEvent*,
    uint32_t*,
    unsigned long long, unsigned long long*, uint32_t*, uint32_t*);
hipError_t mkp_launch_sample_hist1(hipStream_t, const uint32_t*, unsigned long long, uint32_t, uint32_t, uint32_t*);
hipError_t mkp_launch_summary_accumulate(hipStream_t, const MkpReadHdr*, const MkpReadOut*, const uint8_t*, uint32_t, const MkpEvent*,
    unsigned long long*,
    unsigned long long*);
}

// Decode `recs` in sampling mode.  n_vals[i] = number of argmax probabilities record i yields after the filters (0: rejected or
// nothing kept).  The values themselves stay on the device until mkp_internal_sample_take says which reads the schedule took.
namespace {
// the sampling pass over the packed reads of S (host-packed: everything goes up; resident: the arrays of the attached shard are in HBM
// already and S.hdr is the round's selection of its headers)
void sample_decode(mkp_ctx* c, ShardHost& S, bool resident, const uint8_t* bedmask, bool only_mapped, size_t n_expected,
    std::vector<uint32_t>* n_vals) {
  c->tables.build(c->packer.layouts, c->caller);
  MkpRunParams P; memset(&P, 0, sizeof(P));
  P.win_start = S.win_start; P.win_end = S.win_end; P.numeric_mode = c->caller.numeric_mode; P.edge_filter = c->caller.edge;
    P.edge_start = c->caller.edge_start;
  P.edge_end = c->caller.edge_end; P.edge_inverted = c->caller.edge_inverted; P.force_allow = 1;
    P.sample_mode = c->extract_mode ? 3 : c->summary_mode ? 2 : 1; P.only_mapped = only_mapped;
      P.has_focus = bedmask != nullptr;
  hip_check(hipSetDevice(c->device), "hipSetDevice");
  upload(c->d_hdr, S.hdr);
  if (!resident) { upload(c->d_cigar, S.cigar); upload(c->d_seq, S.seq); upload(c->d_tagref, S.tagref); upload(c->d_ranks, S.ranks);
    upload(c->d_ml, S.ml); }
  upload(c->d_layouts, c->tables.dev);
  { std::vector<uint32_t> ids; class_ids(S, c->tables, &ids, c->n_class, false); upload(c->d_read_ids, ids); }
  // the --include-bed mask of the contig: one upload per contig and sampling session, not per round (a 25 MB mask per round was most of
  // the 11.6 s a BED-filtered threshold estimate took on the C5 scale model); mkp_internal_bedmask_reset() starts a session
  const uint8_t* d_mask;
  if (bedmask) {
    const size_t len = (size_t)(S.win_end - S.win_start);
    if (bedmask != c->bedmask_src || len != c->bedmask_len) {
      c->d_bedmask.ensure(len); h2d_copy(c->d_bedmask.p, bedmask, len); c->bedmask_src = bedmask; c->bedmask_len = len;
    }
    d_mask = c->d_bedmask.as<uint8_t>();
  } else { c->d_focus.ensure(16); d_mask = c->d_focus.as<uint8_t>(); }
  const uint64_t cap = std::max<uint64_t>(S.n_events_cap, 1);
  c->d_events.ensure(cap * sizeof(MkpEvent)); c->d_vals.ensure(cap * sizeof(float));
    c->d_readout.ensure(std::max<size_t>(S.hdr.size(), 1) * sizeof(MkpReadOut));
      c->d_misc.ensure(64);
  uint32_t* misc = c->d_misc.as<uint32_t>();
  hip_check(hipMemsetAsync(misc, 0, 16, c->stream), "memset");
  hip_check(mkp_launch_decode(c->stream, c->d_hdr.as<MkpReadHdr>(), c->d_read_ids.as<uint32_t>(), c->n_class, c->d_cigar.as<uint32_t>(),
      c->d_seq.as<uint8_t>(),
      c->d_tagref.as<MkpTagRef>(), c->d_ranks.as<uint32_t>(),
                              c->d_ml.as<uint8_t>(), c->d_layouts.as<MkpLayout>(), &P, c->d_events.as<MkpEvent>(), c->d_readout.as<MkpReadOut>(),
                                  misc + 2, d_mask, c->d_vals.as<float>()), "decode(sample) launch");
  uint32_t h[4]; hip_check(hipMemcpyAsync(h, misc, 16, hipMemcpyDeviceToHost, c->stream), "D2H");
  c->sample_ro.resize(S.hdr.size());
  if (!S.hdr.empty()) hip_check(hipMemcpyAsync(c->sample_ro.data(), c->d_readout.p, S.hdr.size() * sizeof(MkpReadOut), hipMemcpyDeviceToHost,
      c->stream), "D2H");
  hip_check(hipStreamSynchronize(c->stream), "sample sync");
  if (h[2] & 1u) throw Error(MKP_E_DEVICE, "internal: event segment overflow");
  if (S.hdr.size() != n_expected) throw Error(MKP_E_INVALID, "internal: sampler packed a different number of records");
  n_vals->resize(n_expected);
  for (size_t i = 0; i < n_expected; i++) (*n_vals)[i] = c->sample_ro[i].ok ? c->sample_ro[i].n_events : 0u;
  c->resident = false;
}
}  // namespace

int mkp_internal_sample(mkp_ctx* c, int32_t tid, uint32_t win_start, uint32_t win_end, const uint8_t* bedmask, const mkp_record* recs,
                        uint32_t n, bool only_mapped, std::vector<uint32_t>* n_vals) {
  if (!c || !n_vals) return MKP_E_INVALID;
  return guarded(c, [&]() {
    // a device-packed shard of an earlier run: its HBM arrays are about to be reused
    if (c->shard.dev_packed) { c->shard.clear(); c->shard_open = false; c->resident = false; }
    ShardHost& S = c->sample_shard; S.clear(); S.tid = tid; S.win_start = (int32_t)win_start; S.win_end = (int32_t)win_end;
    pack_records(c->packer, S, recs, n, [](const mkp_record&) { return true; });
    sample_decode(c, S, false, bedmask, only_mapped, n, n_vals);
  });
}

// The same pass over reads of the shard the device ingest attached (mkp_internal_shard_attach): `reads` index that shard's records; their
// CIGARs, bases and tags are in HBM already, only the round's headers (with their slices of the event buffer) go up.
int mkp_internal_sample_resident(mkp_ctx* c, uint32_t win_start, uint32_t win_end, const uint8_t* bedmask, const uint32_t* reads, uint32_t n,
    bool only_mapped, std::vector<uint32_t>* n_vals) {
  if (!c || !n_vals || (!reads && n)) return MKP_E_INVALID;
  return guarded(c, [&]() {
    ShardHost& R = c->shard;
    if (!c->shard_open || !R.dev_packed) throw Error(MKP_E_INVALID, "internal: no device-packed shard attached");
    ShardHost& S = c->sample_shard; S.clear(); S.tid = R.tid; S.win_start = (int32_t)win_start; S.win_end = (int32_t)win_end; S.dev_packed = true;
    S.hdr.resize(n); uint64_t off = 0;
    for (uint32_t k = 0; k < n; k++) {
      if (reads[k] >= R.hdr.size() + R.so_hdr.size()) throw Error(MKP_E_INVALID, "internal: sampled read index out of range");
      MkpReadHdr h = reads[k] < R.hdr.size() ? R.hdr[reads[k]] : R.so_hdr[reads[k] - R.hdr.size()]; h.event_off = (uint32_t)off; off += h.event_cap;
      if (off > 0xfffffff0ull) throw Error(MKP_E_UNSUPPORTED, "sampling round exceeds 4 Gi call events");
      S.hdr[k] = h;
    }
    S.n_events_cap = off;
    S.tagref.swap(R.tagref);   // the headers' tag_off index the shard's tag table
    struct Back { ShardHost& a; ShardHost& b; ~Back() { a.tagref.swap(b.tagref); } } back{S, R};
    sample_decode(c, S, true, bedmask, only_mapped, n, n_vals);
  });
}

void mkp_internal_bedmask_reset(mkp_ctx* c) { if (c) { c->bedmask_src = nullptr; c->bedmask_len = 0; } }

int mkp_internal_set_extract(mkp_ctx* c, bool on) {
  if (!c) return MKP_E_INVALID;
  c->extract_mode = on; c->caller.read_base_caller = on; c->resident = false;
  return MKP_OK;
}

int mkp_internal_extract_fetch(mkp_ctx* c, std::vector<MkpEvent>* events, std::vector<float>* vals) {
  if (!c || !events || !vals) return MKP_E_INVALID;
  return guarded(c, [&]() {
    const uint64_t n = c->sample_shard.n_events_cap;
    events->resize(n); vals->resize(n);
    if (!n) return;
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    d2h_copy(events->data(), c->d_events.p, n * sizeof(MkpEvent), c->stream);
    d2h_copy(vals->data(), c->d_vals.p, n * sizeof(float), c->stream);
  });
}

// Add the values of the marked records of the last mkp_internal_sample batch to the resident sample and its level-0 histogram.
int mkp_internal_summary_begin(mkp_ctx* c) {
  if (!c) return MKP_E_INVALID;
  return guarded(c, [&]() {
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    c->d_summary.ensure(134 * 8);
    hip_check(hipMemsetAsync(c->d_summary.p, 0, 134 * 8, c->stream), "memset");
    hip_check(hipStreamSynchronize(c->stream), "sync");
  });
}
int mkp_internal_summary_get(mkp_ctx* c, uint64_t out[134], std::vector<MkpSlot>* slots) {
  if (!c || !out || !slots) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!c->d_summary.p) throw Error(MKP_E_INVALID, "internal: summary table not set up");
    hip_check(hipMemcpy(out, c->d_summary.p, 134 * 8, hipMemcpyDeviceToHost), "D2H");
    *slots = c->tables.st.slots;   // class 2 + s = Modified(slots[s].code_repr) on primary base slots[s].pb
  });
}
int mkp_internal_sample_take(mkp_ctx* c, const std::vector<uint8_t>& take) {
  if (!c) return MKP_E_INVALID;
  return guarded(c, [&]() {
    const size_t n = c->sample_shard.hdr.size();
    if (take.size() != n) throw Error(MKP_E_INVALID, "internal: take mask size");
    uint64_t add = 0; for (size_t i = 0; i < n; i++) if (take[i] && c->sample_ro[i].ok) add += c->sample_ro[i].n_events;
    if (!add) return;
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    if (c->summary_mode) {   // `modkit summary`: count the taken reads' calls
      if (!c->d_summary.p) throw Error(MKP_E_INVALID, "internal: summary table not set up");
      c->d_take.ensure(std::max<size_t>(n, 16));
      hip_check(hipMemcpyAsync(c->d_take.p, take.data(), n, hipMemcpyHostToDevice, c->stream), "H2D");
      hip_check(mkp_launch_summary_accumulate(c->stream, c->d_hdr.as<MkpReadHdr>(), c->d_readout.as<MkpReadOut>(), c->d_take.as<uint8_t>(),
          (uint32_t)n,
          c->d_events.as<MkpEvent>(),
                                              c->d_summary.as<unsigned long long>(), c->d_summary.as<unsigned long long>() + 128),
                                                  "summary accumulate launch");
      hip_check(hipStreamSynchronize(c->stream), "summary accumulate sync");
      return;
    }
    if (!c->d_hist0.p) { c->d_hist0.ensure(4 * 65536 * 4); c->d_hist1.ensure(65536 * 4); c->d_sample_cursor.ensure(16);
        hip_check(hipMemsetAsync(c->d_hist0.p, 0, 4 * 65536 * 4, c->stream), "memset");
          hip_check(hipMemsetAsync(c->d_sample_cursor.p, 0, 16, c->stream), "memset"); }
    const uint64_t need = c->sample_n + add;
    // the device histograms count in 32 bits: with 2^32 or more sampled values one bin could wrap (ML bytes put most values on a handful
    // of f32 patterns) — refused rather than estimated wrongly
    if (need > 0xffffffffull) throw Error(MKP_E_UNSUPPORTED,
        "more than 2^32 - 1 sampled probabilities on one GPU (32-bit histogram counters); shard the estimate over more ranks");
    if (need * 4 > c->d_store.cap) {   // grow the resident sample, keeping what is there
      DevBuf nb; nb.ensure(std::max<uint64_t>(need * 4 * 2, 1u << 20));
      if (c->sample_n) hip_check(hipMemcpyAsync(nb.p, c->d_store.p, c->sample_n * 4, hipMemcpyDeviceToDevice, c->stream), "D2D");
      hip_check(hipStreamSynchronize(c->stream), "sync"); c->d_store.release(); c->d_store = nb; nb.p = nullptr; nb.cap = 0;
    }
    c->d_take.ensure(std::max<size_t>(n, 16));
    hip_check(hipMemcpyAsync(c->d_take.p, take.data(), n, hipMemcpyHostToDevice, c->stream), "H2D");
    uint32_t* misc = c->d_misc.as<uint32_t>();
    hip_check(mkp_launch_sample_accumulate(c->stream, c->d_hdr.as<MkpReadHdr>(), c->d_readout.as<MkpReadOut>(), c->d_take.as<uint8_t>(), (uint32_t)n,
        c->d_vals.as<float>(), c->d_events.as<MkpEvent>(),
                                           c->d_store.as<uint32_t>(), c->d_store.cap / 4, c->d_sample_cursor.as<unsigned long long>(),
                                               c->d_hist0.as<uint32_t>(), misc + 2), "sample accumulate launch");
    hip_check(hipStreamSynchronize(c->stream), "sample accumulate sync");
    c->sample_n = need;
  });
}

namespace {
void require_sample(mkp_ctx* c) { if (!c->d_hist0.p) { hip_check(hipSetDevice(c->device), "hipSetDevice"); c->d_hist0.ensure(4 * 65536 * 4);
    c->d_hist1.ensure(65536 * 4);
    c->d_sample_cursor.ensure(16); hip_check(hipMemset(c->d_hist0.p, 0, 4 * 65536 * 4), "memset");
      hip_check(hipMemset(c->d_sample_cursor.p, 0, 16), "memset"); } }
}

extern "C" {

int mkp_histogram_begin(mkp_ctx* c) {
  if (!c) return MKP_E_INVALID;
  return guarded(c, [&]() {
    require_sample(c);
    hip_check(hipMemset(c->d_hist0.p, 0, 4 * 65536 * 4), "memset"); hip_check(hipMemset(c->d_sample_cursor.p, 0, 16), "memset");
    c->sample_n = 0;
  });
}

int mkp_histogram_get(mkp_ctx* c, uint32_t base, uint32_t level, uint32_t prefix, uint64_t* out) {
  if (!c || !out || base > 3 || level > 1 || prefix > 0xffffu) return MKP_E_INVALID;
  return guarded(c, [&]() {
    require_sample(c);
    std::vector<uint32_t> h(65536);
    if (level == 0) hip_check(hipMemcpy(h.data(), c->d_hist0.as<uint32_t>() + (size_t)base * 65536, 65536 * 4, hipMemcpyDeviceToHost), "D2H");
    else {
      hip_check(hipMemsetAsync(c->d_hist1.p, 0, 65536 * 4, c->stream), "memset");
      hip_check(mkp_launch_sample_hist1(c->stream, c->d_store.as<uint32_t>(), c->sample_n, base, prefix, c->d_hist1.as<uint32_t>()), "hist1 launch");
      hip_check(hipMemcpyAsync(h.data(), c->d_hist1.p, 65536 * 4, hipMemcpyDeviceToHost, c->stream), "D2H");
      hip_check(hipStreamSynchronize(c->stream), "hist1 sync");
    }
    for (size_t i = 0; i < 65536; i++) out[i] = h[i];
  });
}

// The path's one collective behind the C ABI: the histogram is summed over the ranks of an RCCL communicator where it sits — widened to
// u64 in HBM, ncclAllReduce(ncclUint64, ncclSum) over xGMI on the context's stream, one D2H of the result.  librccl is looked up at
// run time (dlopen): a single-GPU build of the caller needs no RCCL.
}  // extern "C"
#include <dlfcn.h>
extern "C" hipError_t mkp_launch_widen(hipStream_t, const uint32_t*, unsigned long long*, uint32_t);
namespace {
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
// the HIP runtime this library is bound to (its device pointers mean something to that copy only): a process that also imports torch
// can hold a second copy of the ROCm libraries, with a librccl of its own next to it
std::string hip_runtime_path() { Dl_info i; if (dladdr((void*)&hipGetDeviceCount, &i) && i.dli_fname) return std::string(i.dli_fname);
  return std::string(); }
nccl_allreduce_fn rccl_allreduce() {
  static nccl_allreduce_fn fn = []() -> nccl_allreduce_fn {
    const char* forced = getenv("MKP_RCCL_LIB");   // (the library the caller made its communicator with)
    std::string beside = hip_runtime_path(); { const size_t sl = beside.rfind('/');
      beside = sl == std::string::npos ? std::string("librccl.so.1") : beside.substr(0, sl + 1) + "librccl.so.1"; }
    for (const char* name : {forced ? forced : beside.c_str(), beside.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
        "/opt/rocm/lib/librccl.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) { if (void* f = dlsym(h, "ncclAllReduce")) return (nccl_allreduce_fn)f;
      } }
    return nullptr; }();
  return fn;
}
}  // namespace
extern "C" const char* mkp_internal_hip_runtime_path() { static const std::string p = hip_runtime_path(); return p.c_str(); }
extern "C" {
int mkp_histogram_allreduce(mkp_ctx* c, void* nccl_comm, uint32_t base, uint32_t level, uint32_t prefix, uint64_t* out) {
  if (!c || !nccl_comm || !out || base > 3 || level > 1 || prefix > 0xffffu) return MKP_E_INVALID;
  return guarded(c, [&]() {
    nccl_allreduce_fn ar = rccl_allreduce();
    if (!ar) throw Error(MKP_E_DEVICE, "librccl.so not found (ncclAllReduce)");
    require_sample(c);
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    c->d_hist64.ensure(65536 * 8);
    const uint32_t* src = c->d_hist0.as<uint32_t>() + (size_t)base * 65536;
    if (level == 1) {
      hip_check(hipMemsetAsync(c->d_hist1.p, 0, 65536 * 4, c->stream), "memset");
      hip_check(mkp_launch_sample_hist1(c->stream, c->d_store.as<uint32_t>(), c->sample_n, base, prefix, c->d_hist1.as<uint32_t>()), "hist1 launch");
      src = c->d_hist1.as<uint32_t>();
    }
    hip_check(mkp_launch_widen(c->stream, src, c->d_hist64.as<unsigned long long>(), 65536u), "widen launch");
    const int rc = ar(c->d_hist64.p, c->d_hist64.p, 65536, /*ncclUint64*/ 5, /*ncclSum*/ 0, nccl_comm, c->stream);
    if (rc != 0) throw Error(MKP_E_DEVICE, "ncclAllReduce failed (" + std::to_string(rc) + ")");
    hip_check(hipMemcpyAsync(out, c->d_hist64.p, 65536 * 8, hipMemcpyDeviceToHost, c->stream), "D2H");
    hip_check(hipStreamSynchronize(c->stream), "all-reduce sync");
  });
}

// the same histograms over values the caller holds on the host (tests, and callers that sample elsewhere)
int mkp_histogram_from_values(const float* vals, uint64_t n, uint32_t level, uint32_t prefix, uint64_t* out) {
  if ((!vals && n) || !out || level > 1 || prefix > 0xffffu) return MKP_E_INVALID;
  memset(out, 0, 65536 * sizeof(uint64_t));
  for (uint64_t i = 0; i < n; i++) {
    uint32_t b; memcpy(&b, &vals[i], 4);
    if (b >> 30) return MKP_E_INVALID;   // probabilities are in [0, 2)
    if (level == 0) out[b >> 16]++; else if ((b >> 16) == prefix) out[b & 0xffffu]++;
  }
  return MKP_OK;
}

// percentile_linear_interp (thresholds.rs:17-38) needs xs[floor((n-1)q)] and xs[ceil((n-1)q)] of the sorted sample: which
// level-0 bins hold them, and at which rank inside the bin
int mkp_histogram_locate(const uint64_t* hist0, float q, uint32_t bins[2], uint64_t ranks_in_bin[2], uint64_t* n_out) {
  if (!hist0 || !bins || !ranks_in_bin) return MKP_E_INVALID;
  uint64_t n = 0; for (size_t i = 0; i < 65536; i++) n += hist0[i];
  if (n_out) *n_out = n;
  if (n < 2 || !(q >= 0.0f) || q > 1.0f) return MKP_E_THRESHOLD;
  uint64_t want[2];
  if (q == 1.0f) want[0] = want[1] = n - 1;
  else { const float l = (float)(n - 1), lq = l * q; want[0] = (uint64_t)floorf(lq); want[1] = (uint64_t)ceilf(lq);
    if (want[1] > n - 1) want[1] = n - 1;
      if (want[0] > n - 1) want[0] = n - 1; }
  for (int k = 0; k < 2; k++) {
    uint64_t cum = 0; bool found = false;
    for (uint32_t b = 0; b < 65536 && !found; b++) { if (want[k] < cum + hist0[b]) { bins[k] = b; ranks_in_bin[k] = want[k] - cum; found = true;
      } cum += hist0[b]; }
    if (!found) return MKP_E_THRESHOLD;
  }
  return MKP_OK;
}

int mkp_histogram_resolve(uint32_t prefix, const uint64_t* hist1, uint64_t rank_in_bin, float* value) {
  if (!hist1 || !value || prefix > 0xffffu) return MKP_E_INVALID;
  uint64_t cum = 0;
  for (uint32_t b = 0; b < 65536; b++) { if (rank_in_bin < cum + hist1[b]) { const uint32_t bits = (prefix << 16) | b; memcpy(value, &bits, 4);
      return MKP_OK;
      } cum += hist1[b]; }
  return MKP_E_THRESHOLD;
}

// the interpolation itself, on the two order statistics: y0 * (1 - g) + y1 * g with g = fract((n-1) q), in f32 as the reference does
int mkp_percentile_from_histogram(uint64_t n, float q, float y0, float y1, float* out) {
  if (!out || n < 2 || !(q >= 0.0f) || q > 1.0f) return MKP_E_THRESHOLD;
  if (q == 1.0f) { *out = y1; return MKP_OK; }
  const float l = (float)(n - 1), lq = l * q, g = lq - truncf(lq);
  *out = y0 * (1.0f - g) + y1 * g;
  return MKP_OK;
}

}  // extern "C"
