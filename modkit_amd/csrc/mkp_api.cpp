// libmkpileup C ABI (include/mkpileup.h): context, shard residency in HBM, kernel launches,
// row read-back.  No CPU fallback lives here: the only way rows are produced is the three HIP
// kernels of mkp_kernels.hip; without a gfx950 device every compute entry point fails with
// MKP_E_DEVICE.
#include <memory>
#include <thread>

#include "mkp_ctx.hpp"

using namespace mkp;

extern "C" {
hipError_t mkp_launch_decode(hipStream_t, const MkpReadHdr*, const uint32_t* /*read ids by class*/, const uint32_t* /*n_class[3]*/, const uint32_t*, const uint8_t*, const MkpTagRef*, const uint32_t*,
                             const uint8_t*, const MkpLayout*, const MkpRunParams*, MkpEvent*, MkpReadOut*, uint32_t*, const uint8_t*, float*);
hipError_t mkp_pileup_set_lds(uint32_t accum_bytes, int tally8);
uint32_t mkp_rows_segments(uint32_t tile);
hipError_t mkp_launch_pileup(hipStream_t, uint32_t, const MkpReadHdr*, const uint32_t*, const uint8_t*, const MkpEvent*, const MkpReadOut*,
                             const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, const MkpRunParams* /*device*/, uint32_t* /*tallies*/, const uint32_t* /*chunk offsets*/, uint32_t* /*error bits*/, int /*8-bit tally layout*/);
hipError_t mkp_launch_rows(hipStream_t, const uint32_t* /*tallies*/, const uint32_t*, uint32_t, uint32_t /*tile*/, uint32_t /*arrays*/, int /*has focus*/, const uint8_t*, const MkpCombo*, const MkpRunParams* /*device*/,
                           const MkpRowsDev*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, int /*8-bit tally layout*/);
hipError_t mkp_launch_gather(hipStream_t, const uint32_t*, const uint32_t*, uint32_t*, uint32_t, uint32_t*, const MkpRowsDev*, const MkpRowsDev*);
}

namespace {

MkpRowsDev carve_rows(DevBuf& b, uint64_t cap) {
  b.ensure(cap * 44 + 64);
  MkpRowsDev r; uint32_t* p = b.as<uint32_t>();
  r.pos = p; r.info = p + cap; r.code = p + 2 * cap; r.n_valid = p + 3 * cap; r.n_mod = p + 4 * cap; r.n_can = p + 5 * cap; r.n_other = p + 6 * cap;
  r.n_del = p + 7 * cap; r.n_fail = p + 8 * cap; r.n_diff = p + 9 * cap; r.n_nocall = p + 10 * cap;
  return r;
}

// reads by decode kernel: [FAST layouts with one tag | FAST layouts with two tags | everything else]
void class_ids(const ShardHost& S, const LayoutTables& T, std::vector<uint32_t>* ids, uint32_t n_class[3]) {
  std::vector<uint32_t> cls[3];
  for (size_t i = 0; i < S.hdr.size(); i++) {
    const MkpReadHdr& h = S.hdr[i]; int c = 2;
    if (!(h.flags & MKP_RF_BAD) && h.n_tags && h.layout < T.dev.size() && T.dev[h.layout].fast && h.n_tags <= 2) c = h.n_tags - 1;
    cls[c].push_back((uint32_t)i);
  }
  // one wave decodes one read start to end: launch the longest reads first so they do not form the kernel's tail
  for (int c = 0; c < 3; c++) std::stable_sort(cls[c].begin(), cls[c].end(), [&](uint32_t x, uint32_t y) { return S.hdr[x].l_seq > S.hdr[y].l_seq; });
  ids->clear();
  for (int c = 0; c < 3; c++) { n_class[c] = (uint32_t)cls[c].size(); ids->insert(ids->end(), cls[c].begin(), cls[c].end()); }
}

template <class V> void upload(DevBuf& b, const V& v) {
  const size_t bytes = v.size() * sizeof(v[0]);
  b.ensure(std::max<size_t>(bytes, 16));
  if (!v.empty()) hip_check(hipMemcpy(b.p, v.data(), bytes, hipMemcpyHostToDevice), "H2D");
}

// derive tile geometry, tile read ranges and the run parameters; upload everything
void make_resident(mkp_ctx* c) {
  auto t0 = std::chrono::steady_clock::now();
  ShardHost& S = c->shard;
  // hazard: the reference's ReadCache is keyed by read NAME (read_cache.rs:28-35); two kept records with
  // one name in one interval share a cache entry there.  Not reproduced -> refuse loudly.
  { std::vector<uint64_t> h(S.name_hash.begin(), S.name_hash.end()); std::sort(h.begin(), h.end()); for (size_t i = 1; i < h.size(); i++) if (h[i] == h[i - 1]) throw Error(MKP_E_UNSUPPORTED, "two primary records share a read name in one shard (unmarked duplicates / paired or split reads); the reference keys its per-interval cache by name and this is not reproduced on the device"); }
  c->tables.build(c->packer.layouts, c->caller);
  MkpRunParams& P = c->prm; memset(&P, 0, sizeof(P));
  P.win_start = S.win_start; P.win_end = S.win_end;
  P.n_counters = c->tables.n_counters; P.n_slots = (uint32_t)c->tables.st.slots.size(); P.n_pb = (uint32_t)c->tables.st.can_pbs.size();
  P.numeric_mode = c->caller.numeric_mode; P.combine_strands = c->caller.combine_strands; P.edge_filter = c->caller.edge; P.edge_start = c->caller.edge_start;
  P.edge_end = c->caller.edge_end; P.edge_inverted = c->caller.edge_inverted; P.force_allow = c->caller.force_allow; P.max_depth = c->caller.max_depth;
  P.has_focus = c->has_focus; P.n_combos = (uint32_t)c->combos.size();
  if (const char* dbg = getenv("MKP_DEBUG_SKIP")) P.debug_skip = (uint32_t)strtoul(dbg, nullptr, 0);
  for (int b = 0; b < 4; b++) { P.can_of_pb[b] = 0xff; P.pb_of_can[b] = 0; }
  for (size_t k = 0; k < c->tables.st.can_pbs.size(); k++) { P.can_of_pb[c->tables.st.can_pbs[k]] = (uint8_t)k; P.pb_of_can[k] = (uint8_t)c->tables.st.can_pbs[k]; }
  std::vector<int> order(P.n_slots); for (uint32_t i = 0; i < P.n_slots; i++) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { const MkpSlot &x = c->tables.st.slots[(size_t)a], &y = c->tables.st.slots[(size_t)b]; return x.code_repr != y.code_repr ? x.code_repr < y.code_repr : x.pb < y.pb; });
  for (uint32_t i = 0; i < P.n_slots; i++) { P.slot_order[i] = (uint8_t)order[i]; P.slots[i] = c->tables.st.slots[i]; }
  if (P.combine_strands && !P.has_focus) throw Error(MKP_E_INVALID, "combine_strands needs motif focus positions");
  // tile geometry from the LDS budget (160 KiB per CU on gfx950): the accumulate kernel runs two workgroups per CU, each
  // holding one tile of packed tallies (one dword per counter / slot and position) plus its waves' scratch in 80 KiB;
  // the row kernel unpacks one tile into twice that
  // tile plan for a given number of packed dwords per position: tile length from the LDS budget, then tile -> [first,last) reads
  // (reads are coordinate sorted; the prefix-max of ends bounds the first candidate)
  struct TilePlan { uint32_t T = 0, n_total = 0; size_t max_reads = 0; std::vector<uint32_t> ids, first, last; };
  auto plan_tiles = [&](uint32_t words_per_pos) {
    TilePlan tp;
    uint32_t T = c->cfg.tile_positions;
    if (const char* te = getenv("MKP_TILE")) T = (uint32_t)strtoul(te, nullptr, 0);   // experiments only
    // two workgroups must be co-resident per CU: 76 KiB each including the kernel's ~1 KiB of static LDS leaves 8 KiB of
    // slack for the allocation granule (at 80 KiB each one GPU box ran them one per CU and the kernel took 1.9x as long)
    const uint32_t budget = 76u * 1024u - 1280u;
    uint32_t maxT = 0;
    for (uint32_t t = 64; t <= 8192; t += 64) if (MKP_PILEUP_LDS_WORDS(words_per_pos, t + 2 * MKP_HALO) * 4u <= budget) maxT = t;
    if (maxT < 64) throw Error(MKP_E_UNSUPPORTED, "too many counters for one LDS tile");
    if (!T || T > maxT) T = std::min<uint32_t>(maxT, 4096u);
    T = std::max<uint32_t>(64u, T & ~63u);
    tp.T = T;
    const uint64_t win = (uint64_t)(S.win_end - S.win_start);
    tp.n_total = (uint32_t)((win + T - 1) / T);
    const size_t n = S.hdr.size(); std::vector<int32_t> pmax(n); int32_t m = INT32_MIN;
    for (size_t i = 0; i < n; i++) { if (i && S.hdr[i].ref_start < S.hdr[i - 1].ref_start) throw Error(MKP_E_INVALID, "records must be coordinate sorted"); m = std::max(m, S.hdr[i].ref_end); pmax[i] = m; }
    size_t first = 0, last = 0;
    for (uint32_t t = 0; t < tp.n_total; t++) {
      const int64_t lo = (int64_t)S.win_start + (int64_t)t * T - MKP_HALO, hi = (int64_t)S.win_start + (int64_t)(t + 1) * T + MKP_HALO;
      while (first < n && pmax[first] <= lo) first++;
      if (last < first) last = first;
      while (last < n && S.hdr[last].ref_start < hi) last++;
      bool any = false; for (size_t i = first; i < last && !any; i++) any = S.hdr[i].ref_end > lo;
      if (any) { tp.ids.push_back(t); tp.first.push_back((uint32_t)first); tp.last.push_back((uint32_t)last); }
      tp.max_reads = std::max(tp.max_reads, last - first);   // every column's reads are among the tile's
    }
    return tp;
  };
  // tile geometry from the LDS budget (160 KiB per CU on gfx950): the accumulate kernel runs two workgroups per CU, each
  // holding one tile of packed tallies plus its waves' scratch.  Default layout: one dword per counter / slot and position
  // (16 bits per strand).  MKP_TALLY8=1 (opt-in until validated on the GPU): four 8-bit fields per dword when no tile sees
  // more than 255 reads — fewer dwords per position, longer tiles, fewer (read, tile) visits.
  uint32_t words_per_pos = P.n_counters + P.n_slots;
  c->tally8 = 0;
  TilePlan tp;
  if (const char* t8 = getenv("MKP_TALLY8")) if (atoi(t8) == 1 && P.n_counters >= 5) {
    const uint32_t words8 = 2u + ((2u + 2u * P.n_slots + 3u) >> 2) + ((2u * (P.n_counters - 5u) + 3u) >> 2);
    tp = plan_tiles(words8);
    if (tp.max_reads <= 255) { c->tally8 = 1; words_per_pos = words8; }
  }
  if (!c->tally8) tp = plan_tiles(words_per_pos);
  // the packed tallies hold 16 bits per strand: no column may be deeper than 65535
  if (tp.max_reads > 65535) throw Error(MKP_E_UNSUPPORTED, "more than 65535 reads over one tile: columns this deep are outside the device path");
  const uint32_t T = tp.T;
  P.tile = T; c->lds_bytes = MKP_PILEUP_LDS_WORDS(words_per_pos, T + 2 * MKP_HALO) * 4u;
  c->words_per_pos = words_per_pos;
  const uint64_t win = (uint64_t)(S.win_end - S.win_start);
  P.n_tiles_total = tp.n_total;
  std::vector<uint32_t>& tile_ids = tp.ids; std::vector<uint32_t>& tf = tp.first; std::vector<uint32_t>& tl = tp.last;
  c->n_tiles = (uint32_t)tile_ids.size();
  c->stats.pack_ms += ms_since(t0);
  auto t1 = std::chrono::steady_clock::now();
  hip_check(hipSetDevice(c->device), "hipSetDevice");
  upload(c->d_hdr, S.hdr); upload(c->d_cigar, S.cigar); upload(c->d_chunk, S.chunk_pfx); upload(c->d_seq, S.seq); upload(c->d_tagref, S.tagref); upload(c->d_ranks, S.ranks); upload(c->d_ml, S.ml);
  upload(c->d_layouts, c->tables.dev); upload(c->d_tile_ids, tile_ids); upload(c->d_tile_first, tf); upload(c->d_tile_last, tl);
  { std::vector<uint32_t> ids; class_ids(S, c->tables, &ids, c->n_class); upload(c->d_read_ids, ids); }
  if (c->has_focus) { upload(c->d_focus, c->focus); upload(c->d_combos, c->combos); } else { c->d_focus.ensure(16); c->d_combos.ensure(64); }
  c->d_events.ensure(std::max<uint64_t>(S.n_events_cap, 1) * sizeof(MkpEvent));
  c->d_readout.ensure(std::max<size_t>(S.hdr.size(), 1) * sizeof(MkpReadOut));
  c->n_segs = c->n_tiles * mkp_rows_segments(P.tile);   // row segments: 256 positions each, in genome order
  c->d_tile_row_off.ensure((size_t)(c->n_segs + 1) * 4); c->d_tile_row_cnt.ensure((size_t)(c->n_segs + 1) * 4); c->d_tile_dst.ensure((size_t)(c->n_segs + 1) * 4);
  c->d_misc.ensure(64);
  c->d_tally.ensure(std::max<size_t>((size_t)c->n_tiles * words_per_pos * (T + 2 * MKP_HALO) * 4u, 16));
  hip_check(mkp_pileup_set_lds(c->lds_bytes, c->tally8), "hipFuncSetAttribute(max dynamic LDS)");
  hip_check(hipDeviceSynchronize(), "upload sync");
  c->stats.h2d_ms = ms_since(t1);
  c->resident = true;
  // algorithmic bytes
  uint64_t b_reads = 0; for (auto& h : S.hdr) b_reads += 16 + 4ull * h.n_cigar + (h.l_seq + 1) / 2;
  c->stats.n_reads = S.hdr.size(); c->stats.n_tiles = c->n_tiles; c->stats.n_positions = win;
  c->stats.alg_bytes_decode = b_reads + S.ranks.size() * 2ull + S.ml.size();  // + 8*events added after the run
  c->stats.alg_bytes_pileup = b_reads;                                          // + 8*events added after the run; rows: 44 B each
}

void run_kernels(mkp_ctx* c, bool time_kernels) {
  MkpRunParams& P = c->prm;
  if (c->row_cap == 0) {
    uint64_t guess = c->has_focus ? 1u << 20 : (uint64_t)c->stats.n_positions * 2 + 1024;
    c->row_cap = std::max<uint64_t>(1u << 16, std::min<uint64_t>(guess, 1ull << 28));
  }
  for (;;) {
    P.row_capacity = (uint32_t)c->row_cap;
    c->rows_src = carve_rows(c->d_rows_src, c->row_cap); c->rows_dst = carve_rows(c->d_rows_dst, c->row_cap);
    uint32_t* misc = c->d_misc.as<uint32_t>();  // [0] row cursor, [1] total rows, [2] error bits
    hip_check(hipMemsetAsync(misc, 0, 16, c->stream), "memset");
    c->d_prm.ensure(sizeof(MkpRunParams));
    hip_check(hipMemcpyAsync(c->d_prm.p, &P, sizeof(MkpRunParams), hipMemcpyHostToDevice, c->stream), "params H2D");
    if (time_kernels) hip_check(hipEventRecord(c->ev[0], c->stream), "event");
    hip_check(mkp_launch_decode(c->stream, c->d_hdr.as<MkpReadHdr>(), c->d_read_ids.as<uint32_t>(), c->n_class, c->d_cigar.as<uint32_t>(), c->d_seq.as<uint8_t>(), c->d_tagref.as<MkpTagRef>(),
                                c->d_ranks.as<uint32_t>(), c->d_ml.as<uint8_t>(), c->d_layouts.as<MkpLayout>(), &P, c->d_events.as<MkpEvent>(), c->d_readout.as<MkpReadOut>(), misc + 2, c->d_focus.as<uint8_t>(), nullptr), "decode launch");
    if (time_kernels) hip_check(hipEventRecord(c->ev[1], c->stream), "event");
    hip_check(mkp_launch_pileup(c->stream, c->lds_bytes, c->d_hdr.as<MkpReadHdr>(), c->d_cigar.as<uint32_t>(), c->d_seq.as<uint8_t>(), c->d_events.as<MkpEvent>(), c->d_readout.as<MkpReadOut>(),
                                c->d_tile_ids.as<uint32_t>(), c->d_tile_first.as<uint32_t>(), c->d_tile_last.as<uint32_t>(), c->n_tiles, c->d_prm.as<MkpRunParams>(), c->d_tally.as<uint32_t>(), c->d_chunk.as<uint32_t>(), misc + 2, c->tally8), "pileup launch");
    if (time_kernels) hip_check(hipEventRecord(c->ev[2], c->stream), "event");
    hip_check(mkp_launch_rows(c->stream, c->d_tally.as<uint32_t>(), c->d_tile_ids.as<uint32_t>(), c->n_tiles, P.tile, c->words_per_pos, (int)P.has_focus, c->d_focus.as<uint8_t>(), c->d_combos.as<MkpCombo>(), c->d_prm.as<MkpRunParams>(),
                              &c->rows_src, misc, c->d_tile_row_off.as<uint32_t>(), c->d_tile_row_cnt.as<uint32_t>(), misc + 2, c->tally8), "rows launch");
    if (time_kernels) hip_check(hipEventRecord(c->ev[3], c->stream), "event");
    hip_check(mkp_launch_gather(c->stream, c->d_tile_row_off.as<uint32_t>(), c->d_tile_row_cnt.as<uint32_t>(), c->d_tile_dst.as<uint32_t>(), c->n_segs, misc + 1, &c->rows_src, &c->rows_dst), "gather launch");
    if (time_kernels) hip_check(hipEventRecord(c->ev[4], c->stream), "event");
    uint32_t h[4];
    hip_check(hipMemcpyAsync(h, misc, 16, hipMemcpyDeviceToHost, c->stream), "D2H");
    hip_check(hipStreamSynchronize(c->stream), "kernel sync");
    if (h[2] & 2u) { c->row_cap *= 2; if (c->row_cap > (1ull << 31)) throw Error(MKP_E_NOMEM, "row buffer would exceed 2^31 rows"); continue; }
    if (h[2] & 1u) throw Error(MKP_E_DEVICE, "internal: event segment overflow");
    if (h[2] & 4u) throw Error(MKP_E_UNSUPPORTED, "a pileup column is deeper than max_depth; htslib's maxcnt read-dropping is not reproduced");
    c->stats.n_rows = h[1];
    if (time_kernels) {
      float a = 0, b = 0, r = 0, d = 0; hip_check(hipEventElapsedTime(&a, c->ev[0], c->ev[1]), "event"); hip_check(hipEventElapsedTime(&b, c->ev[1], c->ev[2]), "event");
      hip_check(hipEventElapsedTime(&r, c->ev[2], c->ev[3]), "event"); hip_check(hipEventElapsedTime(&d, c->ev[3], c->ev[4]), "event");
      c->stats.decode_kernel_ms = a; c->stats.pileup_kernel_ms = b; c->stats.rows_kernel_ms = r; c->stats.gather_kernel_ms = d; c->stats.kernel_ms = a + b + r + d;
    }
    return;
  }
}

void fetch_rows(mkp_ctx* c, mkp_rows* out) {
  auto t0 = std::chrono::steady_clock::now();
  const uint64_t n = c->stats.n_rows, cap = c->row_cap;
  const uint32_t* src[11] = {c->rows_dst.pos, c->rows_dst.info, c->rows_dst.code, c->rows_dst.n_valid, c->rows_dst.n_mod, c->rows_dst.n_can, c->rows_dst.n_other,
                             c->rows_dst.n_del, c->rows_dst.n_fail, c->rows_dst.n_diff, c->rows_dst.n_nocall};
  (void)cap;
  for (int k = 0; k < 11; k++) { c->h_rows[k].resize(n); if (n) hip_check(hipMemcpy(c->h_rows[k].data(), src[k], n * 4, hipMemcpyDeviceToHost), "rows D2H"); }
  c->h_strand.resize(n); c->h_motif.resize(n);
  for (uint64_t i = 0; i < n; i++) { uint32_t inf = c->h_rows[1][i]; c->h_strand[i] = "+-."[inf & 3u]; c->h_motif[i] = (int32_t)(inf >> 8) - 1; }
  // per-read outcome counts
  std::vector<MkpReadOut> ro(c->shard.hdr.size());
  if (!ro.empty()) hip_check(hipMemcpy(ro.data(), c->d_readout.p, ro.size() * sizeof(MkpReadOut), hipMemcpyDeviceToHost), "readout D2H");
  c->n_ok = 0; c->n_bad = 0; uint64_t ev = 0;
  for (auto& r : ro) { if (r.ok) { c->n_ok++; ev += r.n_events; } else c->n_bad++; }
  c->stats.n_events = ev;
  c->stats.d2h_ms = ms_since(t0);
  if (out) {
    out->n_rows = n; out->pos = c->h_rows[0].data(); out->strand = c->h_strand.data(); out->code_repr = c->h_rows[2].data(); out->motif_idx = c->h_motif.data();
    out->n_valid = c->h_rows[3].data(); out->n_mod = c->h_rows[4].data(); out->n_canonical = c->h_rows[5].data(); out->n_other = c->h_rows[6].data();
    out->n_delete = c->h_rows[7].data(); out->n_fail = c->h_rows[8].data(); out->n_diff = c->h_rows[9].data(); out->n_nocall = c->h_rows[10].data();
    out->processed_records = c->n_ok; out->skipped_records = c->n_bad;
  }
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
  if (hipSetDevice(c->device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) { delete c; return MKP_E_DEVICE; }
  for (auto& e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return MKP_E_DEVICE; }
  c->caller = CallerCfg();
  *out = c;
  return MKP_OK;
}

void mkp_ctx_destroy(mkp_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (DevBuf* b : {&c->d_vals, &c->d_hdr, &c->d_cigar, &c->d_seq, &c->d_tagref, &c->d_ranks, &c->d_ml, &c->d_layouts, &c->d_events, &c->d_readout, &c->d_focus, &c->d_combos, &c->d_tile_ids,
                    &c->d_tile_first, &c->d_tile_last, &c->d_prm, &c->d_read_ids, &c->d_tally, &c->d_chunk, &c->d_tile_row_off, &c->d_tile_row_cnt, &c->d_tile_dst, &c->d_misc, &c->d_rows_src, &c->d_rows_dst}) b->release();
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
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
    cc.numeric_mode = k->numeric_mode; cc.collapse_code = k->collapse_code; cc.edge = k->edge_filter != 0; cc.edge_start = k->edge_start; cc.edge_end = k->edge_end;
    cc.edge_inverted = k->edge_inverted != 0; cc.force_allow = k->force_allow_implicit != 0; cc.combine_strands = k->combine_strands != 0;
    cc.max_depth = k->max_depth ? k->max_depth : 8000;
    c->caller = cc; c->caller_set = true; c->resident = false;
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
      c->focus.assign(s->focus, s->focus + (s->end - s->start));
      if (s->n_combos > 64) throw Error(MKP_E_UNSUPPORTED, "more than 64 motif combos");
      c->combos.assign(s->combos, s->combos + s->n_combos);
      if (c->combos.empty()) { mkp_motif_combo z; memset(&z, 0, sizeof(z)); c->combos.push_back(z); }
    } else { c->focus.clear(); c->combos.clear(); }
    c->shard_open = true; c->resident = false; c->row_cap = 0;
    memset(&c->stats, 0, sizeof(c->stats));
  });
}

int mkp_shard_add_records(mkp_ctx* c, const mkp_record* recs, uint32_t n) {
  if (!c || (!recs && n)) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!c->shard_open) throw Error(MKP_E_INVALID, "mkp_shard_begin first");
    auto t0 = std::chrono::steady_clock::now();
    const int32_t tid = c->shard.tid;
    pack_records(c->packer, c->shard, recs, n, [tid](const mkp_record& r) { return r.tid == tid && Packer::keep(r); });
    c->stats.pack_ms += ms_since(t0);
  });
}

int mkp_shard_run(mkp_ctx* c, mkp_rows* out) {
  if (!c) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!c->shard_open) throw Error(MKP_E_INVALID, "mkp_shard_begin first");
    if (!c->caller_set) throw Error(MKP_E_INVALID, "mkp_set_caller first");
    if (!c->resident) make_resident(c);
    run_kernels(c, true);
    fetch_rows(c, out);
    c->stats.alg_bytes_decode += 8ull * c->stats.n_events;
    c->stats.alg_bytes_pileup += 8ull * c->stats.n_events;
    c->stats.alg_bytes_rows = 44ull * c->stats.n_rows;
  });
}

int mkp_shard_rerun(mkp_ctx* c, uint32_t iters, mkp_rows* out) {
  if (!c) return MKP_E_INVALID;
  return guarded(c, [&]() {
    if (!c->resident) throw Error(MKP_E_INVALID, "no resident shard: call mkp_shard_run once first");
    double d = 0, p = 0, r = 0, g = 0;
    for (uint32_t i = 0; i < iters; i++) { run_kernels(c, true); d += c->stats.decode_kernel_ms; p += c->stats.pileup_kernel_ms; r += c->stats.rows_kernel_ms; g += c->stats.gather_kernel_ms; }
    if (iters) { c->stats.decode_kernel_ms = d / iters; c->stats.pileup_kernel_ms = p / iters; c->stats.rows_kernel_ms = r / iters; c->stats.gather_kernel_ms = g / iters; c->stats.kernel_ms = (d + p + r + g) / iters; }
    if (out) fetch_rows(c, out);
  });
}

int mkp_get_stats(const mkp_ctx* c, mkp_stats* out) { if (!c || !out) return MKP_E_INVALID; *out = c->stats; return MKP_OK; }

int mkp_percentile(const float* xs, uint64_t n, float q, float* out) {  // percentile_linear_interp (thresholds.rs:17-38)
  if (!xs || !out || n < 2 || q > 1.0f) return MKP_E_THRESHOLD;
  if (q == 1.0f) { *out = xs[n - 1]; return MKP_OK; }
  float l = (float)(n - 1), lq = l * q, left = floorf(lq); uint64_t right = (uint64_t)ceilf(lq);
  float g = lq - truncf(lq), a = xs[(uint64_t)left] * (1.0f - g), b = xs[right] * g;
  *out = a + b;
  return MKP_OK;
}

int mkp_host_mm_ranks(const char* mm, uint32_t l_seq, uint32_t n_ml, mkp_host_tag* tags, uint32_t tags_cap, uint32_t* ranks, uint32_t ranks_cap) {
  if (!mm || (!tags && tags_cap) || (!ranks && ranks_cap)) return MKP_E_INVALID;
  try {
    // a synthetic one-record shard: qname "r", one M op, all-A SEQ, aux = MM:Z + ML:B:C (n_ml zero bytes)
    std::vector<uint8_t> data; data.push_back('r'); data.push_back(0);
    const uint32_t cg = (l_seq << 4) | 0u; data.insert(data.end(), (const uint8_t*)&cg, (const uint8_t*)&cg + 4);
    data.insert(data.end(), (l_seq + 1) / 2, 0x11); data.insert(data.end(), l_seq, 0xff);
    data.push_back('M'); data.push_back('M'); data.push_back('Z'); data.insert(data.end(), mm, mm + strlen(mm) + 1);
    data.push_back('M'); data.push_back('L'); data.push_back('B'); data.push_back('C'); data.insert(data.end(), (const uint8_t*)&n_ml, (const uint8_t*)&n_ml + 4); data.insert(data.end(), n_ml, 0);
    mkp_record r; memset(&r, 0, sizeof(r)); r.tid = 0; r.pos = 0; r.l_qname = 2; r.n_cigar = 1; r.l_qseq = (int32_t)l_seq; r.l_data = (int32_t)data.size(); r.data = data.data();
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
  try { FxOrder m; for (uint32_t i = 0; i < n; i++) m.insert(code_reprs[i], (int)i); auto c = m.codes(); for (size_t i = 0; i < c.size(); i++) order_out[i] = c[i]; return (int)c.size(); }
  catch (...) { return MKP_E_INVALID; }
}

}  // extern "C"

int mkp_internal_sample(mkp_ctx* c, int32_t tid, uint32_t win_start, uint32_t win_end, const uint8_t* bedmask, const mkp_record* recs,
                        uint32_t n, bool only_mapped, mkp::SampleOut* out) {
  if (!c || !out) return MKP_E_INVALID;
  return guarded(c, [&]() {
    ShardHost S; S.tid = tid; S.win_start = (int32_t)win_start; S.win_end = (int32_t)win_end;
    pack_records(c->packer, S, recs, n, [](const mkp_record&) { return true; });
    c->tables.build(c->packer.layouts, c->caller);
    MkpRunParams P; memset(&P, 0, sizeof(P));
    P.win_start = S.win_start; P.win_end = S.win_end; P.numeric_mode = c->caller.numeric_mode; P.edge_filter = c->caller.edge; P.edge_start = c->caller.edge_start;
    P.edge_end = c->caller.edge_end; P.edge_inverted = c->caller.edge_inverted; P.force_allow = 1; P.sample_mode = 1; P.only_mapped = only_mapped; P.has_focus = bedmask != nullptr;
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    upload(c->d_hdr, S.hdr); upload(c->d_cigar, S.cigar); upload(c->d_seq, S.seq); upload(c->d_tagref, S.tagref); upload(c->d_ranks, S.ranks); upload(c->d_ml, S.ml);
    upload(c->d_layouts, c->tables.dev);
    { std::vector<uint32_t> ids; class_ids(S, c->tables, &ids, c->n_class); upload(c->d_read_ids, ids); }
    if (bedmask) { c->d_focus.ensure((size_t)(win_end - win_start)); hip_check(hipMemcpy(c->d_focus.p, bedmask, (size_t)(win_end - win_start), hipMemcpyHostToDevice), "H2D"); } else c->d_focus.ensure(16);
    const uint64_t cap = std::max<uint64_t>(S.n_events_cap, 1);
    c->d_events.ensure(cap * sizeof(MkpEvent)); c->d_vals.ensure(cap * sizeof(float)); c->d_readout.ensure(std::max<size_t>(S.hdr.size(), 1) * sizeof(MkpReadOut)); c->d_misc.ensure(64);
    uint32_t* misc = c->d_misc.as<uint32_t>();
    hip_check(hipMemsetAsync(misc, 0, 16, c->stream), "memset");
    hip_check(mkp_launch_decode(c->stream, c->d_hdr.as<MkpReadHdr>(), c->d_read_ids.as<uint32_t>(), c->n_class, c->d_cigar.as<uint32_t>(), c->d_seq.as<uint8_t>(), c->d_tagref.as<MkpTagRef>(), c->d_ranks.as<uint32_t>(),
                                c->d_ml.as<uint8_t>(), c->d_layouts.as<MkpLayout>(), &P, c->d_events.as<MkpEvent>(), c->d_readout.as<MkpReadOut>(), misc + 2, c->d_focus.as<uint8_t>(), c->d_vals.as<float>()), "decode(sample) launch");
    hip_check(hipStreamSynchronize(c->stream), "sample sync");
    uint32_t h[4]; hip_check(hipMemcpy(h, misc, 16, hipMemcpyDeviceToHost), "D2H");
    if (h[2] & 1u) throw Error(MKP_E_DEVICE, "internal: event segment overflow");
    std::vector<MkpReadOut> ro(S.hdr.size()); std::vector<MkpEvent> ev(S.n_events_cap); std::vector<float> vals(S.n_events_cap);
    if (!ro.empty()) hip_check(hipMemcpy(ro.data(), c->d_readout.p, ro.size() * sizeof(MkpReadOut), hipMemcpyDeviceToHost), "D2H");
    if (!ev.empty()) { hip_check(hipMemcpy(ev.data(), c->d_events.p, ev.size() * sizeof(MkpEvent), hipMemcpyDeviceToHost), "D2H"); hip_check(hipMemcpy(vals.data(), c->d_vals.p, vals.size() * 4, hipMemcpyDeviceToHost), "D2H"); }
    out->ok.clear(); out->n.clear(); out->off.clear(); out->vals.clear(); out->base.clear();
    size_t total = 0; for (size_t i = 0; i < S.hdr.size(); i++) if (ro[i].ok) total += ro[i].n_events;
    out->vals.reserve(total); out->base.reserve(total);
    for (size_t i = 0; i < S.hdr.size(); i++) {
      out->ok.push_back(ro[i].ok); out->n.push_back(ro[i].ok ? ro[i].n_events : 0); out->off.push_back((uint32_t)out->vals.size());
      if (ro[i].ok && ro[i].n_events) {
        const size_t e0 = S.hdr[i].event_off;
        out->vals.insert(out->vals.end(), vals.begin() + (std::ptrdiff_t)e0, vals.begin() + (std::ptrdiff_t)(e0 + ro[i].n_events));
        for (uint32_t k = 0; k < ro[i].n_events; k++) out->base.push_back((uint8_t)ev[e0 + k].info);
      }
    }
    c->resident = false;
  });
}
