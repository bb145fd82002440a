// Device ingest, host side: one shard window of an indexed BAM goes to the GPU COMPRESSED and comes back as a digest.
// Stands in for htslib's IndexedReader::fetch + records() (src/pileup/mod.rs:732-759) and for the tag getters / MmTagInfo::parse
// (src/mod_bam.rs:1388-1470, 900-1000) on the shard path: the BGZF blocks the index lists are uploaded as they sit in the file,
// inflated (mkp_inflate_wave4.hip), CRC-checked, cut into records, filtered and packed into the shard arrays (mkp_ingest.hip) — the
// inflated bytes never exist on the host.  What returns: one MkpReadHdr + tag table + two hashes per kept record (the planner's input),
// the spans of supplementary records (max-depth guard), block status words and error bits.  MM header structures ("layouts") are
// interned on the host from the key hash of each record; the text of a structure seen for the first time is fetched from HBM and
// parsed by the host packer itself.
#include <atomic>
#include <memory>
#include <thread>

#include "mkp_ingest_dev.hpp"
#include "mkp_ingest_host.hpp"
#include <cmath>

using namespace mkp;

extern "C" {
hipError_t mkp_launch_crc32(hipStream_t, const uint8_t*, const void*, uint32_t, const uint8_t*, uint32_t*);
hipError_t mkp_launch_bgzf_chain_count(hipStream_t, const uint8_t*, const MkpZChain*, uint32_t, uint32_t*, uint32_t*);
hipError_t mkp_launch_bgzf_chain_write(hipStream_t, const uint8_t*, const MkpZChain*, uint32_t, const uint32_t*, uint32_t, MkpZBlk*, uint32_t*);
hipError_t mkp_launch_bgzf_layout(hipStream_t, const MkpZBlk*, uint32_t, unsigned long long*, unsigned long long, void*, uint32_t*);
hipError_t mkp_launch_ingest_count(hipStream_t, const uint8_t*, const MkpIngestParams*, const MkpSeg*, uint32_t*, MkpIngestTotals*);
hipError_t mkp_launch_ingest_parse(hipStream_t, const uint8_t*, const MkpIngestParams*, const int32_t*, const MkpSeg*, const uint32_t*,
    unsigned long long*, MkpRecInfo*, uint32_t*, int32_t*, MkpIngestTotals*);
hipError_t mkp_launch_ingest_pack(hipStream_t, const uint8_t*, uint32_t, const MkpRecInfo*, const uint32_t*, MkpReadHdr*, uint32_t*, uint32_t*,
    uint8_t*, MkpTagRef*, uint32_t*, uint8_t*, MkpRecDigest*, MkpIngestTotals*);
}

namespace {
struct Pinned {
  void* p = nullptr; size_t cap = 0;
  void ensure(size_t n) { if (n <= cap) return; release(); if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) { p = nullptr;
      throw Error(MKP_E_NOMEM, "hipHostMalloc failed"); } cap = n; }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};
struct BgzfBlk { unsigned long long in_off, out_off; uint32_t in_len, out_len; };   // == MkpBgzfBlock
uint64_t fnv64(const std::string& s) { uint64_t h = 1469598103934665603ull; for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; } return h; }
}  // namespace

struct mkp_dev_ingest {
  int device = 0, prio_mid = 0, prio_lo = 0; hipStream_t stream = nullptr, up_stream = nullptr, crc_stream = nullptr;
    hipEvent_t slot_ev[2] = {nullptr, nullptr}, up_done = nullptr, kev[2] = {nullptr, nullptr}, inf_done = nullptr, crc_done = nullptr;
  std::vector<hipEvent_t> stage_ev;      // upload stages: recorded on up_stream behind a stage's last copy
  std::vector<hipEvent_t> tev;           // timed pairs around the inflate launches of the stages
  // the stages' inflate launches: off the ingest stream, whose chain walks (and the host's one sync per stage) then never wait for an inflate.
  hipStream_t inf_stream[1] = {nullptr};
                                           // (Two such streams, launches alternating, ran two stages side by side at half speed each — the same 76 ms
                                           // for the C3 file — and with the
                                           //  process's seventh stream the chain walks began to queue behind inflate launches: 12 ms syncs.)
  std::vector<hipEvent_t> lay_ev;        // a stage's tables are laid out (ingest stream) -> its inflate may start
  // most rounds a stage may take; 0 = whatever has been issued (tests cap it so that a few-MB BAM goes through several stages)
  static constexpr size_t kStageRounds = 0;
  DevBuf rawcur;
  // the staged path's block table and chain counts: written by the chain kernels straight into page-locked host memory (1.3 MB for a chr20 window) —
  Pinned ztab, chain_cnt;
                            // as copies at the end they waited 8-47 ms on a device busy inflating (round 6 trace); the layout kernel reads the table
                            // back over the link
  // upload staging: two halves of kSlots pieces, 32 MiB page-locked in all (allocating it is part of a fresh context's first ingest: 0.22 ms per MiB)
  static constexpr size_t kPiece = (size_t)1 << 20, kSlots = 16;
  Pinned stage, small;                                             // compressed bytes on their way up; tables up / totals + status down
  // the staged path's chain table: the chain kernels read it where it lies (a copy would queue behind the upload's 32 MiB pieces)
  Pinned chain_host;
  DevBuf zin, zblk, zstat, raw, segs, seg_cnt, rec_off, info, sz, extra, tot, dig, parts;
  std::mutex mu;                                                   // one ingest at a time per object
  std::mutex spare_mu; std::vector<DevBuf> spares;                 // buffers the contexts handed back (mkp_internal_ingest_recycle)
  DevBuf take(size_t bytes) {
    DevBuf b;
    { std::lock_guard<std::mutex> g(spare_mu); size_t best = SIZE_MAX;
      for (size_t i = 0; i < spares.size(); i++) if (spares[i].cap >= bytes && (best == SIZE_MAX || spares[i].cap < spares[best].cap)) best = i;
      if (best != SIZE_MAX) { b = spares[best]; spares.erase(spares.begin() + (ptrdiff_t)best); return b; } }
    b.ensure(std::max<size_t>(bytes, 256)); return b;
  }
};

// Test hooks (tests/test_gpu_ingest.py; not part of the ABI): smaller upload pieces / stages and a forced capacity of the inflated window, so
// that a few-MB BAM goes through several stages and outgrows its window in the middle of them.  0 = the product's value.
namespace { struct IngestTune { size_t piece = mkp_dev_ingest::kPiece, stage_rounds = mkp_dev_ingest::kStageRounds; uint64_t raw_cap = 0; } g_tune;
            std::atomic<uint64_t> g_reinflated{0}, g_staged_windows{0}; }
extern "C" uint64_t mkp_internal_ingest_reinflated() { return g_reinflated.load(); }   // windows that outgrew their buffer and were inflated again
extern "C" uint64_t mkp_internal_ingest_staged() { return g_staged_windows.load(); }   // windows whose staged inflate stood
extern "C" void mkp_internal_ingest_tune(uint64_t piece_bytes, uint64_t stage_rounds, uint64_t raw_cap_bytes) {
  g_tune.piece = piece_bytes ? std::min<size_t>((size_t)((piece_bytes + 63) & ~63ull), mkp_dev_ingest::kPiece) : mkp_dev_ingest::kPiece;
  g_tune.stage_rounds = stage_rounds ? (size_t)stage_rounds : mkp_dev_ingest::kStageRounds;
  g_tune.raw_cap = raw_cap_bytes;
}

mkp_dev_ingest* mkp_internal_ingest_create(int device) {
  std::unique_ptr<mkp_dev_ingest> d(new mkp_dev_ingest()); d->device = device;
  // the record kernels' stream goes before the upload and CRC streams: the CRC's thirteen thousand
  // workgroups otherwise hold up the one-workgroup scans that the host waits for (2.6 ms for a 15 us kernel)
  int prio_lo = 0, prio_hi = 0;
  if (hipSetDevice(device) != hipSuccess || hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) return nullptr;
  // (below the contexts' own streams, above the uploads and the CRC)
  const int prio_mid = (prio_lo + prio_hi) / 2 != prio_hi ? (prio_lo + prio_hi) / 2 : prio_hi;
  d->prio_mid = prio_mid; d->prio_lo = prio_lo;   // (streams come from the process-wide pool: 8-10 ms each to create, mkp_ctx.hpp)
  if (pooled_stream_create(&d->stream, device, hipStreamNonBlocking, prio_mid) != hipSuccess
      || pooled_stream_create(&d->up_stream, device, hipStreamNonBlocking, prio_lo) != hipSuccess ||
      pooled_stream_create(&d->crc_stream, device, hipStreamNonBlocking, prio_lo) != hipSuccess) return nullptr;
  for (auto& st : d->inf_stream) if (pooled_stream_create(&st, device, hipStreamNonBlocking, prio_lo) != hipSuccess) return nullptr;
  for (auto& e : d->slot_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&d->up_done, hipEventDisableTiming) != hipSuccess) return nullptr;
  for (auto& e : d->kev) if (hipEventCreate(&e) != hipSuccess) return nullptr;   // (timed: the inflate + chain kernels, for the trace)
  if (hipEventCreateWithFlags(&d->inf_done, hipEventDisableTiming) != hipSuccess
      || hipEventCreateWithFlags(&d->crc_done, hipEventDisableTiming) != hipSuccess) return nullptr;
  return d.release();
}

void mkp_internal_ingest_destroy(mkp_dev_ingest* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  for (DevBuf* b : {&d->zin, &d->zblk, &d->zstat, &d->raw, &d->segs, &d->seg_cnt, &d->rec_off, &d->info, &d->sz, &d->extra, &d->tot, &d->dig,
      &d->parts, &d->rawcur}) b->release();
  d->ztab.release(); d->chain_cnt.release();
  for (auto& e : d->stage_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : d->tev) if (e) (void)hipEventDestroy(e);
  for (auto& e : d->lay_ev) if (e) (void)hipEventDestroy(e);
  for (auto& st : d->inf_stream) pooled_stream_release(st, d->device, hipStreamNonBlocking, d->prio_lo);
  pooled_stream_release(d->crc_stream, d->device, hipStreamNonBlocking, d->prio_lo);
  for (auto& b : d->spares) b.release();
  d->stage.release(); d->small.release(); d->chain_host.release();
  for (auto& e : d->slot_ev) if (e) (void)hipEventDestroy(e);
  if (d->up_done) (void)hipEventDestroy(d->up_done);
  for (auto& e : d->kev) if (e) (void)hipEventDestroy(e);
  if (d->inf_done) (void)hipEventDestroy(d->inf_done);
  if (d->crc_done) (void)hipEventDestroy(d->crc_done);
  pooled_stream_release(d->stream, d->device, hipStreamNonBlocking, d->prio_mid);
  pooled_stream_release(d->up_stream, d->device, hipStreamNonBlocking, d->prio_lo);
  delete d;
}

DevShard::~DevShard() { for (DevBuf* b : {&d_cigar, &d_chunk, &d_seq, &d_tagref, &d_ranks, &d_ml}) b->release(); }

void mkp_internal_ingest_recycle(mkp_dev_ingest* d, DevShard* sh) {
  if (!d || !sh) return;
  std::lock_guard<std::mutex> g(d->spare_mu);
  for (DevBuf* b : {&sh->d_cigar, &sh->d_chunk, &sh->d_seq, &sh->d_tagref, &sh->d_ranks, &sh->d_ml}) { if (b->p && d->spares.size() < 16) {
      d->spares.push_back(*b); b->p = nullptr; b->cap = 0; } }
}

std::unique_ptr<DevShard> mkp_internal_ingest_run(mkp_dev_ingest* d, const BamSource& bam, uint32_t tid, const FetchParts& parts) {
  if (parts.empty()) throw Error(MKP_E_INVALID, "device ingest: no fetch window");
  const uint32_t beg = (uint32_t)std::max<int64_t>(parts.front().first, 0), end = (uint32_t)std::min<int64_t>(parts.back().second, 0x7fffffffll);
  if (!d) throw Error(MKP_E_DEVICE, "device ingest: no ingest object");
  auto t_lock = std::chrono::steady_clock::now();
  std::lock_guard<std::mutex> lock(d->mu);
  auto t0 = std::chrono::steady_clock::now();
    const double wait_ms = std::chrono::duration<double, std::milli>(t0 - t_lock).count(), alloc0 = mkp_tl_alloc_ms();
  std::unique_ptr<DevShard> out(new DevShard());
  static const bool trace_laps = getenv("MKP_TRACE_PLAN") != nullptr; std::string laps; auto t_lap = t0;
  auto lap = [&](const char* what) { if (trace_laps) { auto now = std::chrono::steady_clock::now(); char b[96];
      snprintf(b, sizeof b, " %s %.1f", what, std::chrono::duration<double, std::milli>(now - t_lap).count()); laps += b; t_lap = now; } };
  ShardHost& S = out->S; S.tid = (int32_t)tid; S.dev_packed = true;
  BamSource::IngestPlan plan; bam.ingest_ranges(tid, parts, &plan);
  if (plan.ranges.empty()) return out;   // nothing under the region: an empty shard
  lap("ranges");
  auto ok = [](hipError_t e, const char* what) {
    if (e != hipSuccess) throw Error(MKP_E_DEVICE, std::string("device ingest: ") + what + ": " + hipGetErrorString(e));
    };
  ok(hipSetDevice(d->device), "hipSetDevice");
  // ---- the compressed ranges go up piece by piece — pread into page-locked staging on all cores, async H2D behind them — on a helper
  // thread, while this one walks the block headers (the upload needs the file ranges only)
  std::vector<uint64_t> zbase(plan.ranges.size()); uint64_t zbytes = 0;
  for (size_t r = 0; r < plan.ranges.size(); r++) { zbase[r] = zbytes; zbytes += (plan.ranges[r].file_len + 63) & ~63ull; }
  d->zin.ensure(zbytes + 64);
  lap("zin");
  struct Piece { uint64_t file_off, z_off; size_t n; };
  std::vector<Piece> pieces;
  static const size_t env_rounds = getenv("MKP_STAGE_ROUNDS") ? strtoull(getenv("MKP_STAGE_ROUNDS"), nullptr, 10) : 0;   // (A/B runs)
  // (kPiece / kStageRounds unless a test shrank them)
  const size_t piece_bytes = g_tune.piece, stage_rounds = env_rounds ? env_rounds : g_tune.stage_rounds;
  for (size_t r = 0; r < plan.ranges.size(); r++) for (uint64_t o = 0; o < plan.ranges[r].file_len; o += piece_bytes)
    pieces.push_back({plan.ranges[r].file_off + o, zbase[r] + o, (size_t)std::min<uint64_t>(piece_bytes, plan.ranges[r].file_len - o)});
  d->stage.ensure(2 * mkp_dev_ingest::kSlots * mkp_dev_ingest::kPiece);
  const int fd = bam.fd();
  lap("zin+staging");
  std::unique_ptr<Error> up_err; double up_ms = 0;
  // The upload goes in ROUNDS of kSlots pieces (16 MiB), an event behind each.  The staged path below cuts its STAGES as it goes: a stage is
  // every round that has been issued when the ingest thread gets to it — so the stages grow with what the upload delivered while the stages
  // before were inflating (C3: 16, ~30, ~100, ~120, ~190, ~250 MiB ...).  An inflate launch takes whole multiples of one block's latency
  // (4.4 ms with 4 096 one-wave workgroups resident), so few large launches beat many equal ones: seven 128 MiB stages took 77 ms of
  // inflate for the C3 file, one launch of everything 58 — but only after the whole 30 ms upload.  `stage_rounds` caps a stage (tests).
  const size_t n_rounds = (pieces.size() + mkp_dev_ingest::kSlots - 1) / mkp_dev_ingest::kSlots;
  std::vector<uint64_t> round_end_z(n_rounds, zbytes);   // where a round's bytes end in zin
  for (size_t r = 0; r + 1 < n_rounds; r++) { const size_t last = (r + 1) * mkp_dev_ingest::kSlots - 1;
    round_end_z[r] = pieces[last].z_off + pieces[last].n; }
  while (d->stage_ev.size() < n_rounds) { hipEvent_t e = nullptr; ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "event");
    d->stage_ev.push_back(e); }
  // (an event that has not been recorded yet does not hold a stream back: the consumer waits for the record call itself)
  std::mutex st_mu; std::condition_variable st_cv; size_t rounds_issued = 0; bool up_finished = false;
  std::thread uploader([&]() {
    auto t_up = std::chrono::steady_clock::now();
    try {
      ok(hipSetDevice(d->device), "hipSetDevice");
      std::atomic<bool> read_bad{false};
      for (size_t p0 = 0, round = 0; p0 < pieces.size(); p0 += mkp_dev_ingest::kSlots, round++) {
        const size_t half = round & 1, n = std::min(mkp_dev_ingest::kSlots, pieces.size() - p0);
        if (round >= 2) ok(hipEventSynchronize(d->slot_ev[half]), "staging wait");   // the copies that last used this half are done
        uint8_t* base = (uint8_t*)d->stage.p + half * mkp_dev_ingest::kSlots * mkp_dev_ingest::kPiece;
        HostPool::get().parallel(n, [&](size_t k) {
          const Piece& pc = pieces[p0 + k]; uint8_t* dst = base + k * mkp_dev_ingest::kPiece; size_t got = 0;
          while (got < pc.n) { const ssize_t r = ::pread(fd, dst + got, pc.n - got, (off_t)(pc.file_off + got)); if (r <= 0) { read_bad = true;
              return; } got += (size_t)r; }
        });
        if (read_bad) throw Error(MKP_E_IO, "read error on " + bam.path());
        // one copy per run of pieces that lie back to back in the staging half AND in the window (a whole round, for a window of one file range):
        // 2 MiB copies do not reach the link's rate, 32 MiB ones do
        for (size_t k = 0; k < n;) {
          size_t k1 = k + 1, bytes = pieces[p0 + k].n;
          while (k1 < n && pieces[p0 + k1 - 1].n == mkp_dev_ingest::kPiece
              && pieces[p0 + k1].z_off == pieces[p0 + k1 - 1].z_off + mkp_dev_ingest::kPiece) {
            bytes += pieces[p0 + k1].n; k1++; }
          ok(hipMemcpyAsync(d->zin.as<uint8_t>() + pieces[p0 + k].z_off, base + k * mkp_dev_ingest::kPiece, bytes, hipMemcpyHostToDevice,
              d->up_stream), "H2D");
          k = k1;
        }
        ok(hipEventRecord(d->slot_ev[half], d->up_stream), "event");
        ok(hipEventRecord(d->stage_ev[round], d->up_stream), "event");
        { std::lock_guard<std::mutex> g(st_mu); rounds_issued = round + 1; } st_cv.notify_all();
      }
      ok(hipEventRecord(d->up_done, d->up_stream), "event");
    } catch (const Error& e) { up_err.reset(new Error(e)); }
    { std::lock_guard<std::mutex> g(st_mu); up_finished = true; } st_cv.notify_all();
    up_ms = ms_since(t_up);
  });
  // (whatever way this function is left: the uploader has stopped and the copies it queued out of the page-locked staging have landed — the
  // next ingest on this object rewrites that staging from its first round on)
  struct JoinUp { std::thread& t; hipStream_t up; ~JoinUp() { if (t.joinable()) t.join(); (void)hipStreamSynchronize(up);
    } } join_up{uploader, d->up_stream};
  // ---- block table.  The device walks the BGZF headers of the uploaded bytes, one thread per chain between block starts the index knows
  // (round 4 walked them on the host with one pread per block: 54 000 preads, 40-90 ms beside an upload that wants the same cores);
  // MKP_HOST_BLOCK_TABLE=1 keeps the host walk (A/B runs).
  static const bool host_table = getenv("MKP_HOST_BLOCK_TABLE") && !strcmp(getenv("MKP_HOST_BLOCK_TABLE"), "1");
  if (host_table) { bam.ingest_blocks(&plan); out->ms_plan = ms_since(t0); uploader.join(); if (up_err) throw *up_err; out->ms_upload = up_ms; }
  bool staged_done = false; double staged_kernel_ms = 0; std::chrono::steady_clock::time_point t_inf_staged;
    hipEvent_t last_inf[2] = {nullptr, nullptr}; size_t staged_stages = 0; std::vector<uint32_t> staged_nblk;
  if (!host_table) {
    // ---- the STAGED path: the window goes up in rounds of 16 MiB, cut into stages as they arrive (above), and a stage's blocks are found, laid out
    // and inflated while the next stage is still on its way (round 4: whole upload, then block table, then one inflate launch — the GPU idle for the
    // 40-90 ms of the
    // upload, the host idle for the inflate).  Per stage, on the ingest stream: wait for the stage's last copy; walk its chains
    // (count -> scan; ONE host sync for the block count, which sizes the launches); write its MkpZBlk entries; mkp_bgzf_layout turns them
    // into the inflate's table behind a device-side cursor of the inflated window; inflate; CRC on a stream of its own.  The inflated
    // size is not known before the last stage: the window buffer is sized at 6 x the compressed bytes and the layout kernel reports an
    // overflow, on which the whole window is inflated again into an exact allocation (below).
    // (whatever leaves this block by exception must not leave kernels reading buffers the next ingest rewrites)
    struct Drain { mkp_dev_ingest* d; int n = std::uncaught_exceptions(); ~Drain() { if (std::uncaught_exceptions() > n) {
          (void)hipStreamSynchronize(d->stream); for (auto& st : d->inf_stream) (void)hipStreamSynchronize(st);
          (void)hipStreamSynchronize(d->crc_stream); } } } drain{d};
    std::vector<BamSource::IngestChain> chains; bam.ingest_chains(plan, &chains);
    const size_t nc = chains.size();
    lap("chains");
    if (nc > 0xfffffff0ull) throw Error(MKP_E_UNSUPPORTED, "shard window holds too many BGZF chains; use smaller shards");
    // the chain table is written into page-locked memory and read from there by the chain kernels (115 KB for a chr20 window): as a copy it
    // waited on the copy engine behind the upload pieces already queued — 44 ms in which no stage could start (round 6 trace)
    d->chain_host.ensure(std::max<size_t>(nc, 1) * sizeof(MkpZChain));
    MkpZChain* zc = (MkpZChain*)d->chain_host.p; std::vector<uint64_t> chain_zend(nc);   // chain_zend: the chain reads nothing at or behind this
    for (size_t i = 0; i < nc; i++) { const BamSource::IngestRange& rg = plan.ranges[chains[i].range];
      const uint64_t zb = zbase[chains[i].range], fo = rg.file_off;
      MkpZChain c; c.start = zb + (chains[i].start - fo); c.stop = chains[i].stop == UINT64_MAX ? ~0ull : zb + (chains[i].stop - fo);
        c.range_end = zb + rg.file_len;
      c.ce = (rg.vend >> 16) >= fo ? zb + ((rg.vend >> 16) - fo) : 0; c.ue = (uint32_t)(rg.vend & 0xffff); c.pad = 0; zc[i] = c;
      chain_zend[i] = c.stop == ~0ull ? c.range_end : std::min<uint64_t>(c.stop, c.range_end); }
    std::vector<size_t> stage_c0(1, 0);   // stage j = chains [stage_c0[j], stage_c0[j + 1]): those that end inside the rounds the stage took
    out->ms_plan = ms_since(t0);
    lap("chain table");
    // the window's capacity as the layout kernel enforces it: a block that would end behind it gets no room at all (out_len 0: the inflate and
    // the CRC of its stage touch nothing), the overflow bit goes up, and the stages after it only build their tables
    const uint64_t raw_cap = g_tune.raw_cap ? g_tune.raw_cap : std::max<uint64_t>(d->raw.cap, plan.comp_total * 6 + (64ull << 20));
    d->raw.ensure(raw_cap); lap("raw window"); d->chain_cnt.ensure((nc + n_rounds + 2) * 4); d->tot.ensure(sizeof(MkpIngestTotals));
      d->rawcur.ensure(16);
    d->small.ensure(4096);
    uint32_t* h_small = (uint32_t*)d->small.p;   // [0] error bits, [1] blocks of the stage; [2..3] the cursor of the inflated window (at the end)
    auto stage_events = [&](size_t j) {   // (created as the stages come: their number is not known ahead)
      while (d->tev.size() < 2 * (j + 1)) { hipEvent_t e = nullptr; ok(hipEventCreate(&e), "event"); d->tev.push_back(e); }
      while (d->lay_ev.size() < j + 1) { hipEvent_t e = nullptr; ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "event");
        d->lay_ev.push_back(e); } };
    lap("small buffers + events");   // (a temporary: through the library's page-locked staging, mkp_ctx.hpp)
    ok(hipMemsetAsync(d->tot.p, 0, sizeof(MkpIngestTotals), d->stream), "memset");
    ok(hipMemsetAsync(d->rawcur.p, 0, 16, d->stream), "memset");
    size_t blk_cap = std::max<size_t>({d->zblk.cap / sizeof(BgzfBlk), d->ztab.cap / sizeof(MkpZBlk), (size_t)(plan.comp_total / 8192 + 4096)});
    d->zblk.ensure(blk_cap * sizeof(BgzfBlk)); d->ztab.ensure(blk_cap * sizeof(MkpZBlk)); d->zstat.ensure(blk_cap * 4 + 16);
    blk_cap = std::min({d->zblk.cap / sizeof(BgzfBlk), d->ztab.cap / sizeof(MkpZBlk), (d->zstat.cap - 16) / 4});
    // (a window of unusually small blocks: the tables grow, keeping what the stages before wrote)
    auto grow = [&](DevBuf& b, size_t need, size_t keep) {
      DevBuf nb; nb.ensure(need); ok(hipStreamSynchronize(d->stream), "sync"); for (auto& st : d->inf_stream) ok(hipStreamSynchronize(st), "sync");
        ok(hipStreamSynchronize(d->crc_stream), "sync");
      if (keep) ok(hipMemcpy(nb.p, b.p, keep, hipMemcpyDeviceToDevice), "D2D"); b.release(); b = nb; nb.p = nullptr; nb.cap = 0; };
    size_t blkbase = 0; std::vector<uint32_t> stage_nblk; std::vector<hipEvent_t> inf_end; size_t r_done = 0;
    t_inf_staged = std::chrono::steady_clock::now();
    lap("buffers");
    for (size_t j = 0; r_done < n_rounds; j++) {
      // pacing: a stage is cut when the one before it has inflated — cut earlier it would be a smaller one (with one stage queued behind the
      // running one, MKP_STAGE_DEPTH=2, the C3 file took 5-6 launches and 73-76 ms of inflate; so: 4 launches, 66-70 ms, ~1 ms idle between them)
      // (A/B runs)
      static const size_t depth = getenv("MKP_STAGE_DEPTH") ? std::max<size_t>(1, strtoull(getenv("MKP_STAGE_DEPTH"), nullptr, 10)) : 1;
      if (inf_end.size() >= depth) ok(hipEventSynchronize(inf_end[inf_end.size() - depth]), "stage pacing");
      size_t r_now;
      { std::unique_lock<std::mutex> lk(st_mu); st_cv.wait(lk, [&] { return rounds_issued > r_done || up_finished; }); r_now = rounds_issued; }
      if (trace_laps) { char b[64]; snprintf(b, sizeof b, " [s%zu: rounds %zu-%zu, issued", j, r_done, r_now); lap(b); }
      if (up_err || r_now <= r_done) break;   // (the uploader stopped early: its error is reported below)
      if (stage_rounds && r_now - r_done > stage_rounds) r_now = r_done + stage_rounds;
      ok(hipStreamWaitEvent(d->stream, d->stage_ev[r_now - 1], 0), "wait for the upload rounds");
      stage_events(j);
      const uint64_t z_end = round_end_z[r_now - 1];
      const size_t c0 = stage_c0[j]; size_t c1 = c0; while (c1 < nc && (r_now == n_rounds || chain_zend[c1] <= z_end)) c1++;
      stage_c0.push_back(c1); stage_nblk.push_back(0); r_done = r_now;
      const size_t n = c1 - c0;
      uint32_t* cnt = (uint32_t*)d->chain_cnt.p + c0 + j;   // the stage's counts -> offsets, its total behind them
      ok(mkp_launch_bgzf_chain_count(d->stream, d->zin.as<uint8_t>(), zc + c0, (uint32_t)n, cnt, d->tot.as<uint32_t>()), "block table launch");
      ok(hipMemcpyAsync(h_small, d->tot.p, 4, hipMemcpyDeviceToHost, d->stream), "D2H");
        ok(hipMemcpyAsync(h_small + 1, cnt + n, 4, hipMemcpyDeviceToHost, d->stream), "D2H");
      ok(hipStreamSynchronize(d->stream), "block table sync");
      lap("count");
      if (h_small[0] & (MKP_ZE_BAD | MKP_ZE_CHAIN)) throw Error(MKP_E_IO,
          "bad BGZF block in " + bam.path() + " (or the index does not match the file)");
      // an earlier stage ran out of window: the rest is inflated below, into an exact allocation
      const bool outgrown = (h_small[0] & MKP_ZE_RAWCAP) != 0;
      const uint32_t nblk = h_small[1]; stage_nblk[j] = nblk;
      if (blkbase + nblk > 0xfffffff0ull) throw Error(MKP_E_UNSUPPORTED, "shard window holds too many BGZF blocks; use smaller shards");
      if (blkbase + nblk > blk_cap) { const size_t want = std::max<size_t>(2 * blk_cap, blkbase + nblk + 4096);
        grow(d->zblk, want * sizeof(BgzfBlk), blkbase * sizeof(BgzfBlk)); { Pinned nt; nt.ensure(want * sizeof(MkpZBlk));
          ok(hipStreamSynchronize(d->stream), "sync"); if (blkbase) memcpy(nt.p, d->ztab.p, blkbase * sizeof(MkpZBlk)); d->ztab.release();
          d->ztab = nt; } grow(d->zstat, want * 4 + 16, blkbase * 4); blk_cap = want; }
      if (!nblk) continue;
      MkpZBlk* ztab = (MkpZBlk*)d->ztab.p + blkbase; BgzfBlk* zblk = d->zblk.as<BgzfBlk>() + blkbase;
        uint32_t* zst = d->zstat.as<uint32_t>() + blkbase;
      ok(mkp_launch_bgzf_chain_write(d->stream, d->zin.as<uint8_t>(), zc + c0, (uint32_t)n, cnt, nblk, ztab, d->tot.as<uint32_t>()),
          "block table launch");
      ok(mkp_launch_bgzf_layout(d->stream, ztab, nblk, d->rawcur.as<unsigned long long>(), raw_cap, zblk, d->tot.as<uint32_t>()), "layout launch");
      ok(hipMemsetAsync(zst, 0xff, (size_t)nblk * 4, d->stream), "memset");
      // the inflate itself goes to a stream of its own: the ingest stream stays free for the next stage's chain walk, the host's one sync per
      // stage no longer waits for an inflate, and the tables come back (and the window is laid out) while the last stages still inflate
      hipStream_t inf = d->inf_stream[0];
      ok(hipEventRecord(d->lay_ev[j], d->stream), "event");
      ok(hipStreamWaitEvent(inf, d->lay_ev[j], 0), "wait for the stage's tables");
      ok(hipEventRecord(d->tev[2 * j], inf), "event");
      if (!outgrown) ok(mkp_launch_inflate_auto(inf, d->zin.as<uint8_t>(), zblk, nblk, d->raw.as<uint8_t>(), zst), "inflate launch");
      ok(hipEventRecord(d->tev[2 * j + 1], inf), "event");
      last_inf[0] = d->tev[2 * j + 1]; inf_end.push_back(d->tev[2 * j + 1]);
      if (!outgrown) {
        ok(hipStreamWaitEvent(d->crc_stream, d->tev[2 * j + 1], 0), "wait for the inflate");
        ok(mkp_launch_crc32(d->crc_stream, d->zin.as<uint8_t>(), zblk, nblk, d->raw.as<uint8_t>(), zst), "crc launch");
      }
      blkbase += nblk;
    }
    lap("stages issued");
    uploader.join();
    lap("uploader joined");
    if (up_err) { (void)hipStreamSynchronize(d->stream); for (auto& st : d->inf_stream) (void)hipStreamSynchronize(st);
      (void)hipStreamSynchronize(d->crc_stream); throw *up_err; }
    out->ms_upload = up_ms;
    // the whole table comes back for the window layout the record kernels need (entry points of the chains) and for error reports
    const MkpZBlk* zb = (const MkpZBlk*)d->ztab.p; const uint32_t* cbase_all = (const uint32_t*)d->chain_cnt.p;   // (complete behind the sync below)
    ok(hipMemcpyAsync(h_small, d->tot.p, 4, hipMemcpyDeviceToHost, d->stream), "D2H");
      ok(hipMemcpyAsync(h_small + 2, d->rawcur.p, 8, hipMemcpyDeviceToHost, d->stream), "D2H");
    ok(hipStreamSynchronize(d->stream), "inflate sync");
    lap("tables back + sync");
    const size_t n_stages = stage_nblk.size();
    staged_stages = n_stages; staged_nblk = stage_nblk;
    const uint32_t zerr = h_small[0];
    if (zerr & (MKP_ZE_BAD | MKP_ZE_CHAIN)) throw Error(MKP_E_IO, "bad BGZF block in " + bam.path() + " (or the index does not match the file)");
    if (zerr & MKP_ZE_ISIZE) throw Error(MKP_E_IO, "BGZF block inflates to more than 64 KiB in " + bam.path());
    std::vector<std::vector<BamSource::IngestBlk>> parts(nc);
    { size_t base = 0;
      for (size_t j = 0; j < n_stages; j++) { const size_t c0 = stage_c0[j], c1 = stage_c0[j + 1]; const uint32_t* cb = cbase_all + c0 + j;
        for (size_t i = c0; i < c1; i++) { const uint64_t zbs = zbase[chains[i].range], fo = plan.ranges[chains[i].range].file_off;
          parts[i].reserve(cb[i - c0 + 1] - cb[i - c0]);
          for (uint32_t k = cb[i - c0]; k < cb[i - c0 + 1]; k++) { const MkpZBlk& z = zb[base + k];
            parts[i].push_back({fo + (z.coff - zbs), z.hdr, z.clen, z.isize, 0}); } }
        base += stage_nblk[j]; } }
    bam.ingest_layout(&plan, chains, parts);
    lap("layout");
    unsigned long long cur; memcpy(&cur, h_small + 2, 8);
    // (a window larger than the estimate: inflated again below, into an exact allocation)
    staged_done = !(zerr & MKP_ZE_RAWCAP) && cur == plan.raw_total && plan.blks.size() == blkbase;
    if (!staged_done) { for (auto& st : d->inf_stream) ok(hipStreamSynchronize(st), "sync"); ok(hipStreamSynchronize(d->crc_stream), "sync");
      if (!(zerr & MKP_ZE_RAWCAP)) throw Error(MKP_E_DEVICE, "internal: the device's window layout differs from the host's");
      g_reinflated++; }
    else g_staged_windows++;
  }
  bam.bytes_read += plan.comp_total;
  if (plan.raw_total == 0) return out;
  if (plan.blks.size() > 0xfffffff0ull || plan.entries.size() > 0xfffffff0ull) throw Error(MKP_E_UNSUPPORTED,
      "shard window holds too many BGZF blocks; use smaller shards");
  // ---- tables: BGZF blocks, chain segments
  const size_t nb = plan.blks.size(), ns = plan.entries.size();
  std::vector<BgzfBlk> blks(nb);
  for (size_t r = 0; r < plan.ranges.size(); r++) for (size_t k = plan.ranges[r].blk0; k < plan.ranges[r].blk1; k++) {
    const BamSource::IngestBlk& b = plan.blks[k];
    if (b.isize > 65536u) throw Error(MKP_E_IO, "BGZF block inflates to more than 64 KiB in " + bam.path());
    blks[k] = {zbase[r] + (b.coff - plan.ranges[r].file_off) + b.hdr, b.doff, b.clen, b.isize};
  }
  const std::vector<MkpSeg> segs = mkp_plan_segments<MkpSeg>(plan);
  lap("segments");
  d->zblk.ensure(nb * sizeof(BgzfBlk)); d->zstat.ensure(nb * 4 + 16); d->raw.ensure(plan.raw_total + 64); d->segs.ensure(ns * sizeof(MkpSeg));
    d->seg_cnt.ensure((ns + 1) * 4); d->tot.ensure(sizeof(MkpIngestTotals));
  const size_t small_need = nb * sizeof(BgzfBlk) + ns * sizeof(MkpSeg) + 2 * (nb * 4 + 64) + sizeof(MkpIngestTotals) + 256;
  d->small.ensure(small_need + small_need / 4);
  uint8_t* sm = (uint8_t*)d->small.p; uint8_t* sm_blk = sm; uint8_t* sm_seg = sm + nb * sizeof(BgzfBlk);
    uint8_t* sm_tot = sm_seg + ns * sizeof(MkpSeg); uint8_t* sm_stat = sm_tot + ((sizeof(MkpIngestTotals) + 63) & ~(size_t)63);
    uint8_t* sm_stat0 = sm_stat + ((nb * 4 + 63) & ~(size_t)63);
  memcpy(sm_blk, blks.data(), nb * sizeof(BgzfBlk)); memcpy(sm_seg, segs.data(), ns * sizeof(MkpSeg));
  ok(hipMemcpyAsync(d->segs.p, sm_seg, ns * sizeof(MkpSeg), hipMemcpyHostToDevice, d->stream), "H2D");
  ok(hipMemsetAsync(d->tot.p, 0, sizeof(MkpIngestTotals), d->stream), "memset");
  auto t_inf = staged_done ? t_inf_staged : std::chrono::steady_clock::now();
  if (!staged_done) {   // the whole window in one launch: the host-table path, or a window that outgrew the staged path's estimate
    ok(hipMemcpyAsync(d->zblk.p, sm_blk, nb * sizeof(BgzfBlk), hipMemcpyHostToDevice, d->stream), "H2D");
    ok(hipMemsetAsync(d->zstat.p, 0xff, nb * 4, d->stream), "memset");
    ok(hipStreamWaitEvent(d->stream, d->up_done, 0), "wait for the upload");
    // ---- inflate + CRC
    ok(hipEventRecord(d->kev[0], d->stream), "event");
    ok(mkp_launch_inflate_auto(d->stream, d->zin.as<uint8_t>(), d->zblk.p, (uint32_t)nb, d->raw.as<uint8_t>(), d->zstat.as<uint32_t>()),
        "inflate launch");
    // the CRC-32 of every block runs on a stream of its own, beside the record kernels below: its verdict — and the decoders'
    // status words it is OR-ed into — is only read when something has gone wrong, or at the very end (corrupt())
    ok(hipEventRecord(d->inf_done, d->stream), "event");
    ok(hipStreamWaitEvent(d->crc_stream, d->inf_done, 0), "wait for the inflate");
    ok(mkp_launch_crc32(d->crc_stream, d->zin.as<uint8_t>(), d->zblk.p, (uint32_t)nb, d->raw.as<uint8_t>(), d->zstat.as<uint32_t>()), "crc launch");
  } else {   // the record kernels read what the stages' inflate launches wrote
    (void)copy_stage();   // (this thread's page-locked copy staging, needed for the digest: allocated while the last stages inflate)
    for (hipEvent_t e : last_inf) if (e) ok(hipStreamWaitEvent(d->stream, e, 0), "wait for the inflate");
    ok(hipEventRecord(d->kev[0], d->stream), "event");
  }
  // ---- record chains
  MkpIngestParams P; memset(&P, 0, sizeof(P));
  P.raw_len = plan.raw_total; P.tid = (int32_t)tid; P.beg = (int32_t)std::min<uint32_t>(beg, 0x7fffffffu);
    P.end = (int32_t)std::min<uint32_t>(end, 0x7fffffffu); P.n_ref = (int32_t)bam.ref_names.size(); P.n_seg = (uint32_t)ns;
  ok(mkp_launch_ingest_count(d->stream, d->raw.as<uint8_t>(), &P, d->segs.as<MkpSeg>(), d->seg_cnt.as<uint32_t>(), d->tot.as<MkpIngestTotals>()),
      "count launch");
  MkpIngestTotals* tot = (MkpIngestTotals*)sm_tot;
  ok(hipEventRecord(d->kev[1], d->stream), "event");
  ok(hipMemcpyAsync(sm_stat, d->zstat.p, nb * 4, hipMemcpyDeviceToHost, d->crc_stream), "D2H");
  ok(hipEventRecord(d->crc_done, d->crc_stream), "event");
  // the decoders' own status (low byte; the CRC kernel may be OR-ing bit 8 in meanwhile)
  ok(hipMemcpyAsync(sm_stat0, d->zstat.p, nb * 4, hipMemcpyDeviceToHost, d->stream), "D2H");
  ok(hipMemcpyAsync(tot, d->tot.p, sizeof(MkpIngestTotals), hipMemcpyDeviceToHost, d->stream), "D2H");
  ok(hipStreamSynchronize(d->stream), "inflate sync");
  out->ms_inflate = ms_since(t_inf);
  lap("count+sync");
  // the stages' inflate launches overlap (two streams): what is reported is the span from the first launch's start to the last one's end
  if (staged_done) {
    size_t j0 = SIZE_MAX; float span = 0;
    for (size_t j = 0; j < staged_stages; j++) if (staged_nblk[j]) { if (j0 == SIZE_MAX) j0 = j; float b = 0;
      if (hipEventElapsedTime(&b, d->tev[2 * j0], d->tev[2 * j + 1]) == hipSuccess) span = std::max(span, b);
      if (trace_laps) { float a = 0; (void)hipEventElapsedTime(&a, d->tev[2 * j0], d->tev[2 * j]); char t[64];
        snprintf(t, sizeof t, " {inflate %zu: %.1f-%.1f}", j, a, b); laps += t; } }
    staged_kernel_ms = span;
  }
  // (staged: the stages' inflate launches + the chain kernels)
  { float kms = 0; if (hipEventElapsedTime(&kms, d->kev[0], d->kev[1]) == hipSuccess) out->ms_kernel = kms + staged_kernel_ms; }
  bam.bytes_inflated += plan.raw_total; bam.bytes_inflated_device += plan.raw_total;
  // whatever leaves this function early must not leave the CRC kernel reading buffers the next ingest rewrites
  struct CrcJoin { hipEvent_t ev; ~CrcJoin() { (void)hipEventSynchronize(ev); } } crc_join{d->crc_done};
  auto corrupt = [&]() {   // block decoder status / CRC verdict: a corrupt block is what gets reported, whatever the record kernels made of its bytes
    ok(hipEventSynchronize(d->crc_done), "crc sync");
    const uint32_t* st = (const uint32_t*)sm_stat;
      for (size_t i = 0; i < nb; i++) if (st[i] != 0) throw Error(MKP_E_IO, "corrupt BGZF data in " + bam.path() +
        ((st[i] & 0x100u)
            ? " (CRC32 mismatch" : " (decoder status " + std::to_string(st[i] & 0xffu)) + ", block at " + std::to_string(plan.blks[i].coff) + ")");
          };
  // a block that did not inflate: nothing behind it is worth parsing
  { const uint32_t* st0 = (const uint32_t*)sm_stat0; bool bad = false; for (size_t i = 0; i < nb && !bad; i++) bad = (st0[i] & 0xffu) != 0;
    if (bad) corrupt();
    }
  auto check = [&](uint32_t err) {
    if (err) corrupt();
    if (err & MKP_IE_TRUNCATED) throw Error(MKP_E_IO, "truncated BAM record at the end of " + bam.path());
    if (err & MKP_IE_CORRUPT) throw Error(MKP_E_IO, "corrupt BAM record");
    if (err & MKP_IE_CHAIN) throw Error(MKP_E_IO,
        "the BAM index does not match the file (a record chain misses an indexed record start): " + bam.path() + ".bai");
    if (err & (MKP_IE_TABLE | MKP_IE_4G)) throw Error(MKP_E_UNSUPPORTED, "shard exceeds 4 GiB of packed bases; use smaller shards");
    if (err & MKP_IE_QLEN) throw Error(MKP_E_INVALID, "CIGAR query length does not match SEQ length");
    if (err & MKP_IE_SPAN) throw Error(MKP_E_UNSUPPORTED,
        "a read or its alignment spans 2^26 bases or more (the depth walk packs query offsets in 27 bits)");
    if (err & MKP_IE_NONASCII) throw Error(MKP_E_UNSUPPORTED, "non-ASCII mod code");
    if (err & MKP_IE_CODES) throw Error(MKP_E_UNSUPPORTED, "more than 4 mod codes in one MM tag");
    if (err & MKP_IE_TAGS) throw Error(MKP_E_UNSUPPORTED, "more than 8 MM tags in one read");
  };
  check(tot->err);
  // ---- records: offsets, checks + region test + aux walk, sizes -> offsets
  auto t_scan = std::chrono::steady_clock::now();
  const uint32_t n_all = tot->n_all;
  P.rec_cap = std::max<uint32_t>(n_all, 1u);
  d->rec_off.ensure((size_t)P.rec_cap * 8); d->info.ensure((size_t)P.rec_cap * sizeof(MkpRecInfo)); d->sz.ensure(6 * (size_t)P.rec_cap * 4);
    d->extra.ensure(2 * (size_t)P.rec_cap * 4);
  const int32_t* d_parts = nullptr;
  if (parts.size() > 1) {   // the windows of a multi-part fetch, next to the params
    std::vector<int32_t> pv; pv.reserve(2 * parts.size()); for (auto& pr : parts) { pv.push_back((int32_t)std::max<int64_t>(pr.first, 0));
      pv.push_back((int32_t)std::min<int64_t>(pr.second, 0x7fffffffll)); }
    d->parts.ensure(pv.size() * 4); ok(hipMemcpyAsync(d->parts.p, pv.data(), pv.size() * 4, hipMemcpyHostToDevice, d->stream), "H2D");
      ok(hipStreamSynchronize(d->stream), "sync");
    P.n_parts = (uint32_t)parts.size(); d_parts = d->parts.as<int32_t>();
  }
  ok(mkp_launch_ingest_parse(d->stream, d->raw.as<uint8_t>(), &P, d_parts, d->segs.as<MkpSeg>(), d->seg_cnt.as<uint32_t>(),
      d->rec_off.as<unsigned long long>(), d->info.as<MkpRecInfo>(), d->sz.as<uint32_t>(),
                             d->extra.as<int32_t>(), d->tot.as<MkpIngestTotals>()), "parse launch");
  ok(hipMemcpyAsync(tot, d->tot.p, sizeof(MkpIngestTotals), hipMemcpyDeviceToHost, d->stream), "D2H");
  ok(hipStreamSynchronize(d->stream), "parse sync");
  check(tot->err);
  const uint32_t n = tot->n_kept, n_so = tot->n_sample_only, n_pk = n + n_so;
  // ---- pack
  out->d_cigar = d->take((tot->cigar_words + 16) * 4); out->d_chunk = d->take((tot->chunk_pairs + 4) * 8); out->d_seq = d->take(tot->seq_bytes + 64);
  out->d_tagref = d->take(((size_t)n_pk * MKP_MAX_TAGS + 1) * sizeof(MkpTagRef)); out->d_ranks = d->take((tot->ml_bytes + 16) * 4);
    out->d_ml = d->take(tot->ml_bytes + 64);
  DevBuf d_hdr = d->take(((size_t)n_pk + 1) * sizeof(MkpReadHdr));
  struct Back { mkp_dev_ingest* d; DevBuf b; ~Back() { std::lock_guard<std::mutex> g(d->spare_mu); if (b.p) {
        if (d->spares.size() < 16) d->spares.push_back(b);
        else b.release();
      } } } back{d, d_hdr};
  d->dig.ensure(((size_t)n_pk + 1) * sizeof(MkpRecDigest));
  ok(mkp_launch_ingest_pack(d->stream, d->raw.as<uint8_t>(), P.rec_cap, d->info.as<MkpRecInfo>(), d->sz.as<uint32_t>(), d_hdr.as<MkpReadHdr>(),
      out->d_cigar.as<uint32_t>(), out->d_chunk.as<uint32_t>(),
                            out->d_seq.as<uint8_t>(), out->d_tagref.as<MkpTagRef>(), out->d_ranks.as<uint32_t>(), out->d_ml.as<uint8_t>(),
                                d->dig.as<MkpRecDigest>(), d->tot.as<MkpIngestTotals>()), "pack launch");
  S.hdr.resize(n); S.so_hdr.resize(n_so); S.tagref.resize((size_t)n_pk * MKP_MAX_TAGS); S.name_hash.resize(n);
  std::vector<MkpRecDigest> dig(n_pk); std::vector<int32_t> extra(2 * (size_t)tot->n_extra);
  // (into pageable vectors through the library's page-locked staging: handed these directly, the runtime pins them in place, and their
  //  release — with the shard — stalls every queue of the process; mkp_ctx.hpp)
  d2h_copy(S.hdr.data(), d_hdr.p, (size_t)n * sizeof(MkpReadHdr), d->stream);
  d2h_copy(S.so_hdr.data(), d_hdr.as<MkpReadHdr>() + n, (size_t)n_so * sizeof(MkpReadHdr), d->stream);
  if (n_pk) { d2h_copy(S.tagref.data(), out->d_tagref.p, (size_t)n_pk * MKP_MAX_TAGS * sizeof(MkpTagRef), d->stream);
              d2h_copy(dig.data(), d->dig.p, (size_t)n_pk * sizeof(MkpRecDigest), d->stream); }
  d2h_copy(extra.data(), d->extra.p, extra.size() * 4, d->stream);
  ok(hipMemcpyAsync(tot, d->tot.p, sizeof(MkpIngestTotals), hipMemcpyDeviceToHost, d->stream), "D2H");
  ok(hipStreamSynchronize(d->stream), "pack sync");
  check(tot->err);
  corrupt();
  out->ms_pack = ms_since(t_scan);
  lap("parse+pack");
  // ---- digest -> what the planner reads: layout ids (this shard's own table; mkp_internal_shard_attach maps them into the context's), flags
  auto t_dig = std::chrono::steady_clock::now();
  S.n_calls = tot->n_calls; S.dev_n_ranks = tot->n_calls; S.dev_n_ml = tot->n_ml_used;
  S.dev_sum2.resize(n); S.dev_name_hash2.resize(n); S.dev_win_idx.resize(n); S.so_name_hash.resize(n_so); S.so_name_hash2.resize(n_so);
    S.so_win_idx.resize(n_so);
  for (size_t k = 0; k < extra.size(); k += 2) S.extra_spans.push_back({extra[k], extra[k + 1]});
  std::unordered_map<uint64_t, uint16_t> by_hash; std::vector<uint8_t> recbuf;
  uint64_t ev_cap = 0, last_hash = 0; uint16_t last_id = 0; bool have_last = false;
  for (uint32_t j = 0; j < n_pk; j++) {
    const bool so = j >= n;
    MkpReadHdr& h = so ? S.so_hdr[j - n] : S.hdr[j];
    if (so) { S.so_name_hash[j - n] = dig[j].name_hash; S.so_name_hash2[j - n] = dig[j].name_hash2; S.so_win_idx[j - n] = (uint32_t)dig[j].win_idx;
      h.pad = 0; }
    else { S.name_hash[j] = dig[j].name_hash; S.dev_name_hash2[j] = dig[j].name_hash2; S.dev_win_idx[j] = (uint32_t)dig[j].win_idx;
      S.dev_sum2[j] = (uint8_t)(h.pad & 1u); h.pad = 0; ev_cap += h.event_cap; }
    if (!h.n_tags || (h.flags & MKP_RF_BAD)) continue;
    if (have_last && dig[j].key_hash == last_hash) { h.layout = last_id; continue; }   // (runs of one structure: most of a file)
    auto it = by_hash.find(dig[j].key_hash);
    if (it == by_hash.end()) {
      // a structure not seen in this shard yet: its record comes back from HBM and goes through the host packer, which interns the layout
      // (and must arrive at the same key: a colliding hash would otherwise attach the wrong caller tables)
      if (out->info_host.empty()) { out->info_host.resize(n_all);
        d2h_copy(out->info_host.data(), d->info.p, (size_t)n_all * sizeof(MkpRecInfo), d->stream); }
      const uint64_t wi = dig[j].win_idx;
      if (wi >= n_all || (out->info_host[wi].kind != 1 && out->info_host[wi].kind != 3)) throw Error(MKP_E_DEVICE,
          "internal: device ingest digest points at a record it did not pack");
      const MkpRecInfo ri = out->info_host[wi];
      recbuf.resize((size_t)ri.bs + 4);
      ok(hipMemcpyAsync(recbuf.data(), d->raw.as<uint8_t>() + (ri.core - 4), recbuf.size(), hipMemcpyDeviceToHost, d->stream), "D2H (record)");
        ok(hipStreamSynchronize(d->stream), "sync");
      mkp_record r; const uint8_t* c = recbuf.data() + 4;
      memcpy(&r.tid, c, 4); memcpy(&r.pos, c + 4, 4); r.l_qname = c[8]; uint16_t nc; memcpy(&nc, c + 12, 2); r.n_cigar = nc;
        memcpy(&r.flag, c + 14, 2); memcpy(&r.l_qseq, c + 16, 4);
      r.l_data = (int32_t)ri.bs - 32; r.data = c + 32;
      ShardHost scratch; scratch.tid = (int32_t)tid;
      const size_t before = out->layouts.layouts.size();
      out->layouts.add(r, scratch);
      if (scratch.hdr.size() != 1 || (scratch.hdr[0].flags & MKP_RF_BAD) || !scratch.hdr[0].n_tags) throw Error(MKP_E_DEVICE,
          "internal: device and host tokenisers disagree on a record's tags");
      const uint16_t id = scratch.hdr[0].layout;
      if (fnv64(out->layouts.layout_keys[id]) != dig[j].key_hash) throw Error(MKP_E_DEVICE,
          "internal: device and host tokenisers disagree on a record's MM header structure");
      if (id < before) throw Error(MKP_E_DEVICE, "internal: two MM header structures share a 64-bit key hash");
      it = by_hash.emplace(dig[j].key_hash, id).first;
    }
    h.layout = it->second; last_hash = dig[j].key_hash; last_id = it->second; have_last = true;
  }
  S.n_events_cap = ev_cap;
  std::vector<MkpRecInfo>().swap(out->info_host);
  out->ms_digest = ms_since(t_dig);
  lap("digest");
  out->n_blocks = nb; out->n_segments = ns; out->n_records = n_all; out->raw_bytes = plan.raw_total; out->comp_bytes = plan.comp_total;
  out->ms_total = ms_since(t0); out->ms_alloc = mkp_tl_alloc_ms() - alloc0; out->ms_wait = wait_ms;
  if (getenv("MKP_TRACE_PLAN")) fprintf(stderr,
      "[mkpileup ingest] tid %u [%u, %u) in %zu window(s): %zu blocks, %zu segments, %u records (%u kept), %.1f MB -> %.1f MB; plan %.1f upload %.1f inflate+chains %.1f (kernels %.1f) parse+pack %.1f digest %.1f total %.1f ms, of which hipMalloc/hipFree %.1f; began at %.1f; laps:%s\n",
      tid, beg, end, parts.size(), nb, ns, n_all, n, plan.comp_total / 1e6, plan.raw_total / 1e6, out->ms_plan, out->ms_upload, out->ms_inflate,
          out->ms_kernel, out->ms_pack, out->ms_digest, out->ms_total, out->ms_alloc,
      std::chrono::duration<double,
          std::milli>(t0.time_since_epoch()).count() - 1000.0 * std::floor(std::chrono::duration<double>(t0.time_since_epoch()).count() / 100.0) * 100.0,
          laps.c_str());
  return out;
}
