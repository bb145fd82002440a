// Internal (not installed): the context behind the opaque mkp_ctx handle, shared by mkp_api.cpp and
// mkp_driver.cpp.
#pragma once
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <mutex>
#include <vector>

#include "mkp_pack.hpp"

namespace mkp {

inline double& mkp_tl_alloc_ms() { static thread_local double v = 0; return v; }   // what this thread spent in hipMalloc / hipFree (traces)
struct DevBuf {
  void* p = nullptr; size_t cap = 0;
  struct Clock { std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now(); ~Clock() {
      mkp_tl_alloc_ms() += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); } };
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    Clock clk;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; throw Error(MKP_E_NOMEM, "hipMalloc of " + std::to_string(want) + " bytes failed"); }
    cap = want;
  }
  void release() { if (p) { Clock clk; (void)hipFree(p); } p = nullptr; cap = 0; }
  template <class T> T* as() const { return (T*)p; }
};

inline void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw Error(MKP_E_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

// Streams are recycled process-wide.  Creating one costs 8-10 ms on this stack (an HSA queue each: tools/dbg/api_cost.hip), and a context with
// its ingest object holds five of them, a thread's copy staging one more: 45-60 ms of every fresh context, before its first byte moves.
// A destroyed context hands its (drained) streams back; the next one of the same device, flags and priority takes them.  The pool is never
// torn down (the runtime may be gone by the time static destructors run).
struct StreamPool {
  struct Item { int dev; unsigned flags; int prio; hipStream_t st; };
  std::mutex mu; std::vector<Item> free;
  static StreamPool& get() { static StreamPool* p = new StreamPool(); return *p; }
};
inline hipError_t pooled_stream_create(hipStream_t* out, int dev, unsigned flags, int prio) {
  { StreamPool& P = StreamPool::get(); std::lock_guard<std::mutex> g(P.mu);
    for (size_t i = 0; i < P.free.size(); i++) if (P.free[i].dev == dev && P.free[i].flags == flags && P.free[i].prio == prio) { *out = P.free[i].st;
      P.free.erase(P.free.begin() + (ptrdiff_t)i); return hipSuccess; } }
  return hipStreamCreateWithPriority(out, flags, prio);
}
inline void pooled_stream_release(hipStream_t st, int dev, unsigned flags, int prio) {
  if (!st) return;
  if (hipStreamSynchronize(st) != hipSuccess) { (void)hipStreamDestroy(st); return; }   // (a stream that reports an error is not kept)
  StreamPool& P = StreamPool::get(); std::lock_guard<std::mutex> g(P.mu);
  if (P.free.size() < 64) P.free.push_back({dev, flags, prio, st}); else (void)hipStreamDestroy(st);
}

// Host -> device copies of plan data (read headers, work records, focus bytes, tiles) go through page-locked staging owned by the library.
// Handed a large PAGEABLE buffer, the HIP runtime pins the caller's pages in place (a userptr mapping) for the DMA; when that memory is
// later freed or remapped — the planner's vectors are temporaries — the kernel driver evicts every queue of the process and restores
// them after a delay: measured as 10-25 ms between the first launch of the next pass and its first event (round 5; with
// GPU_PINNED_MIN_XFER_SIZE raised so that the runtime never pins in place the wait is 0.00 ms).  So the library never hands the
// runtime a large pageable buffer: two 8 MiB halves per calling thread, filled on the host pool while the other half is in flight.
struct H2DStage {
  static constexpr size_t kHalf = 8u << 20;
  uint8_t* p = nullptr; hipStream_t st = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; int dev = -1;
  ~H2DStage() { if (st) { for (auto& e : ev) if (e) (void)hipEventDestroy(e); pooled_stream_release(st, dev, hipStreamNonBlocking, 0);
    } if (p) (void)hipHostFree(p); }
};
inline H2DStage& copy_stage() {   // this thread's staging, its stream and events on the current device
  static thread_local H2DStage S;
  int dev = 0; hip_check(hipGetDevice(&dev), "hipGetDevice");
  if (!S.p && hipHostMalloc(reinterpret_cast<void**>(&S.p), 2 * H2DStage::kHalf, hipHostMallocPortable) != hipSuccess) { S.p = nullptr;
    throw Error(MKP_E_NOMEM, "hipHostMalloc of the copy staging failed"); }
  if (S.dev != dev) {   // the stream and events belong to a device
    if (S.st) { for (auto& e : S.ev) if (e) (void)hipEventDestroy(e); pooled_stream_release(S.st, S.dev, hipStreamNonBlocking, 0); S.st = nullptr;
      S.ev[0] = S.ev[1] = nullptr; }
    hip_check(pooled_stream_create(&S.st, dev, hipStreamNonBlocking, 0), "hipStreamCreate");
    for (auto& e : S.ev) hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    S.dev = dev;
  }
  return S;
}
inline void staged_memcpy(uint8_t* dst, const uint8_t* src, size_t n) {
  const size_t grain = 1u << 20, pieces = (n + grain - 1) / grain;
  if (pieces > 1) HostPool::get().parallel(pieces, [&](size_t i) { memcpy(dst + i * grain, src + i * grain, std::min(grain, n - i * grain)); });
  else memcpy(dst, src, n);
}
// device -> host into pageable memory (digests, read headers, readout): the same staging the other way.  Ordered after the work already
// queued on `stream`; returns when the bytes are in `dst`.
inline void d2h_copy(void* dst, const void* src_dev, size_t bytes, hipStream_t stream) {
  if (!bytes) return;
  if (bytes <= 4096) { hip_check(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, stream), "D2H");
    hip_check(hipStreamSynchronize(stream), "D2H sync"); return; }
  H2DStage& S = copy_stage();
  const uint8_t* s = static_cast<const uint8_t*>(src_dev); uint8_t* d = static_cast<uint8_t*>(dst);
  const size_t n_chunks = (bytes + H2DStage::kHalf - 1) / H2DStage::kHalf;
  for (size_t k = 0; k <= n_chunks; k++) {
    if (k < n_chunks) {   // chunk k into half k & 1 (drained of chunk k - 2 one iteration ago)
      const size_t off = k * H2DStage::kHalf, n = std::min(H2DStage::kHalf, bytes - off), h = k & 1u;
      hip_check(hipMemcpyAsync(S.p + h * H2DStage::kHalf, s + off, n, hipMemcpyDeviceToHost, stream), "D2H");
      hip_check(hipEventRecord(S.ev[h], stream), "event");
    }
    if (k >= 1) {
      const size_t j = k - 1, off = j * H2DStage::kHalf, n = std::min(H2DStage::kHalf, bytes - off), h = j & 1u;
      hip_check(hipEventSynchronize(S.ev[h]), "event sync");
      staged_memcpy(d + off, S.p + h * H2DStage::kHalf, n);
    }
  }
}
inline void h2d_copy(void* dst, const void* src, size_t bytes) {   // synchronous, like the hipMemcpy it replaces
  if (!bytes) return;
  // (small copies take the runtime's own staging buffer)
  if (bytes <= 4096) { hip_check(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), "H2D"); return; }
  H2DStage& S = copy_stage();
  const uint8_t* s = static_cast<const uint8_t*>(src); uint8_t* d = static_cast<uint8_t*>(dst);
  size_t k = 0;
  for (size_t off = 0; off < bytes; off += H2DStage::kHalf, k++) {
    const size_t n = std::min(H2DStage::kHalf, bytes - off), h = k & 1u;
    if (k >= 2) hip_check(hipEventSynchronize(S.ev[h]), "event sync");   // the copy that last used this half is done
    uint8_t* stage = S.p + h * H2DStage::kHalf;
    staged_memcpy(stage, s + off, n);
    hip_check(hipMemcpyAsync(d + off, stage, n, hipMemcpyHostToDevice, S.st), "H2D");
    hip_check(hipEventRecord(S.ev[h], S.st), "event");
  }
  hip_check(hipStreamSynchronize(S.st), "H2D sync");
}

inline double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }


}  // namespace mkp

struct mkp_ctx {
  int device = 0; int stream_prio = 0; hipStream_t stream = nullptr; std::string err; mkp_config cfg;
  mkp::CallerCfg caller; bool caller_set = false;
  mkp::Packer packer; mkp::ShardHost shard; mkp::LayoutTables tables; bool shard_open = false, resident = false;
  // (focus: a byte per position of the shard window, copied on all cores)
  mkp::PodVec<uint8_t> focus; bool has_focus = false; std::vector<mkp_motif_combo> combos;
  MkpRunParams prm; uint32_t lds_bytes = 0, n_tiles = 0; uint64_t row_cap = 0, n_slots_total = 0;
  mkp::DevBuf d_hdr, d_vals, d_cigar, d_seq, d_tagref, d_ranks, d_ml, d_layouts, d_events, d_readout, d_focus, d_combos, d_tiles, d_slotbm,
      d_tile_row_off, d_tile_row_cnt, d_tile_dst, d_misc, d_rows_src, d_rows_dst, d_prm, d_read_ids, d_chunk;
  uint32_t n_class[7] = {0, 0, 0, 0, 0, 0, 0};   // reads per decode kernel class (class_ids) launched through mkp_launch_decode
  // slot pipeline (focus runs, mkp_slots.hip): slot positions, feature stream, per-read visit records, stream tiles;
  // reads by kernel: [fused, longer than one base window | fused | cover]
  bool slot_mode = false; uint32_t n_slot_class[3] = {0, 0, 0}; uint32_t read_ids_dec_off = 0; uint64_t cov_bytes = 0;
  uint32_t n_dup_cons = 0; mkp::DevBuf d_dupcons, d_dupsegs;   // records answered from another record of their name (MkpDupCons / MkpDupSeg)
  mkp::DevBuf d_slot_pos, d_cov, d_visits, d_stiles, d_slot_ids, d_fdesc, d_work;
  // threshold sample, resident in HBM: keys = base << 30 | f32 bit pattern of an argmax probability; level-0 histogram per base
  mkp::ShardHost sample_shard; std::vector<MkpReadOut> sample_ro; uint64_t sample_n = 0;
  mkp::DevBuf d_store, d_hist0, d_hist1, d_sample_cursor, d_take, d_hist64;
  MkpRowsDev rows_src, rows_dst;
  // row columns on the host: one page-locked arena (pageable D2H of a chromosome's 120 MB of rows ran at under 5 GB/s), kept across shards
  struct RowArena { void* p = nullptr; size_t cap = 0; uint32_t* col[11] = {};
    void ensure(size_t n_rows) { const size_t need = 11 * n_rows * 4 + 64; if (need > cap) { if (p) (void)hipHostFree(p); p = nullptr; cap = 0;
        const size_t want = need + need / 4;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr;
          throw mkp::Error(MKP_E_NOMEM, "hipHostMalloc of the row arena failed"); } cap = want; }
      for (int k = 0; k < 11; k++) col[k] = (uint32_t*)p + (size_t)k * n_rows; }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; } } h_rows;
  std::vector<uint8_t> h_strand; std::vector<int32_t> h_motif; std::vector<uint32_t> h_key;
  // mkp_batch_run: the rows of all groups of the batch
  std::vector<uint32_t> batch_cols[11]; std::vector<uint8_t> batch_strand; std::vector<int32_t> batch_motif; std::vector<uint32_t> batch_key;
  // --partition-tag: tag names, the shard's key names (index = key id, 0 = "ungrouped"), the key ids present (one accumulate pass each)
  std::vector<uint32_t> iv_starts;   // the reference's interval grid inside the open shard (duplicate-name rule); empty = one interval
  std::vector<std::string> partition_tags, key_names{"ungrouped"}; std::vector<const char*> key_name_ptrs;
    std::vector<uint32_t> key_passes{0xffffffffu};
  uint64_t n_ok = 0, n_bad = 0;
  // pileup-hemi (mkp_hemi_shard_run): mode of the resident plan, partner offset, interval starts, pattern element -> mod code
  bool hemi = false, resident_hemi = false; int32_t hemi_off = 0; std::vector<uint32_t> hemi_iv; mkp::DevBuf d_hemi_iv;
  uint32_t hemi_codes[4][MKP_KMAX + 2] = {}; std::vector<uint8_t> h_hemi_base; std::vector<uint32_t> h_hemi_pat[2];
  mkp::DevBuf d_zin, d_zout, d_zblk, d_zstat; std::vector<uint8_t> h_inflated;   // mkp_bgzf_inflate
  // sampling: the contig's --include-bed mask last uploaded (host pointer + length identify it within a session)
  mkp::DevBuf d_bedmask; const uint8_t* bedmask_src = nullptr; size_t bedmask_len = 0;
  // `modkit summary` (mkp_summary): sampling rounds count calls instead of storing probabilities; device table [4][2][16] + reads_with[6] (u64)
  bool extract_mode = false;   // `extract calls`: the sampling kernels emit one record per call (forward position, classes, call_prob)
  bool summary_mode = false; mkp::DevBuf d_summary; std::vector<uint8_t> h_sum_base; std::vector<uint32_t> h_sum_code;
    std::vector<uint64_t> h_sum_pass, h_sum_fail;
  // the read-independent part of a focus shard's plan (slot bitmap, its running popcount, the slot positions, their uploads), made ahead
  // of the reads by mkp_internal_shard_preplan while the device ingest of the same shard is still running
  struct WindowPlan { bool valid = false; std::vector<uint32_t> slotbm, wpfx, slot_pos; } wplan;
  // device ingest of indexed BAMs (mkp_ingest_host.cpp): created on first use, lives with the context (staging + window buffers are reused)
  struct mkp_dev_ingest* ingest = nullptr;
  mkp_stats stats;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // 64 page-locked bytes: the pass's cursor / total / error words come back into these (a D2H into pageable memory is staged and synchronised by the
  // runtime)
  uint32_t* h_words = nullptr;
  // the parameter block as last sent to d_prm (kept alive for the async copy; re-launches skip an unchanged one)
  std::vector<uint8_t> prm_uploaded; const void* prm_uploaded_to = nullptr;
};

// threshold sampling pass on the device (decode kernel in sample mode); `recs` need not pass Packer::keep
int mkp_internal_sample(mkp_ctx* c, int32_t tid, uint32_t win_start, uint32_t win_end, const uint8_t* bedmask, const mkp_record* recs,
                        uint32_t n, bool only_mapped, std::vector<uint32_t>* n_vals);
int mkp_internal_sample_resident(mkp_ctx* c, uint32_t win_start, uint32_t win_end, const uint8_t* bedmask, const uint32_t* reads, uint32_t n,
    bool only_mapped,
                                 std::vector<uint32_t>* n_vals);   // the same pass over reads of the device-packed shard attached to the context
int mkp_internal_sample_take(mkp_ctx* c, const std::vector<uint8_t>& take);
// between mkp_shard_begin and the records: the window's slot bitmap / positions computed and uploaded ahead (plain pileup focus shards)
int mkp_internal_shard_preplan(mkp_ctx* c);
void mkp_internal_bedmask_reset(mkp_ctx* c);   // a new sampling session: host mask pointers of the last one mean nothing any more
// summary mode: zero the device table / read it back (134 u64: table[4][2][16] then reads_with[6])
int mkp_internal_summary_begin(mkp_ctx* c);
// `extract calls`: switch the sampling kernels to per-call records (and the caller tables to read-base thresholds); fetch the records of the
// last mkp_internal_sample batch: per packed record its header (event_off), readout, and the events / call_prob values
int mkp_internal_set_extract(mkp_ctx* c, bool on);
int mkp_internal_extract_fetch(mkp_ctx* c, std::vector<MkpEvent>* events, std::vector<float>* vals);
int mkp_internal_summary_get(mkp_ctx* c, uint64_t out[134], std::vector<MkpSlot>* slots);
// --device-inflate: the indexed fetch's inflate stage on the GPU (own stream and buffers; BamSource::dev_inflate = mkp_internal_device_inflate)
struct mkp_dev_inflater;
mkp_dev_inflater* mkp_internal_inflater_create(int device);
void mkp_internal_inflater_destroy(mkp_dev_inflater* d);
bool mkp_internal_device_inflate(void* user, const mkp::InflateJob& job);
