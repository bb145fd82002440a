// Host-side ingest for libmkpileup: BGZF/BAM reader (block-parallel inflate), FASTA (+.fai not
// required), BED.  This is the IO substrate the reference gets from rust-htslib / bio
// (src/pileup/mod.rs:732-743, src/fasta.rs:34,106-111, src/position_filter.rs:230-347); it feeds
// mkp_record views to the packer.  Whole-file residency (decompressed BAM kept in host RAM) is the
// round-1 design; BAI-indexed streaming is listed under "next" in DESIGN.md.
#pragma once
#include <exception>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include "mkp_inflate_host.hpp"
#include "mkp_crc32.hpp"
#include <sched.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mkpileup.h"

namespace mkp {

struct Error : std::runtime_error {
  int status;
  Error(int st, const std::string& m) : std::runtime_error(m), status(st) {}
};

// A process-wide pool of host worker threads for the ingest's short parallel sections (block inflate, record indexing).
// Spawning threads per section costs a stack mapping each — with many sections running concurrently (sampler heads, shard
// prefetch) that serialises on the address-space lock; pooled workers are created once.
class HostPool {
 public:
  static HostPool& get() { static HostPool p; return p; }
  // f(i) for every i in [0, n), on the pool's workers and the calling thread; returns when all are done.  Re-entrant and
  // callable from several threads at once.
  // CPUs this process may actually use: the scheduler affinity mask, cut by the cgroup's CPU quota (a container on a 256-thread
  // host with `cpu.max = 1600000 100000` runs 16 threads' worth; 64 pool threads there spend their time being throttled), capped at 64
  static unsigned host_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set; if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota|max> <period>"
      char q[32]; unsigned long long period = 0;
      if (fscanf(f, "%31s %llu", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) n = std::min(n,
          (unsigned)std::max(1ull, (strtoull(q, nullptr, 10) + period - 1) / period));
      fclose(f);
    } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
      long long quota = -1, period = 0; if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g);
      if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 0; fclose(h); }
      if (quota > 0 && period > 0) n = std::min(n, (unsigned)std::max(1ll, (quota + period - 1) / period));
    }
    return std::max(1u, std::min(64u, n));
  }
  unsigned size() const { return (unsigned)workers_.size() + 1u; }   // the calling thread works too
  static bool& background() { static thread_local bool b = false; return b; }
  template <class F> void parallel(size_t n, F f) {
    if (n == 0) return;
    if (n == 1 || workers_.empty()) { for (size_t i = 0; i < n; i++) f(i); return; }
    Job job; job.n = n; job.fn = [&f](size_t i) { f(i); };
    // workers claim from the front: a job of a thread that declared itself background work (the next shard's prefetch) queues behind
    // everything else, so the steps of the shard in hand are never starved by a thousand queued inflate tasks
    { std::lock_guard<std::mutex> lk(mu_); if (background()) jobs_.push_back(&job); else jobs_.push_front(&job); }
    cv_.notify_all();
    for (;;) { const size_t i = job.next.fetch_add(1); if (i >= n) break; run_one(&job, i); }
    { std::unique_lock<std::mutex> lk(mu_);
      for (auto it = jobs_.begin(); it != jobs_.end(); ++it) if (*it == &job) { jobs_.erase(it); break; }   // no new claims from now on
      done_cv_.wait(lk, [&] { return job.done.load() + job.skipped.load() >= std::min(n, job.next.load()); }); }
    // an exception thrown by f on any thread: every index is still accounted for and the job retired before it reaches the caller
    if (job.failed.load()) std::rethrow_exception(job.error);
  }
 private:
  struct Job { std::function<void(size_t)> fn; size_t n = 0; std::atomic<size_t> next{0}, done{0},
      skipped{0}; std::atomic<bool> failed{false}; std::exception_ptr error; std::mutex emu; };
  static void run_one(Job* job, size_t i) {   // never lets an exception escape: the first one is kept for the caller, the index counts as done
    try { if (!job->failed.load()) job->fn(i); }
    catch (...) { std::lock_guard<std::mutex> lk(job->emu); if (!job->failed.load()) { job->error = std::current_exception(); job->failed.store(true);
      } }
    job->done.fetch_add(1);
  }
  std::vector<std::thread> workers_; std::mutex mu_; std::condition_variable cv_, done_cv_; std::deque<Job*> jobs_; bool stop_ = false;
  HostPool() {
    unsigned n = host_cpus();
    if (const char* e = getenv("MKP_POOL_THREADS")) n = std::max(1u, std::min(512u, (unsigned)strtoul(e, nullptr, 10)));   // experiments
    for (unsigned t = 1; t < n; t++) workers_.emplace_back([this] { loop(); });
  }
  ~HostPool() { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } cv_.notify_all(); for (auto& t : workers_) t.join(); }
  void loop() {
    for (;;) {
      Job* job = nullptr; size_t i = 0;
      { std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
          if (stop_) return;
          // claimed under the lock: the job cannot be retired in between
          for (Job* j : jobs_) { const size_t k = j->next.fetch_add(1); if (k < j->n) { job = j; i = k; break; } }
          if (job) break;
          cv_.wait(lk);
        } }
      run_one(job, i);
      { std::lock_guard<std::mutex> lk(mu_); }   // pairs with the waiter's predicate check
      done_cv_.notify_all();
    }
  }
};

struct BamIndexEntry {  // one alignment record inside `raw`
  uint64_t off;         // offset of the record's 32-byte core (after block_size)
  int32_t tid, pos, end;  // end = bam_endpos (pos+1 for records without reference length)
  int32_t reflen;
  uint16_t flag;
};

// A byte buffer whose pages are first touched by whoever writes them (the inflate workers), not zero-filled up front;
// anonymous mapping with transparent huge pages requested (one fault per 2 MiB instead of per 4 KiB where the host allows it)
struct ByteBuf {
  uint8_t* p = nullptr; size_t n = 0, cap = 0; bool heap = false;
  ByteBuf() = default;
  ByteBuf(ByteBuf&& o) noexcept : p(o.p), n(o.n), cap(o.cap), heap(o.heap) { o.p = nullptr; o.n = o.cap = 0; }
  ByteBuf& operator=(ByteBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; cap = o.cap; heap = o.heap; o.p = nullptr;
      o.n = o.cap = 0; } return *this; }
  ByteBuf(const ByteBuf&) = delete; ByteBuf& operator=(const ByteBuf&) = delete;
  ~ByteBuf() { release(); }
  // Large mappings are recycled through a process-wide spare list: a shard's inflated windows are ~1 GB, and unmapping them after the
  // pack (~45 ms per shard, the address-space lock held against every other thread's faults) plus faulting fresh ones for the next
  // shard cost more than the copy itself.  At most spare_limit() bytes stay parked (recycled mappings are not zeroed: callers write what they read,
  // including the 8 spare bytes behind a window); trim_spares() (BamSource destructor) returns them.
  struct Spares { std::mutex mu; std::vector<std::pair<uint8_t*, size_t>> free; size_t bytes = 0; };
  static Spares& spares() { static Spares s; return s; }
  // at most 6 GiB parked, and never more than a quarter of what this process may use (the cgroup's memory limit where there is one):
  // parked mappings are resident pages nobody is using
  static size_t spare_limit() {
    static const size_t lim = []() { size_t v = (size_t)6 << 30;
      if (FILE* f = fopen("/sys/fs/cgroup/memory.max", "r")) { char q[64]; if (fscanf(f, "%63s", q) == 1 && strcmp(q, "max") != 0) {
          const unsigned long long m = strtoull(q, nullptr, 10); if (m) v = std::min<size_t>(v, (size_t)(m / 4)); } fclose(f); }
      else if (FILE* g = fopen("/sys/fs/cgroup/memory/memory.limit_in_bytes", "r")) { unsigned long long m = 0;
        if (fscanf(g, "%llu", &m) == 1 && m && m < (1ull << 60)) v = std::min<size_t>(v, (size_t)(m / 4));
        fclose(g); }
      return v; }();
    return lim;
  }
  static void trim_spares() { Spares& s = spares(); std::lock_guard<std::mutex> g(s.mu); for (auto& f : s.free) munmap(f.first, f.second);
    s.free.clear(); s.bytes = 0; }
  void release() {
    if (p) {
      if (heap) free(p);
      else {
        Spares& s = spares(); bool parked = false;
        { std::lock_guard<std::mutex> g(s.mu); if (s.bytes + cap <= spare_limit() && s.free.size() < 64) { s.free.push_back({p, cap}); s.bytes += cap;
            parked = true; } }
        if (!parked) munmap(p, cap);
      }
    }
    p = nullptr; n = cap = 0; heap = false;
  }
  // small buffers come from the heap: many threads mapping, faulting and unmapping small regions serialise on the address-space lock
  void alloc(size_t bytes) {
    release();
    if (bytes < (32u << 20)) { p = (uint8_t*)malloc(std::max<size_t>(bytes, 1));
      if (!p) throw Error(MKP_E_NOMEM, "out of host memory for the decompressed BAM");
      n = cap = bytes; heap = true; return; }
    // sizes in 32 MiB steps: the windows of one file inflate to similar sizes and then land in the same step
    const size_t want = (std::max<size_t>(bytes, 1) + (32u << 20) - 1) & ~(size_t)((32u << 20) - 1);
    // the smallest parked mapping that holds it without wasting more than its size again
    { Spares& s = spares(); std::lock_guard<std::mutex> g(s.mu);
      size_t best = SIZE_MAX;
      for (size_t i = 0; i < s.free.size(); i++) if (s.free[i].second >= want && s.free[i].second <= 2 * want
          && (best == SIZE_MAX || s.free[i].second < s.free[best].second)) best = i;
      if (best != SIZE_MAX) { p = s.free[best].first; cap = s.free[best].second; n = bytes; s.bytes -= cap;
        s.free.erase(s.free.begin() + (ptrdiff_t)best); return; }
      // nothing parked fits: the parked ones belong to a shape of work that is over; give them back so that the peak follows the shard in hand
      for (auto& f : s.free) munmap(f.first, f.second);
      s.free.clear(); s.bytes = 0; }
    cap = want;
    void* m = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) { cap = 0; throw Error(MKP_E_NOMEM, "out of host memory for the decompressed BAM"); }
    madvise(m, cap, MADV_HUGEPAGE);
    p = (uint8_t*)m; n = bytes;
  }
  size_t size() const { return n; }
  const uint8_t* data() const { return p; }
  uint8_t* data() { return p; }
  const uint8_t& operator[](size_t i) const { return p[i]; }
  uint8_t& operator[](size_t i) { return p[i]; }
};

// Read-only view of a whole file (mmap: the page cache is read in place)
struct FileMap {
  const uint8_t* p = nullptr; size_t n = 0;
  explicit FileMap(const std::string& path) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error(MKP_E_IO, "cannot open " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw Error(MKP_E_IO, "cannot stat " + path); }
    n = (size_t)st.st_size;
    if (n) {
      void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { close(fd); throw Error(MKP_E_IO, "cannot map " + path); }
      madvise(m, n, MADV_SEQUENTIAL);
      p = (const uint8_t*)m;
    }
    close(fd);
  }
  ~FileMap() { if (p) munmap((void*)p, n); }
  FileMap(const FileMap&) = delete; FileMap& operator=(const FileMap&) = delete;
  size_t size() const { return n; }
  const uint8_t& operator[](size_t i) const { return p[i]; }
};

struct BamData {
  std::vector<std::string> ref_names;
  std::vector<uint32_t> ref_lens;
  ByteBuf raw;  // decompressed stream
  std::vector<BamIndexEntry> recs;
  std::vector<size_t> tid_first;  // first record index per tid (+ sentinel), records are coordinate sorted

  int tid_of(const std::string& n) const { for (size_t i = 0; i < ref_names.size(); i++) if (ref_names[i] == n) return (int)i; return -1; }

  mkp_record view(const BamIndexEntry& e) const {
    const uint8_t* c = &raw[e.off];
    mkp_record r;
    memcpy(&r.tid, c, 4); memcpy(&r.pos, c + 4, 4);
    r.l_qname = c[8];
    uint16_t nc; memcpy(&nc, c + 12, 2); r.n_cigar = nc;
    memcpy(&r.flag, c + 14, 2);
    memcpy(&r.l_qseq, c + 16, 4);
    uint32_t bs; memcpy(&bs, c - 4, 4);
    r.l_data = (int32_t)bs - 32;
    r.data = c + 32;
    return r;
  }
};

// one BGZF block's deflate payload -> dst[0, dlen).  The payload is followed by the block's CRC32 + ISIZE (8 readable bytes).
// The library's own decoder (mkp_inflate_host.hpp) first; zlib when it declines, which is also where the error comes from.
// Either way the inflated bytes must carry the CRC32 the block's trailer names (as htslib's bgzf reader demands): a decoder bug or a
// flipped bit that still yields ISIZE bytes is an error, not data.
static inline void check_block_crc(const uint8_t* src, size_t clen, const uint8_t* dst, size_t dlen) {
  uint32_t want; memcpy(&want, src + clen, 4);
  if (crc32_of(dst, dlen) != want) throw Error(MKP_E_IO, "corrupt BGZF block (CRC32 mismatch)");
}
static inline void inflate_block(const uint8_t* src, size_t clen, uint8_t* dst, size_t dlen) {
  if (hostinf::inflate(src, clen, dst, dlen)) { check_block_crc(src, clen, dst, dlen); return; }   // (blocks the table decoder declines go to zlib)
  z_stream zs; memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, -15) != Z_OK) throw Error(MKP_E_IO, "zlib init failed");
  zs.next_in = const_cast<Bytef*>(src); zs.avail_in = (uInt)clen; zs.next_out = dst; zs.avail_out = (uInt)dlen;
  int rc = inflate(&zs, Z_FINISH);
  inflateEnd(&zs);
  if (rc != Z_STREAM_END || zs.avail_out != 0) throw Error(MKP_E_IO, "corrupt BGZF block");
  check_block_crc(src, clen, dst, dlen);
}

// require_sorted = false: file order is all the caller needs (`extract calls` walks records one by one; fetches are not valid then)
static inline BamData load_bam(const std::string& path, unsigned threads = 0, bool require_sorted = true) {
  FileMap comp(path);
  struct Blk { size_t coff, clen, doff, dlen; };
  std::vector<Blk> blks; size_t o = 0, dtotal = 0;
  while (o + 18 <= comp.size()) {
    if (comp[o] != 31 || comp[o + 1] != 139) throw Error(MKP_E_IO, "not BGZF: " + path);
    uint16_t xlen; memcpy(&xlen, &comp[o + 10], 2);
    size_t x = o + 12, xe = x + xlen; uint32_t bsize = 0; bool found = false;
    if (xe > comp.size()) throw Error(MKP_E_IO, "bad BGZF block in " + path);
    while (x + 4 <= xe) { uint16_t sl; memcpy(&sl, &comp[x + 2], 2); if (comp[x] == 'B' && comp[x + 1] == 'C' && sl == 2 && x + 6 <= xe) { uint16_t b;
        memcpy(&b, &comp[x + 4], 2); bsize = (uint32_t)b + 1; found = true; } x += 4 + (size_t)sl; }
    if (!found || o + bsize > comp.size() || bsize < (uint32_t)xlen + 20u) throw Error(MKP_E_IO, "bad BGZF block in " + path);
    uint32_t isize; memcpy(&isize, &comp[o + bsize - 4], 4);
    blks.push_back({o + 12 + xlen, bsize - xlen - 20, dtotal, isize});
    dtotal += isize; o += bsize;
  }
  BamData bd; bd.raw.alloc(dtotal);
  if (!threads) threads = HostPool::host_cpus();
  std::atomic<size_t> next{0}; std::atomic<bool> bad{false};
  auto work = [&]() { for (;;) { size_t i = next++; if (i >= blks.size()) break; if (!blks[i].dlen) continue; try {
        inflate_block(&comp[blks[i].coff], blks[i].clen, &bd.raw[blks[i].doff], blks[i].dlen); } catch (...) { bad = true; } } };
  if (threads <= 1 || blks.size() < 4) work(); else { std::vector<std::thread> th; for (unsigned t = 0; t < threads; t++) th.emplace_back(work);
    for (auto& t : th) t.join();
    }
  if (bad) throw Error(MKP_E_IO, "corrupt BGZF data in " + path);
  const ByteBuf& d = bd.raw; o = 0;
  auto need = [&](size_t n) { if (o + n > d.size()) throw Error(MKP_E_IO, "truncated BAM " + path); };
  need(12);
  if (memcmp(&d[0], "BAM\1", 4) != 0) throw Error(MKP_E_IO, "not a BAM file: " + path);
  int32_t l_text; memcpy(&l_text, &d[4], 4); o = 8;
  if (l_text < 0) throw Error(MKP_E_IO, "corrupt BAM header: negative text length");
  need((size_t)l_text + 4); o += (size_t)l_text;
  int32_t n_ref; memcpy(&n_ref, &d[o], 4); o += 4;
  if (n_ref < 0) throw Error(MKP_E_IO, "corrupt BAM header: negative reference count");
  for (int i = 0; i < n_ref; i++) {
    need(4); int32_t ln; memcpy(&ln, &d[o], 4); o += 4;
    if (ln <= 0) throw Error(MKP_E_IO, "corrupt BAM header: reference name length");
    need((size_t)ln + 4);
    bd.ref_names.push_back(std::string((const char*)&d[o], ln > 0 ? (size_t)ln - 1 : 0)); o += (size_t)ln;
    uint32_t lr; memcpy(&lr, &d[o], 4); o += 4; bd.ref_lens.push_back(lr);
  }
  while (o + 4 <= d.size()) {  // record boundaries: one hop per record
    int32_t bs; memcpy(&bs, &d[o], 4); o += 4; need((size_t)bs);
    if (bs < 32) throw Error(MKP_E_IO, "corrupt BAM record");
    BamIndexEntry e; e.off = o;
    memcpy(&e.tid, &d[o], 4); memcpy(&e.pos, &d[o + 4], 4);
    uint8_t lq = d[o + 8]; uint16_t nc; memcpy(&nc, &d[o + 12], 2); memcpy(&e.flag, &d[o + 14], 2);
    int32_t lseq; memcpy(&lseq, &d[o + 16], 4);
    if (lseq < 0 || (uint64_t)32 + lq + 4ull * nc + ((uint64_t)lseq + 1) / 2 + (uint64_t)lseq > (uint64_t)bs) throw Error(MKP_E_IO,
        "corrupt BAM record");
    if (e.tid < -1 || e.tid >= n_ref || e.pos < -1 || e.pos >= 0x7ffffff0) throw Error(MKP_E_IO,
        "corrupt BAM record: reference id or position out of range");
    e.reflen = 0; e.end = e.pos + 1;
    // fetches assume a coordinate-sorted file (reference ids ascending, unplaced records last, positions ascending inside a reference)
    // -1 sorts last as unsigned
    if (require_sorted && !bd.recs.empty()) { const BamIndexEntry& q = bd.recs.back(); const uint32_t ta = (uint32_t)q.tid, tb = (uint32_t)e.tid;
      if (tb < ta || (tb == ta && e.tid >= 0 && e.pos < q.pos)) throw Error(MKP_E_INVALID, "the BAM is not coordinate sorted: " + path); }
    bd.recs.push_back(e); o += (size_t)bs;
  }
  {  // reference spans (bam_endpos): CIGAR walks, all cores
    const size_t n = bd.recs.size(); std::atomic<bool> span_bad{false};
    auto span = [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; i++) {
        BamIndexEntry& e = bd.recs[i];
        const uint8_t* c = &d[e.off]; uint16_t nc; memcpy(&nc, c + 12, 2);
        const uint8_t* cg = c + 32 + c[8]; int64_t rl = 0;
        for (uint16_t k = 0; k < nc; k++) { uint32_t w; memcpy(&w, cg + 4 * k, 4); uint32_t op = w & 15;
          if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += w >> 4;
          }
        if ((int64_t)e.pos + rl > 0x7ffffff0ll) { span_bad = true; rl = 0; }   // alignment runs past 2^31: corrupt record
        e.reflen = (int32_t)rl; e.end = e.pos + (rl > 0 ? (int32_t)rl : 1);
      }
    };
    const unsigned nt = n >= 4096 ? threads : 1u;
    if (nt <= 1) span(0, n);
    else { std::vector<std::thread> th; for (unsigned t = 0; t < nt; t++) th.emplace_back(span, n * t / nt, n * (t + 1) / nt);
      for (auto& t : th) t.join();
      }
    if (span_bad) throw Error(MKP_E_IO, "corrupt BAM record: alignment runs past 2^31");
  }
  bd.tid_first.assign(bd.ref_names.size() + 1, bd.recs.size());
  for (size_t i = bd.recs.size(); i-- > 0;) { int t = bd.recs[i].tid; if (t >= 0 && (size_t)t < bd.ref_names.size()) bd.tid_first[(size_t)t] = i; }
  for (size_t t = bd.ref_names.size(); t-- > 0;) if (bd.tid_first[t] == bd.recs.size()
      && t + 1 <= bd.ref_names.size()) bd.tid_first[t] = bd.tid_first[t + 1];
  return bd;
}

// ------------------------------------------------------------------------------------ indexed, bounded ingest
// What the reference gets from htslib's IndexedReader::fetch (src/pileup/mod.rs:732-743): only the BGZF blocks the BAI (SAM spec
// 5.2) lists for a region are read (pread: no whole-file mapping) and inflated, so memory follows the shard, not the file, and a
// rank of a multi-GPU run touches only its own part of the file.  Without a .bai next to the BAM the whole file is loaded once
// (load_bam) and fetches are views into it.
struct BamBatch {               // the records of one fetch, in file order
  const uint8_t* base = nullptr;   // record offsets are relative to this (resident file), or absolute addresses when null
  std::vector<ByteBuf> owned;      // the inflated ingest windows (indexed source); empty when `base` points into a resident file
  std::vector<BamIndexEntry> recs;
  const uint8_t* at(const BamIndexEntry& e) const { return reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(base) + (uintptr_t)e.off); }
  void clear() { recs.clear(); for (auto& b : owned) b.release(); owned.clear(); base = nullptr; }
  mkp_record view(const BamIndexEntry& e) const {
    const uint8_t* c = at(e);
    mkp_record r;
    memcpy(&r.tid, c, 4); memcpy(&r.pos, c + 4, 4);
    r.l_qname = c[8];
    uint16_t nc; memcpy(&nc, c + 12, 2); r.n_cigar = nc;
    memcpy(&r.flag, c + 14, 2);
    memcpy(&r.l_qseq, c + 16, 4);
    uint32_t bs; memcpy(&bs, c - 4, 4);
    r.l_data = (int32_t)bs - 32;
    r.data = c + 32;
    return r;
  }
  std::string qname(const BamIndexEntry& e) const { const uint8_t* c = at(e); return std::string((const char*)c + 32, c[8] ? (size_t)c[8] - 1 : 0); }
  uint32_t l_seq(const BamIndexEntry& e) const { uint32_t lq; memcpy(&lq, at(e) + 16, 4); return lq; }
};

// one record header at d[o] (after its block_size) -> index entry; false when the record is malformed
static inline bool index_record(const uint8_t* d, size_t o, int32_t bs, int32_t n_ref, BamIndexEntry* e) {
  if (bs < 32) return false;
  e->off = o; memcpy(&e->tid, d + o, 4); memcpy(&e->pos, d + o + 4, 4);
  const uint8_t lq = d[o + 8]; uint16_t nc; memcpy(&nc, d + o + 12, 2); memcpy(&e->flag, d + o + 14, 2);
  int32_t lseq; memcpy(&lseq, d + o + 16, 4);
  if (lseq < 0 || (uint64_t)32 + lq + 4ull * nc + ((uint64_t)lseq + 1) / 2 + (uint64_t)lseq > (uint64_t)bs) return false;
  if (e->tid < -1 || e->tid >= n_ref || e->pos < -1 || e->pos >= 0x7ffffff0) return false;
  const uint8_t* cg = d + o + 32 + lq; int64_t rl = 0;
  for (uint16_t k = 0; k < nc; k++) { uint32_t w; memcpy(&w, cg + 4 * k, 4); const uint32_t op = w & 15;
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += w >> 4;
    }
  if ((int64_t)e->pos + rl > 0x7ffffff0ll) return false;
  e->reflen = (int32_t)rl; e->end = e->pos + (rl > 0 ? (int32_t)rl : 1);
  return true;
}

struct BaiIndex {
  struct Chunk { uint64_t beg, end; };
  struct Ref { std::map<uint32_t, std::vector<Chunk>> bins; std::vector<uint64_t> lin; uint64_t mapped = 0, unmapped = 0, off_beg = 0,
      off_end = 0; bool has_counts = false; };
  std::vector<Ref> refs; uint64_t no_coor = 0; bool has_no_coor = false;
  static bool load(const std::string& path, BaiIndex* out) {
    FILE* f = fopen(path.c_str(), "rb"); if (!f) return false;
    std::vector<uint8_t> b; { fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); b.resize((size_t)std::max(n, 0l));
      if (n > 0 && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f);
        return false; } }
    fclose(f);
    size_t o = 0; auto need = [&](size_t n) { if (o + n > b.size()) throw Error(MKP_E_IO, "truncated BAM index " + path); };
    auto u32 = [&]() { need(4); uint32_t v; memcpy(&v, &b[o], 4); o += 4; return v; }; auto u64 = [&]() { need(8); uint64_t v; memcpy(&v, &b[o], 8);
      o += 8; return v; };
    need(4); if (memcmp(b.data(), "BAI\1", 4) != 0) throw Error(MKP_E_IO, "not a BAI index: " + path); o = 4;
    const uint32_t n_ref = u32(); if (n_ref > (1u << 24)) throw Error(MKP_E_IO, "corrupt BAM index " + path);
    out->refs.assign(n_ref, Ref());
    for (uint32_t r = 0; r < n_ref; r++) {
      Ref& R = out->refs[r]; const uint32_t n_bin = u32();
      for (uint32_t k = 0; k < n_bin; k++) {
        const uint32_t bin = u32(), n_chunk = u32(); need((size_t)n_chunk * 16);
        // htslib's metadata pseudo-bin
        if (bin == 37450 && n_chunk == 2) { R.off_beg = u64(); R.off_end = u64(); R.mapped = u64(); R.unmapped = u64(); R.has_counts = true; continue;
          }
        std::vector<Chunk>& v = R.bins[bin]; for (uint32_t c = 0; c < n_chunk; c++) { Chunk ch; ch.beg = u64(); ch.end = u64(); v.push_back(ch); }
      }
      const uint32_t n_intv = u32(); need((size_t)n_intv * 8); R.lin.resize(n_intv); for (uint32_t i = 0; i < n_intv; i++) R.lin[i] = u64();
    }
    if (o + 8 <= b.size()) { out->no_coor = u64(); out->has_no_coor = true; }
    return true;
  }
  static void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t>* bins) {  // SAM spec 5.3
    bins->clear(); --end; bins->push_back(0);
    for (int64_t k = 1 + (beg >> 26); k <= 1 + (end >> 26); k++) bins->push_back((uint32_t)k);
    for (int64_t k = 9 + (beg >> 23); k <= 9 + (end >> 23); k++) bins->push_back((uint32_t)k);
    for (int64_t k = 73 + (beg >> 20); k <= 73 + (end >> 20); k++) bins->push_back((uint32_t)k);
    for (int64_t k = 585 + (beg >> 17); k <= 585 + (end >> 17); k++) bins->push_back((uint32_t)k);
    for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); k++) bins->push_back((uint32_t)k);
  }
  // merged virtual-offset ranges that can hold records of `tid` overlapping [beg, end)
  std::vector<Chunk> query(uint32_t tid, int64_t beg, int64_t end) const {
    std::vector<Chunk> out; if (tid >= refs.size() || end <= beg) return out;
    const Ref& R = refs[tid]; std::vector<uint32_t> bins; reg2bins(beg, end, &bins);
    uint64_t min_off = 0; { const size_t w = (size_t)(beg >> 14); if (!R.lin.empty()) min_off = R.lin[std::min(w, R.lin.size() - 1)]; }
    // upper bound: a record that lies wholly inside a 16 kb window past `end` starts behind every record the region can hold (the file is
    // coordinate sorted), so the first chunk of that window's own bin ends the search — what htslib gets by stopping at the first
    // record with pos >= end, known here before anything is read (the device ingest uploads whole ranges; a sparse BED asks for hundreds)
    uint64_t max_off = UINT64_MAX;   // (the smallest chunk start of that bin: the BAI format does not promise a bin's chunks sorted)
    for (int64_t w = ((end - 1) >> 14) + 1, tries = 0; tries < 256 && w < (1 << 15); w++, tries++) { auto it = R.bins.find((uint32_t)(4681 + w));
      if (it != R.bins.end() && !it->second.empty()) { max_off = UINT64_MAX;
        for (auto& ck : it->second) max_off = std::min<uint64_t>(max_off, ck.beg);
        break; } }
    for (uint32_t b : bins) { auto it = R.bins.find(b); if (it == R.bins.end()) continue;
      for (auto c : it->second) if (c.end > min_off && c.beg < max_off) { c.end = std::min(c.end, max_off);
        out.push_back(c); } }
    std::sort(out.begin(), out.end(), [](const Chunk& a, const Chunk& b) { return a.beg < b.beg; });
    // chunks are record-granular and those of neighbouring bins interleave in the file: ranges that overlap, touch, or lie within
    // one block (64 KiB compressed) of each other are read as one — the records in between belong to other bins of the same
    // region or are dropped by the position test, and reading through the gap beats re-reading the same blocks chunk by chunk
    std::vector<Chunk> m; for (auto& c : out) {
      if (!m.empty() && (c.beg >> 16) <= (m.back().end >> 16) + (1u << 16)) m.back().end = std::max(m.back().end, c.end);
      else m.push_back(c);
      }
    return m;
  }
  // the same for several disjoint ascending windows of one reference at once (a shard made of BED spans): one merged list
  std::vector<Chunk> query_parts(uint32_t tid, const std::vector<std::pair<int64_t, int64_t>>& parts) const {
    if (parts.size() == 1) return query(tid, parts[0].first, parts[0].second);
    std::vector<Chunk> all; for (auto& pr : parts) { const std::vector<Chunk> q = query(tid, pr.first, pr.second);
      all.insert(all.end(), q.begin(), q.end()); }
    std::sort(all.begin(), all.end(), [](const Chunk& a, const Chunk& b) { return a.beg < b.beg; });
    std::vector<Chunk> m; for (auto& c : all) {
      if (!m.empty() && (c.beg >> 16) <= (m.back().end >> 16) + (1u << 16)) m.back().end = std::max(m.back().end, c.end);
      else m.push_back(c);
      }
    return m;
  }
};

// windows of a multi-part fetch: ascending, disjoint; a record belongs to the fetch when its [pos, end) meets one of them
using FetchParts = std::vector<std::pair<int64_t, int64_t>>;
static inline bool overlaps_parts(const FetchParts& parts, int64_t pos, int64_t end) {
  size_t lo = 0, hi = parts.size();   // first part that ends behind pos
  while (lo < hi) { const size_t mid = (lo + hi) / 2; if (parts[mid].second > pos) hi = mid; else lo = mid + 1; }
  return lo < parts.size() && parts[lo].first < end;
}

// Optional device stage of the indexed fetch (--device-inflate): one window's BGZF blocks inflated on the GPU (mkp_inflate_wave4.hip) straight
// into the window's host buffer.  false = not done (any block failed, or no device): the host decoder then runs and reports.
struct InflateBlk { unsigned long long in_off, out_off; uint32_t in_len, out_len; };   // == MkpBgzfBlock
struct InflateJob { const uint8_t* comp; size_t comp_len; const InflateBlk* blks; size_t n_blks; uint8_t* dst; size_t dtotal; };
using DeviceInflateFn = bool (*)(void* user, const InflateJob& job);

class BamSource {
 public:
  std::vector<std::string> ref_names; std::vector<uint32_t> ref_lens;
  DeviceInflateFn dev_inflate = nullptr; void* dev_inflate_user = nullptr;   // set by the driver for --device-inflate
  mutable std::atomic<uint64_t> bytes_inflated_device{0};
  mutable std::atomic<uint64_t> bytes_read{0}, bytes_inflated{0};   // compressed bytes pread / bytes inflated so far
  int tid_of(const std::string& n) const { for (size_t i = 0; i < ref_names.size(); i++) if (ref_names[i] == n) return (int)i; return -1; }
  bool indexed() const { return fd_ >= 0; }
  ~BamSource() { if (fd_ >= 0) close(fd_); ByteBuf::trim_spares(); }

  // `use_index`: read through <path>.bai when it exists; otherwise (or when there is none) load the whole file
  static std::unique_ptr<BamSource> open(const std::string& path, unsigned threads, bool use_index = true) {
    std::unique_ptr<BamSource> s(new BamSource()); s->path_ = path; s->threads_ = threads ? threads : HostPool::host_cpus();
    if (use_index && BaiIndex::load(path + ".bai", &s->bai_)) {
      s->fd_ = ::open(path.c_str(), O_RDONLY); if (s->fd_ < 0) throw Error(MKP_E_IO, "cannot open " + path);
      struct stat st; if (fstat(s->fd_, &st) != 0) throw Error(MKP_E_IO, "cannot stat " + path); s->fsize_ = (uint64_t)st.st_size;
      s->read_header();
      if (s->bai_.refs.size() != s->ref_names.size()) throw Error(MKP_E_IO, "BAM index does not match the BAM header: " + path + ".bai");
      // an index without htslib's per-reference counts (metadata pseudo-bins) cannot drive the sampling schedule: load the file instead
      bool complete = s->bai_.has_no_coor; for (auto& R : s->bai_.refs) if (!R.has_counts && !R.bins.empty()) complete = false;
      if (!complete) { close(s->fd_); s->fd_ = -1; s->ref_names.clear(); s->ref_lens.clear(); s->bai_ = BaiIndex(); }
      else { uint64_t n = s->bai_.no_coor; for (auto& R : s->bai_.refs) n += R.mapped + R.unmapped;
        if (n) s->avg_rec_bytes_ = (double)s->fsize_ / (double)n;
        }
    }
    if (s->fd_ < 0) {
      s->resident_ = load_bam(path, s->threads_); s->ref_names = s->resident_.ref_names; s->ref_lens = s->resident_.ref_lens;
      s->bytes_inflated += s->resident_.raw.size();
    }
    return s;
  }

  // records of `tid` overlapping [beg, end), file order (IndexedReader::fetch + records()); at most about `max_records` of them
  // when the caller only needs the first ones (the threshold sampler's first-N schedule)
  void fetch(uint32_t tid, uint32_t beg, uint32_t end, BamBatch* out, size_t max_records = SIZE_MAX) const {
    out->clear();
    if (tid >= ref_names.size() || end <= beg) return;
    if (!indexed()) {
      const BamData& bam = resident_; out->base = bam.raw.data();
      for (size_t i = bam.tid_first[tid]; i < bam.tid_first[tid + 1] && i < bam.recs.size(); i++) {
        const BamIndexEntry& e = bam.recs[i];
        if (e.tid != (int32_t)tid) continue;
        if ((int64_t)e.pos >= (int64_t)end) break;
        if ((int64_t)e.end > (int64_t)beg) { out->recs.push_back(e); if (out->recs.size() >= max_records) break; }
      }
      return;
    }
    read_chunks(bai_.query(tid, beg, end), (int32_t)tid, (int64_t)beg, (int64_t)end, out, max_records);
  }

  // records of `tid` overlapping any of the windows (ascending, disjoint), file order, each once: what one fetch per window would
  // return, without reading the blocks between far-apart windows and without the duplicates of a read that spans two windows
  void fetch_parts(uint32_t tid, const FetchParts& parts, BamBatch* out) const {
    out->clear();
    if (tid >= ref_names.size() || parts.empty()) return;
    if (parts.size() == 1) {
      fetch(tid, (uint32_t)std::max<int64_t>(parts[0].first, 0), (uint32_t)std::min<int64_t>(parts[0].second, 0xffffffffll), out); return; }
    const int64_t beg = parts.front().first, end = parts.back().second;
    if (!indexed()) {
      const BamData& bam = resident_; out->base = bam.raw.data();
      for (size_t i = bam.tid_first[tid]; i < bam.tid_first[tid + 1] && i < bam.recs.size(); i++) {
        const BamIndexEntry& e = bam.recs[i];
        if (e.tid != (int32_t)tid) continue;
        if ((int64_t)e.pos >= end) break;
        if ((int64_t)e.end > beg && overlaps_parts(parts, e.pos, e.end)) out->recs.push_back(e);
      }
      return;
    }
    read_chunks(bai_.query_parts(tid, parts), (int32_t)tid, beg, end, out, SIZE_MAX, &parts);
  }

  // the records without coordinates (tid < 0), at the end of a sorted file
  void fetch_unmapped(BamBatch* out) const {
    out->clear();
    if (!indexed()) { out->base = resident_.raw.data(); for (auto& e : resident_.recs) if (e.tid < 0) out->recs.push_back(e); return; }
    uint64_t from = first_record_voff_;
    for (auto& R : bai_.refs) { for (auto& kv : R.bins) for (auto& c : kv.second) from = std::max(from, c.end);
      if (R.has_counts) from = std::max(from, R.off_end);
      }
    std::vector<BaiIndex::Chunk> ch(1); ch[0].beg = from; ch[0].end = fsize_ << 16;
    read_chunks(ch, -1, 0, 0, out, SIZE_MAX);
  }

  // per-reference counts as htslib's idxstats gives them (sampling_schedule.rs:685-710)
  void counts(std::vector<uint64_t>* mapped, std::vector<uint64_t>* unmapped, uint64_t* no_coor) const {
    mapped->assign(ref_names.size(), 0); unmapped->assign(ref_names.size(), 0); *no_coor = 0;
    if (!indexed()) { for (auto& r : resident_.recs) { if (r.tid < 0) (*no_coor)++; else if (r.flag & 4) (*unmapped)[(size_t)r.tid]++;
        else (*mapped)[(size_t)r.tid]++;
      } return; }
    bool complete = bai_.has_no_coor; for (auto& R : bai_.refs) if (!R.has_counts && !R.bins.empty()) complete = false;
    if (!complete) throw Error(MKP_E_UNSUPPORTED,
        "the BAM index carries no per-reference read counts (metadata pseudo-bin): re-index the file with samtools index");
    for (size_t t = 0; t < bai_.refs.size(); t++) { (*mapped)[t] = bai_.refs[t].mapped; (*unmapped)[t] = bai_.refs[t].unmapped; }
    *no_coor = bai_.no_coor;
  }

  // Where in the (compressed) file the data of position `pos` of `tid` starts, by the index's 16 kb linear windows: monotone in
  // (tid, pos) for a sorted file, so differences measure the bytes under a reference range.  Without an index: a base-pair
  // coordinate over the concatenated references (shards are then balanced by length).
  uint64_t offset_at(uint32_t tid, uint64_t pos) const {
    if (!indexed()) { uint64_t o = 0; for (uint32_t t = 0; t < tid && t < ref_lens.size(); t++) o += ref_lens[t];
      return o + std::min<uint64_t>(pos, tid < ref_lens.size() ? ref_lens[tid] : 0); }
    if (tid >= bai_.refs.size()) return fsize_;
    // references without records take the offset of the next one that has any
    for (uint32_t t = tid; t < bai_.refs.size(); t++) {
      const BaiIndex::Ref& R = bai_.refs[t];
      if (R.lin.empty() && !R.has_counts) continue;
      if (t != tid) return R.has_counts ? R.off_beg >> 16 : (R.lin.empty() ? fsize_ : R.lin[0] >> 16);
      const size_t w = (size_t)(pos >> 14);
      if (w < R.lin.size()) { const uint64_t v = R.lin[w] >> 16; if (v || w == 0) return v ? v : (R.has_counts ? R.off_beg >> 16 : 0); }
      return R.has_counts ? R.off_end >> 16 : (R.lin.empty() ? fsize_ : R.lin.back() >> 16);
    }
    return fsize_;
  }

  // ---- device ingest (mkp_ingest_host.cpp): what the indexed fetch of [beg, end) would read and inflate, as a plan — the file
  // ranges holding the region's BGZF blocks, the block table with each block's place in the inflated window, and the record starts the
  // index knows (chunk starts + the 16 kb linear index) from which the device walks the `block_size` chains in parallel.
  struct IngestBlk { uint64_t coff; uint32_t hdr, clen, isize; uint64_t doff; };
  struct IngestRange { uint64_t file_off = 0, file_len = 0, vbeg = 0, vend = 0; size_t blk0 = 0, blk1 = 0; uint64_t raw_start = 0, raw_limit = 0;
    size_t entry0 = 0, entry1 = 0; };
  struct IngestPlan { uint32_t tid = 0; std::vector<IngestRange> ranges; std::vector<IngestBlk> blks; std::vector<uint64_t> entries; uint64_t raw_total = 0,
      comp_total = 0; };
  int fd() const { return fd_; }
  const std::string& path() const { return path_; }
  // phase 1: the file ranges (what has to go up) — the chunk list of the index and one header read at each chunk's end block
  void ingest_ranges(uint32_t tid, uint32_t beg, uint32_t end, IngestPlan* out) const {
    ingest_ranges(tid, FetchParts{{(int64_t)beg, (int64_t)end}}, out); }
  void ingest_ranges(uint32_t tid, const FetchParts& parts, IngestPlan* out) const {
    *out = IngestPlan(); out->tid = tid;
    if (!indexed() || tid >= ref_names.size() || parts.empty() || parts.back().second <= parts.front().first) return;
    for (auto& ch : bai_.query_parts(tid, parts)) {
      const uint64_t cb = ch.beg >> 16, ce = ch.end >> 16, ue = ch.end & 0xffff;
      if (cb >= fsize_ || !(cb < ce || (cb == ce && ue > 0))) continue;
      IngestRange rg; rg.file_off = cb; rg.vbeg = ch.beg; rg.vend = ch.end;
      uint64_t stop = std::min<uint64_t>(ce, fsize_);
      if (ue > 0 && ce < fsize_) {   // the end block is part of the chunk: its size from its header
        std::vector<uint8_t> hb((size_t)std::min<uint64_t>(512, fsize_ - ce)); pread_all(ce, hb.data(), hb.size()); bytes_read -= hb.size();
        Blk b; if (!block_at_header(hb, ce, &b)) throw Error(MKP_E_IO, "truncated BGZF block in " + path_);
        stop = std::min<uint64_t>(fsize_, ce + b.hdr + b.clen + 8);
      }
      rg.file_len = stop - cb; out->comp_total += rg.file_len;
      out->ranges.push_back(rg);
    }
  }
  // phase 2: block table, window layout, entry points.  The table is walked in CHAINS — runs of blocks between block starts the index
  // knows (every chunk boundary of the reference's bins, every linear-index offset) — by the device over the uploaded bytes
  // (mkp_ingest_host.cpp; round 5: 54 000 preads of the host walk below were the longest step of a whole-contig ingest) or by the host pool.
  // file offsets; stop = the next known block start inside the range, UINT64_MAX for the range's last chain
  struct IngestChain { size_t range; uint64_t start, stop; };
  void ingest_chains(const IngestPlan& plan, std::vector<IngestChain>* out) const {
    out->clear();
    const uint32_t tid = plan.tid; if (tid >= bai_.refs.size()) return;
    const BaiIndex::Ref& R = bai_.refs[tid];
    std::vector<uint64_t> known_all; known_all.reserve(R.lin.size() + 64);
    for (auto& kv : R.bins) for (auto& c : kv.second) { known_all.push_back(c.beg >> 16); known_all.push_back(c.end >> 16); }
    for (uint64_t v : R.lin) known_all.push_back(v >> 16);
    std::sort(known_all.begin(), known_all.end()); known_all.erase(std::unique(known_all.begin(), known_all.end()), known_all.end());
    for (size_t r = 0; r < plan.ranges.size(); r++) {
      const IngestRange& rg = plan.ranges[r]; const uint64_t cb = rg.file_off, range_end = cb + rg.file_len;
      const size_t first = out->size();
      out->push_back({r, cb, UINT64_MAX});
      for (auto it = std::upper_bound(known_all.begin(), known_all.end(), cb); it != known_all.end() && *it < range_end; ++it) {
        out->back().stop = *it; out->push_back({r, *it, UINT64_MAX}); }
      (void)first;
    }
  }
  // the host walk of one chain: one pread per block brings the trailer of the block in hand (ISIZE) and the header of the next
  // (a mapping of the range costs a page fault per block, all of them on one address-space lock); false = bad block / the chain misses `stop`
  bool ingest_walk_chain_host(const IngestPlan& plan, const IngestChain& ch, std::vector<IngestBlk>* blks) const {
    const IngestRange& rg = plan.ranges[ch.range];
    const uint64_t ce = rg.vend >> 16, ue = rg.vend & 0xffff, range_end = rg.file_off + rg.file_len, stop = ch.stop;
    uint64_t c = ch.start;
    try {
      std::vector<uint8_t> hb(600); Blk b; bool have = false;
      auto read_at = [&](uint64_t off, size_t n) { n = (size_t)std::min<uint64_t>(n, range_end > off ? range_end - off : 0); hb.resize(n);
        size_t got = 0;
        while (got < n) { const ssize_t r = ::pread(fd_, hb.data() + got, n - got, (off_t)(off + got));
          if (r <= 0) throw Error(MKP_E_IO, "read error on " + path_);
          got += (size_t)r; } };
      for (;;) {
        if (c >= stop || c > ce || (c == ce && ue == 0) || c + 18 > range_end) break;
        if (!have) { read_at(c, 600); if (!block_at_header(hb, c, &b)) break; }
        const uint64_t next = c + b.hdr + b.clen + 8;
        if (next > range_end) break;
        // [next - 8, next): CRC32 + ISIZE of this block; [next, ..): the next block's header
        read_at(next - 8, 8 + 600);
        if (hb.size() < 8) break;
        uint32_t isize; memcpy(&isize, hb.data() + 4, 4);
        blks->push_back({c, b.hdr, b.clen, isize, 0});
        c = next; have = false;
        if (hb.size() >= 8 + 18 && !(c >= stop || c > ce || (c == ce && ue == 0))) { Blk nb;
          if (block_at_header(hb.data() + 8, hb.size() - 8, c, &nb)) { b = nb;
            have = true; } }
      }
    } catch (...) { return false; }
    // the chain of block sizes must land on the block start the index names
    return !(stop != UINT64_MAX && c != stop && !(c > ce || (c == ce && ue == 0)));
  }
  // window layout and entry points from the chains' blocks (parts[i] = the blocks of chain i, in file order)
  void ingest_layout(IngestPlan* out, const std::vector<IngestChain>& chains, std::vector<std::vector<IngestBlk>>& parts) const {
    const BaiIndex::Ref& R = bai_.refs[out->tid];
    size_t ci = 0;
    for (size_t r = 0; r < out->ranges.size(); r++) {
      IngestRange& rg = out->ranges[r];
      const uint64_t cb = rg.file_off, ce = rg.vend >> 16, ue = rg.vend & 0xffff; const uint32_t ub = (uint32_t)(rg.vbeg & 0xffff);
      rg.blk0 = out->blks.size(); const uint64_t d0 = out->raw_total; uint64_t expect = cb;
      for (; ci < chains.size() && chains[ci].range == r; ci++) for (auto& b : parts[ci]) {
        if (b.coff != expect) throw Error(MKP_E_IO, "the BAM index does not match the file: " + path_ + ".bai");
        b.doff = out->raw_total; out->raw_total += b.isize; expect = b.coff + b.hdr + b.clen + 8; out->blks.push_back(b); }
      rg.blk1 = out->blks.size();
      if (rg.blk1 == rg.blk0 || expect != cb + rg.file_len) throw Error(MKP_E_IO, "truncated BGZF block in " + path_);
      rg.raw_start = d0 + ub; rg.raw_limit = out->raw_total;
      for (size_t k = rg.blk0; k < rg.blk1; k++) if (out->blks[k].coff == ce) rg.raw_limit = out->blks[k].doff + ue;
      // entry points: the chunk start, then every linear-index offset that falls inside the chunk
      rg.entry0 = out->entries.size(); out->entries.push_back(rg.raw_start);
      uint64_t last_v = 0;
      for (uint64_t v : R.lin) {
        if (v == last_v || v <= rg.vbeg || v >= rg.vend) continue;
        last_v = v;
        const uint64_t vc = v >> 16; size_t lo = rg.blk0, hi = rg.blk1;
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (out->blks[mid].coff < vc) lo = mid + 1; else hi = mid; }
        // not a block start of this range: the chain does without it
        if (lo >= rg.blk1 || out->blks[lo].coff != vc || (v & 0xffff) >= out->blks[lo].isize) continue;
        const uint64_t at = out->blks[lo].doff + (v & 0xffff);
        if (at > out->entries.back() && at + 4 <= rg.raw_limit) out->entries.push_back(at);
      }
      rg.entry1 = out->entries.size();
    }
  }
  void ingest_blocks(IngestPlan* out) const {   // the host form: chains on all cores
    if (out->tid >= bai_.refs.size()) return;
    std::vector<IngestChain> chains; ingest_chains(*out, &chains);
    std::vector<std::vector<IngestBlk>> parts(chains.size()); std::atomic<bool> bad{false};
    HostPool::get().parallel(chains.size(), [&](size_t i) { if (!ingest_walk_chain_host(*out, chains[i], &parts[i])) bad = true; });
    if (bad) throw Error(MKP_E_IO, "bad BGZF block in " + path_ + " (or the index does not match the file)");
    ingest_layout(out, chains, parts);
  }
  void ingest_plan(uint32_t tid, uint32_t beg, uint32_t end, IngestPlan* out) const { ingest_ranges(tid, beg, end, out); ingest_blocks(out); }

 private:
  std::string path_; unsigned threads_ = 1; int fd_ = -1; uint64_t fsize_ = 0, first_record_voff_ = 0; BaiIndex bai_; BamData resident_;
  double avg_rec_bytes_ = 0;   // compressed bytes per record over the whole file (indexed source: the index's counts)

  void pread_all(uint64_t off, uint8_t* dst, size_t n) const {
    size_t got = 0; while (got < n) { const ssize_t r = ::pread(fd_, dst + got, n - got, (off_t)(off + got));
      if (r <= 0) throw Error(MKP_E_IO, "read error on " + path_);
      got += (size_t)r; }
    bytes_read += n;
  }
  struct Blk { uint64_t coff; uint32_t hdr, clen, isize; uint64_t doff; };
  // a read-only view of file bytes [off, off + n): mapped (no copy out of the page cache), page-aligned underneath
  struct Window {
    const uint8_t* p = nullptr; size_t n = 0; void* map = nullptr; size_t map_len = 0; std::vector<uint8_t> copy;
    Window() = default; Window(const Window&) = delete; Window& operator=(const Window&) = delete;
    ~Window() { if (map) munmap(map, map_len); }
    size_t size() const { return n; }
    const uint8_t& operator[](size_t i) const { return p[i]; }
  };
  void map_window(uint64_t off, size_t n, Window* w) const {
    // small: a plain read (no mapping churn)
    if (n < (16u << 20)) { w->copy.resize(n); pread_all(off, w->copy.data(), n); w->p = w->copy.data(); w->n = n; return; }
    const uint64_t page = 4096, a0 = off & ~(page - 1);
    w->map_len = (size_t)(off - a0) + n;
    w->map = mmap(nullptr, w->map_len, PROT_READ, MAP_PRIVATE, fd_, (off_t)a0);
    if (w->map == MAP_FAILED) { w->map = nullptr; throw Error(MKP_E_IO, "cannot map " + path_); }
    madvise(w->map, w->map_len, MADV_WILLNEED);
    w->p = (const uint8_t*)w->map + (off - a0); w->n = n; bytes_read += n;
  }
  // BGZF block at compressed offset `coff` inside buf (which starts at file offset buf_off): sizes from its header / trailer
  template <class Buf> bool block_at(const Buf& buf, uint64_t buf_off, uint64_t coff, Blk* b) const {
    const size_t o = (size_t)(coff - buf_off);
    if (o + 18 > buf.size()) return false;
    if (buf[o] != 31 || buf[o + 1] != 139 || !(buf[o + 3] & 4)) throw Error(MKP_E_IO, "not BGZF: " + path_);
    uint16_t xlen; memcpy(&xlen, &buf[o + 10], 2);
    size_t x = o + 12, xe = x + xlen; uint32_t bsize = 0; bool found = false;
    if (xe > buf.size()) return false;
    while (x + 4 <= xe) { uint16_t sl; memcpy(&sl, &buf[x + 2], 2); if (buf[x] == 'B' && buf[x + 1] == 'C' && sl == 2 && x + 6 <= xe) { uint16_t v;
        memcpy(&v, &buf[x + 4], 2); bsize = (uint32_t)v + 1; found = true; } x += 4 + (size_t)sl; }
    if (!found || bsize < (uint32_t)xlen + 20u) throw Error(MKP_E_IO, "bad BGZF block in " + path_);
    if (o + bsize > buf.size()) return false;
    b->coff = coff; b->hdr = 12u + xlen; b->clen = bsize - xlen - 20u; memcpy(&b->isize, &buf[o + bsize - 4], 4); b->doff = 0;
    return true;
  }

  // sizes of the BGZF block whose header sits at the start of `hb` (file offset coff); false when the header itself is cut short
  bool block_at_header(const std::vector<uint8_t>& v, uint64_t coff, Blk* b) const { return block_at_header(v.data(), v.size(), coff, b); }
  bool block_at_header(const uint8_t* hb, size_t hn, uint64_t coff, Blk* b) const {
    if (hn < 18) return false;
    if (hb[0] != 31 || hb[1] != 139 || !(hb[3] & 4)) throw Error(MKP_E_IO, "not BGZF: " + path_);
    uint16_t xlen; memcpy(&xlen, &hb[10], 2);
    size_t x = 12, xe = x + xlen; uint32_t bsize = 0; bool found = false;
    if (xe > hn) return false;
    while (x + 4 <= xe) { uint16_t sl; memcpy(&sl, &hb[x + 2], 2); if (hb[x] == 'B' && hb[x + 1] == 'C' && sl == 2 && x + 6 <= xe) { uint16_t v;
        memcpy(&v, &hb[x + 4], 2); bsize = (uint32_t)v + 1; found = true; } x += 4 + (size_t)sl; }
    if (!found || bsize < (uint32_t)xlen + 20u) throw Error(MKP_E_IO, "bad BGZF block in " + path_);
    b->coff = coff; b->hdr = 12u + xlen; b->clen = bsize - xlen - 20u; b->isize = 0; b->doff = 0;
    return true;
  }

  void read_header() {
    // the header sits in the first blocks: inflate block by block until it is complete
    std::vector<uint8_t> d; std::vector<std::pair<uint64_t, uint32_t>> blocks; uint64_t coff = 0;
    auto ensure = [&](size_t n) {
      while (d.size() < n) {
        if (coff >= fsize_) throw Error(MKP_E_IO, "truncated BAM " + path_);
        std::vector<uint8_t> buf((size_t)std::min<uint64_t>((1u << 16) + 64, fsize_ - coff)); pread_all(coff, buf.data(), buf.size());
        Blk b; if (!block_at(buf, coff, coff, &b)) throw Error(MKP_E_IO, "truncated BGZF block in " + path_);
        const size_t at = d.size(); d.resize(at + b.isize); if (b.isize) inflate_block(&buf[b.hdr], b.clen, &d[at], b.isize);
        blocks.push_back({coff, b.isize}); coff += (uint64_t)b.hdr + b.clen + 8;
      }
    };
    ensure(12);
    if (memcmp(d.data(), "BAM\1", 4) != 0) throw Error(MKP_E_IO, "not a BAM file: " + path_);
    int32_t l_text; memcpy(&l_text, &d[4], 4); if (l_text < 0) throw Error(MKP_E_IO, "corrupt BAM header: negative text length");
    size_t o = 8 + (size_t)l_text; ensure(o + 4);
    int32_t n_ref; memcpy(&n_ref, &d[o], 4); o += 4; if (n_ref < 0) throw Error(MKP_E_IO, "corrupt BAM header: negative reference count");
    for (int32_t i = 0; i < n_ref; i++) {
      ensure(o + 4); int32_t ln; memcpy(&ln, &d[o], 4); if (ln <= 0) throw Error(MKP_E_IO, "corrupt BAM header: reference name length");
      ensure(o + 4 + (size_t)ln + 4);
      ref_names.push_back(std::string((const char*)&d[o + 4], (size_t)ln - 1)); uint32_t lr; memcpy(&lr, &d[o + 4 + (size_t)ln], 4);
        ref_lens.push_back(lr); o += 8 + (size_t)ln;
    }
    // virtual offset of the first record: `o` bytes into the inflated stream
    uint64_t acc = 0; first_record_voff_ = coff << 16;
    for (auto& b : blocks) { if (o < acc + b.second) { first_record_voff_ = (b.first << 16) | (uint64_t)(o - acc); break; } acc += b.second; }
  }

  // inflate the blocks under the (merged, ascending) virtual-offset ranges and index their records that pass the region test
  void read_chunks(const std::vector<BaiIndex::Chunk>& chunks, int32_t tid, int64_t beg, int64_t end, BamBatch* out, size_t max_records,
      const FetchParts* parts = nullptr) const {
    const int32_t n_ref = (int32_t)ref_names.size();
    // groups of chunks are processed until enough records are in hand; each group: pread, block walk, parallel inflate, record scan
    bool stop = false;
    for (size_t ci = 0; ci < chunks.size() && !stop; ci++) {
      uint64_t cb = chunks[ci].beg >> 16; const uint64_t ce = chunks[ci].end >> 16, ue = chunks[ci].end & 0xffff;
        uint32_t ub = (uint32_t)(chunks[ci].beg & 0xffff);
      // a bounded window of compressed bytes at a time (a chunk may be the whole contig): 64 MiB, or — when the caller wants only
      // the first records of the region — 4 MiB growing to that
      // (heads: sized from the file's mean compressed record — index counts over file size — plus slack for the records of the first
      // chunk that end before the region; a fixed 2 MiB start was short for 10 kb reads and its doubling then inflated 2.4x what was needed)
      uint64_t window = max_records == SIZE_MAX ? (64u << 20) : avg_rec_bytes_ > 0
          ? std::min<uint64_t>(64u << 20, std::max<uint64_t>(256u << 10, (uint64_t)((double)max_records * avg_rec_bytes_ * 1.2) + (192u << 10)))
              : (2u << 20);
      while (cb < fsize_ && (cb < ce || (cb == ce && ue > 0)) && !stop) {
        const uint64_t want_end = std::min<uint64_t>(fsize_, std::min<uint64_t>(ce + (1u << 16) + 64, cb + window));
        const bool window_at_max = window >= (64u << 20) || want_end < cb + window;   // this window cannot be made larger
        window = std::min<uint64_t>(window * 2, 64u << 20);
        Window buf; map_window(cb, (size_t)(want_end - cb), &buf);
        std::vector<Blk> blks; uint64_t c = cb, dtotal = 0;
        for (;;) { Blk b; if (c > ce || (c == ce && ue == 0) || !block_at(buf, cb, c, &b)) break; b.doff = dtotal; dtotal += b.isize;
          blks.push_back(b); c += b.hdr + b.clen + 8; }
        if (blks.empty()) throw Error(MKP_E_IO, "truncated BGZF block in " + path_);
        ByteBuf d; d.alloc((size_t)dtotal + 8);
        bool on_device = false;
        if (dev_inflate && dtotal >= (32u << 20)) {   // big windows only: a sampler's 2 MiB head is not worth a round trip
          std::vector<InflateBlk> jb(blks.size());
          for (size_t i = 0; i < blks.size(); i++) jb[i] = {(unsigned long long)((blks[i].coff - cb) + blks[i].hdr), (unsigned long long)blks[i].doff,
              blks[i].clen, blks[i].isize};
          InflateJob job{&buf[0], buf.size(), jb.data(), jb.size(), d.data(), (size_t)dtotal};
          on_device = dev_inflate(dev_inflate_user, job);
          if (on_device) bytes_inflated_device += dtotal;
        }
        if (!on_device)
        { std::atomic<bool> bad{false};
          HostPool::get().parallel(blks.size(), [&](size_t i) { if (!blks[i].isize) return; try {
              inflate_block(&buf[(size_t)(blks[i].coff - cb) + blks[i].hdr], blks[i].clen, &d[(size_t)blks[i].doff], blks[i].isize); } catch (...) {
              bad = true; } });
          if (bad) throw Error(MKP_E_IO, "corrupt BGZF data in " + path_); }
        bytes_inflated += dtotal;
        // records: from `ub` in the first block to the chunk end (or the end of this window's last complete record)
        const bool last_window = c > ce || (c == ce && ue == 0) || c >= fsize_;
        uint64_t o = ub, limit = dtotal;
        if (last_window) { for (auto& b : blks) if (b.coff == ce) limit = b.doff + ue; }
        // record boundaries: one hop per record; then the per-record work (field checks, CIGAR walk for the end) on all cores
        std::vector<uint64_t> starts; uint64_t consumed = o;
        while (o + 4 <= limit) {
          int32_t bs; memcpy(&bs, &d[(size_t)o], 4);
          if (bs < 32) throw Error(MKP_E_IO, "corrupt BAM record");
          if (o + 4 + (uint64_t)bs > dtotal) {   // the record continues in the next window — unless there is none, or no window could hold it
            if (last_window) throw Error(MKP_E_IO, "truncated BAM record at the end of " + path_);
            if ((uint64_t)bs > (64u << 20)) throw Error(MKP_E_IO, "BAM record larger than the ingest window (corrupt block_size?)");
            break;
          }
          starts.push_back(o); o += 4 + (uint64_t)bs; consumed = o;
        }
        std::vector<BamIndexEntry> all(starts.size()); std::atomic<bool> rec_bad{false};
        { const size_t grain = 512, pieces = (starts.size() + grain - 1) / grain;
          HostPool::get().parallel(pieces, [&](size_t pc) { for (size_t i = pc * grain; i < std::min(starts.size(), (pc + 1) * grain); i++) {
              int32_t bs; memcpy(&bs, &d[(size_t)starts[i]], 4);
              if (!index_record(d.data(), (size_t)starts[i] + 4, bs, n_ref, &all[i])) rec_bad = true;
            } }); }
        if (rec_bad) throw Error(MKP_E_IO, "corrupt BAM record");
        std::vector<BamIndexEntry> recs;
        for (size_t i = 0; i < all.size(); i++) {
          const BamIndexEntry& e = all[i];
          if (tid >= 0) { if (e.tid != tid || (int64_t)e.pos >= end) { if (e.tid > tid || e.tid < 0 || (e.tid == tid && (int64_t)e.pos >= end)) {
                stop = true; break; } continue; } if ((int64_t)e.end <= beg) continue; if (parts && !overlaps_parts(*parts, e.pos, e.end)) continue; }
          else if (e.tid >= 0) continue;
          recs.push_back(e);
          if (recs.size() + out->recs.size() >= max_records) { stop = true; break; }
        }
        if (!recs.empty()) {   // the window stays alive with the batch; its records are addressed absolutely
          const uintptr_t wbase = reinterpret_cast<uintptr_t>(d.data());
          for (auto& e : recs) { e.off += (uint64_t)wbase; out->recs.push_back(e); }
          out->owned.push_back(std::move(d));
        }
        if (stop || last_window) break;
        // next window starts at the block holding the first unconsumed byte
        size_t bi = blks.size() - 1; while (bi > 0 && blks[bi].doff > consumed) bi--;
        // no progress: the same first record again.  A larger window may hold it; the largest one did not
        if (blks[bi].coff == cb && consumed - blks[bi].doff == ub && window_at_max) throw Error(MKP_E_IO,
            "BAM record larger than the ingest window (corrupt block_size?)");
        cb = blks[bi].coff; ub = (uint32_t)(consumed - blks[bi].doff);
      }
    }
    out->base = nullptr;
  }
};

// ------------------------------------------------------------------------------------ FASTA
// one contig's bases: a plain buffer, grown without being zeroed (a std::string would clear tens of megabytes on one thread — and take
// their page faults there — before the parallel fill overwrites them)
struct FastaSeq {
  std::unique_ptr<char[]> mem; size_t n = 0;
  FastaSeq() = default;
  FastaSeq(FastaSeq&&) = default; FastaSeq& operator=(FastaSeq&&) = default;
  // (test harnesses fill a contig from a string)
  FastaSeq& operator=(const std::string& s) { mem.reset(); n = 0; if (!s.empty()) memcpy(grow(s.size()), s.data(), s.size()); return *this; }
  size_t size() const { return n; }
  const char* data() const { return mem.get(); }
  char* grow(size_t extra) {   // room for `extra` more bytes; returns where they go (what was there is kept: a name that comes twice)
    std::unique_ptr<char[]> m(new char[n + extra + 1]);
    if (n) memcpy(m.get(), mem.get(), n);
    mem = std::move(m); char* at = mem.get() + n; n += extra; mem[n] = 0;
    return at;
  }
};
struct Fasta {
  std::map<std::string, FastaSeq> seqs;
  // One thread, line by line: the formulation the parallel loader below is tested against (tests/test_host_fasta.py).
  static std::map<std::string, std::string> load_serial(const std::string& path) {
    std::map<std::string, std::string> f; FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) throw Error(MKP_E_IO, "cannot open fasta " + path);
    std::string buf; { fseek(fp, 0, SEEK_END); const long n = ftell(fp); fseek(fp, 0, SEEK_SET); buf.resize((size_t)std::max(n, 0l));
      if (n > 0 && fread(&buf[0], 1, (size_t)n, fp) != (size_t)n) { fclose(fp);
        throw Error(MKP_E_IO, "short read on fasta " + path); } }
    fclose(fp);
    std::string* cur = nullptr; size_t o = 0; const size_t n = buf.size();
    while (o < n) {
      const char* nl = (const char*)memchr(buf.data() + o, '\n', n - o); size_t e = nl ? (size_t)(nl - buf.data()) : n, le = e;
      if (le > o && buf[le - 1] == '\r') le--;
      if (le > o) {
        if (buf[o] == '>') { std::string name(buf.data() + o + 1, le - o - 1); const size_t sp = name.find_first_of(" \t");
          if (sp != std::string::npos) name.resize(sp);
          cur = &f[name]; }
        else if (cur) cur->append(buf.data() + o, le - o);
      }
      o = e + 1;
    }
    return f;
  }
  // The whole file on all cores: positional reads of 8 MiB pieces into one buffer (mapping the file instead measured slower: a fault per
  // page of the page cache costs more than the copy), the records found by a memchr walk over the '>' at
  // line starts, every record's lines joined in two parallel passes over 1 MiB pieces (count the bytes that stay, then place them).  A
  // 3 Gb reference loads in a few hundred milliseconds instead of two seconds on one thread; same result as load_serial byte for byte
  // (CRLF line ends, blank lines, text in front of the first header, '>' inside header text, a name that comes twice).
  static Fasta load(const std::string& path) {
    Fasta f;
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error(MKP_E_IO, "cannot open fasta " + path);
    struct stat st; if (fstat(fd, &st) != 0 || st.st_size < 0) { ::close(fd); throw Error(MKP_E_IO, "cannot stat fasta " + path); }
    const size_t n = (size_t)st.st_size;
    std::unique_ptr<char[]> mem(new char[n + 1]);   // (uninitialised: first touched by the readers below, spread over the pool)
    char* buf = mem.get();
    { const size_t piece = (size_t)8 << 20, np = (n + piece - 1) / piece; std::atomic<bool> bad{false};
      HostPool::get().parallel(np, [&](size_t i) {
        size_t off = i * piece; const size_t end = std::min(n, off + piece);
        while (off < end) { const ssize_t r = ::pread(fd, buf + off, end - off, (off_t)off); if (r <= 0) { bad = true; return; } off += (size_t)r; }
      });
      ::close(fd);
      if (bad) throw Error(MKP_E_IO, "short read on fasta " + path); }
    // (a byte of a record's body stays unless it is a line feed or the carriage return in front of one — or in front of the end of the file)
    size_t o = 0;
    // text in front of the first header line belongs to no record
    while (o < n && buf[o] != '>') { const char* nl = (const char*)memchr(buf + o, '\n', n - o); o = nl ? (size_t)(nl - buf) + 1 : n; }
    while (o < n) {   // buf[o] == '>' at a line start
      const char* nl = (const char*)memchr(buf + o, '\n', n - o); const size_t e = nl ? (size_t)(nl - buf) : n; size_t le = e;
      if (le > o && buf[le - 1] == '\r') le--;
      std::string name(buf + o + 1, le - o - 1); { const size_t sp = name.find_first_of(" \t"); if (sp != std::string::npos) name.resize(sp); }
      const size_t b0 = std::min(n, e + 1);
      size_t b1 = b0;   // the body ends at the next '>' that starts a line
      for (;;) { const char* g = b1 < n ? (const char*)memchr(buf + b1, '>', n - b1) : nullptr; if (!g) { b1 = n; break; } b1 = (size_t)(g - buf);
        if (b1 == b0 || buf[b1 - 1] == '\n') break;
        b1++; }
      FastaSeq& dst = f.seqs[name];
      if (b1 > b0) {
        const size_t piece = (size_t)1 << 20, np = (b1 - b0 + piece - 1) / piece;
        std::vector<size_t> cnt(np + 1, 0);
        // one walk for both passes, whole lines at a time: memchr to the line feed, the line's bytes counted or copied in one go (a
        // piece may start or end inside a line: the carriage-return test looks at the byte behind the piece, or at the end of the file)
        auto walk = [&](size_t lo, size_t hi, char* q) {
          size_t kept = 0, k = lo;
          while (k < hi) {
            const char* l = (const char*)memchr(buf + k, '\n', hi - k); const size_t le2 = l ? (size_t)(l - buf) : hi;
            size_t keep_end = le2; if (keep_end > k && buf[keep_end - 1] == '\r' && (keep_end == n || buf[keep_end] == '\n')) keep_end--;
            if (q) memcpy(q + kept, buf + k, keep_end - k);
            kept += keep_end - k; k = le2 + 1;
          }
          return kept;
        };
        HostPool::get().parallel(np, [&](size_t i) { const size_t lo = b0 + i * piece; cnt[i + 1] = walk(lo, std::min(b1, lo + piece), nullptr); });
        for (size_t i = 0; i < np; i++) cnt[i + 1] += cnt[i];
        char* out = dst.grow(cnt[np]);   // (not cleared: the fill below touches it first, on all cores)
        HostPool::get().parallel(np, [&](size_t i) { const size_t lo = b0 + i * piece; (void)walk(lo, std::min(b1, lo + piece), out + cnt[i]); });
      }
      o = b1;
    }
    return f;
  }
  const FastaSeq* get(const std::string& name) const { auto it = seqs.find(name); return it == seqs.end() ? nullptr : &it->second; }
};

}  // namespace mkp
