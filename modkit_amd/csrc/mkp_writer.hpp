// bedMethyl output of mkp_pileup_main: rows -> text (mkp_format.hpp) on all host cores, text -> file on a writer thread.
// Host-only C++ (no device code); tests/test_format_cpu.py drives it against a one-thread formulation.
#pragma once
#include <algorithm>
#include <atomic>
#include <sys/mman.h>
#include <unistd.h>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mkpileup.h"
#include "mkp_bam.hpp"
#include "mkp_format.hpp"
#include "mkp_bgzf_out.hpp"

namespace mkp {

// bedMethyl text (writers.rs:87-156) through mkp_format.hpp: row ranges are formatted by all host cores into per-thread
// buffers; a writer thread puts the buffers of one shard on disk, in order, while the next shard is packed and run
struct RowWriter {
  struct TextBuf { std::unique_ptr<char[]> mem; size_t n = 0; std::string chrom; BgzfPiece piece; };
  std::unique_ptr<BgzfTabixSink> bz;   // --bgzf: BGZF blocks + TBI index instead of plain text
  FILE* f = nullptr; bool mixed = false; std::vector<std::string> labels; uint64_t n = 0;
  bool bz_finished = false;
  std::thread io; std::mutex mu; std::condition_variable cv; std::deque<std::vector<TextBuf>> pending;
    bool closing = false, io_failed = false, io_started = false;
  // plain text into a seekable file goes out with pwrite from all cores (decided at the first write: what the caller wrote through `f`
  // before — a header line — is flushed and the offset taken from there)
  int pos_mode = -1; uint64_t file_off = 0;
  bool positional() {
    if (pos_mode < 0) { pos_mode = 0; if (f && f != stdout && fflush(f) == 0) { const off_t o = ftello(f); if (o >= 0) { file_off = (uint64_t)o;
          pos_mode = 1; } } }
    return pos_mode == 1;
  }
  static size_t row_bound(size_t chrom_n) { return chrom_n + 96 + 14 * 11 + 32; }   // chrom + name + 14 numbers + separators
  void format_range(const std::string& chrom, const mkp_rows& r, uint64_t lo, uint64_t hi, TextBuf* out) const {
    const char sp = mixed ? ' ' : '\t';
    out->mem.reset(new char[(size_t)(hi - lo) * row_bound(chrom.size()) + 1]);   // uninitialised: only the written pages get touched
    char* p = out->mem.get();
    std::vector<uint32_t> lens; if (bz) lens.reserve((size_t)(hi - lo));
    for (uint64_t i = lo; i < hi; i++) {
      const char* p_line = p;
      char name[96]; uint32_t code = r.code_repr[i];
      int k = (code & 0x80000000u) ? snprintf(name, sizeof(name), "%u", code & 0x7fffffffu) : (name[0] = (char)code, name[1] = 0, 1);
      if (labels.size() >= 2 && r.motif_idx[i] >= 0 && (size_t)r.motif_idx[i] < labels.size()) k += snprintf(name + k, sizeof(name) - (size_t)k,
          ",%s", labels[(size_t)r.motif_idx[i]].c_str());
      p = format_row(p, chrom.data(), chrom.size(), name, (size_t)std::min<int>(k, (int)sizeof(name) - 1), sp, r.pos[i], (char)r.strand[i],
          r.n_valid[i], r.n_mod[i], r.n_canonical[i], r.n_other[i],
                     r.n_delete[i], r.n_fail[i], r.n_diff[i], r.n_nocall[i]);
      if (bz) lens.push_back((uint32_t)(p - p_line));
    }
    out->n = (size_t)(p - out->mem.get());
    if (bz) { out->chrom = chrom; out->piece.build(out->mem.get(), lens.data(), r.pos + lo, (size_t)(hi - lo)); out->mem.reset(); }
  }
  // pileup-hemi rows (PileupWriter<DuplexModBasePileup>, writers.rs:185-258): the same 18 columns with name = "<pos>,<neg>,<base>",
  // strand '.', count / canonical / other-pattern in the modified / canonical / other columns
  void format_range_hemi(const std::string& chrom, const mkp_hemi_rows& r, uint64_t lo, uint64_t hi, TextBuf* out) const {
    const char sp = mixed ? ' ' : '\t';
    out->mem.reset(new char[(size_t)(hi - lo) * row_bound(chrom.size()) + 1]);
    char* p = out->mem.get();
    std::vector<uint32_t> lens; if (bz) lens.reserve((size_t)(hi - lo));
    auto element = [](char* q, uint32_t code) { if (code == MKP_HEMI_CANONICAL) { *q++ = '-'; return q;
      } if (code & 0x80000000u) return put_u32(q, code & 0x7fffffffu); *q++ = (char)code; return q; };
    for (uint64_t i = lo; i < hi; i++) {
      const char* p_line = p;
      char name[32]; char* q = element(name, r.pattern_pos[i]); *q++ = ','; q = element(q, r.pattern_neg[i]); *q++ = ',';
        *q++ = (char)r.primary_base[i];
      p = format_row(p, chrom.data(), chrom.size(), name, (size_t)(q - name), sp, r.pos[i], '.', r.n_valid[i], r.count[i], r.n_canonical[i],
          r.n_other_pattern[i],
                     r.n_delete[i], r.n_fail[i], r.n_diff[i], r.n_nocall[i]);
      if (bz) lens.push_back((uint32_t)(p - p_line));
    }
    out->n = (size_t)(p - out->mem.get());
    if (bz) { out->chrom = chrom; out->piece.build(out->mem.get(), lens.data(), r.pos + lo, (size_t)(hi - lo)); out->mem.reset(); }
  }
  void io_loop() {
    for (;;) {
      std::vector<TextBuf> job;
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return closing || !pending.empty(); }); if (pending.empty()) return;
        job = std::move(pending.front()); }
      bool ok = true;
      if (bz) { for (auto& b : job) bz->write(b.chrom, b.piece); ok = !bz->failed; }
      else if (positional()) {
        // a regular file: every buffer's place is known, all cores write theirs at once.  (Measured on C3's 190 MB: one thread's fwrite
        // 60 ms, positional writes from the pool 39 ms — they still queue on the inode lock; growing the file and copying into a shared
        // mapping from all cores 120 ms: a page fault per 4 KiB of a fresh file costs more than the lock.)
        std::vector<uint64_t> at(job.size()); uint64_t o = file_off; for (size_t i = 0; i < job.size(); i++) { at[i] = o; o += job[i].n; }
        std::atomic<bool> bad{false}; const int fd = fileno(f);
        HostPool::get().parallel(job.size(), [&](size_t i) { size_t done = 0; while (done < job[i].n) {
            const ssize_t w = ::pwrite(fd, job[i].mem.get() + done, job[i].n - done, (off_t)(at[i] + done)); if (w <= 0) { bad = true; return;
            } done += (size_t)w; } });
        file_off = o; ok = !bad;
      }
      else for (auto& b : job) if (b.n && fwrite(b.mem.get(), 1, b.n, f) != b.n) ok = false;
      // popped after the write: `pending` bounds the text held in memory
      { std::lock_guard<std::mutex> lk(mu); pending.pop_front(); if (!ok) io_failed = true; }
      cv.notify_all();
    }
  }
  void write(const std::string& chrom, const mkp_rows& r) {
    write_rows(chrom, r.n_rows, [&](uint64_t lo, uint64_t hi, TextBuf* out) { format_range(chrom, r, lo, hi, out); });
  }
  void write_hemi(const std::string& chrom, const mkp_hemi_rows& r) {
    write_rows(chrom, r.n_rows, [&](uint64_t lo, uint64_t hi, TextBuf* out) { format_range_hemi(chrom, r, lo, hi, out); });
  }
  template <class Fmt> void write_rows(const std::string& chrom, uint64_t n_rows, Fmt fmt) {
    if (chrom.size() > 4096) throw Error(MKP_E_UNSUPPORTED, "contig name longer than 4096 bytes");
    if (n_rows == 0) return;
    const unsigned n_thr = n_rows >= 65536 ? HostPool::host_cpus() : 1u;
    // a large shard goes to the writer thread in pieces: the first piece is on its way to the file while the next is formatted
    // (C3's 2.77 M rows = 190 MB on the GPU box's 16 CPUs, tools/dbg/writer_bench.cpp: 1 piece 59 ms — 10 of formatting, then 49 of pwrite —
    //  2 pieces 35, 4 pieces 39-46, 8 pieces 37-41, 16 pieces 32-34)
    static const uint64_t env_pieces = getenv("MKP_WRITE_PIECES") ? strtoull(getenv("MKP_WRITE_PIECES"), nullptr, 10) : 0;   // (experiments)
    const uint64_t n_pieces = n_rows >= (1u << 20) ? (env_pieces ? env_pieces : 16) : 1;
    for (uint64_t pc = 0; pc < n_pieces; pc++) {
      const uint64_t p_lo = n_rows * pc / n_pieces, p_n = n_rows * (pc + 1) / n_pieces - p_lo;
      std::vector<TextBuf> bufs(n_thr);
      HostPool::get().parallel(n_thr, [&](size_t t) { fmt(p_lo + p_n * t / n_thr, p_lo + p_n * (t + 1) / n_thr, &bufs[t]); });
      {
        std::unique_lock<std::mutex> lk(mu);
        if (!io_started) { io_started = true; io = std::thread([this] { io_loop(); }); }
        cv.wait(lk, [&] { return pending.size() < 2 || io_failed; });
        if (io_failed) throw Error(MKP_E_IO, "short write on the bedMethyl output");
        pending.push_back(std::move(bufs));
      }
      cv.notify_all();
    }
    n += n_rows;
  }
  // waits for the writer thread; throws if any write came up short
  void finish() {
    if (io_started) { { std::lock_guard<std::mutex> lk(mu); closing = true; } cv.notify_all(); io.join(); io_started = false; }
    if (io_failed) throw Error(MKP_E_IO, "short write on the bedMethyl output");
    if (bz && !bz_finished) { bz_finished = true; bz->finish();
      if (bz->failed) throw Error(MKP_E_IO, "short write on the bedMethyl output or its index");
      }
    if (f && pos_mode == 1) fseeko(f, (off_t)file_off, SEEK_SET);   // the stream's own position follows what pwrite put behind it
    if (f && fflush(f) != 0) throw Error(MKP_E_IO, "short write on the bedMethyl output");
  }
  ~RowWriter() { if (io_started) { { std::lock_guard<std::mutex> lk(mu); closing = true; } cv.notify_all(); io.join(); } }
};

}  // namespace mkp
