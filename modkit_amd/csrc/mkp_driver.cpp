// `modkit pileup` as a library call: mkp_pileup_main mirrors ModBamPileup::run
// (src/pileup/subcommand.rs:382-816) — same flags, same option resolution, same bedMethyl text —
// with process_region_batch replaced by shards on the GPU.  Host work here is scheduling only:
// BAM ingest, the interval grid + focus positions, choosing which reads the threshold sampler takes
// (reads_sampler/*, sampling_schedule.rs), and formatting rows (writers.rs:87-156).
#include <sys/stat.h>

#include <cerrno>
#include <condition_variable>
#include <deque>
#include <future>
#include <memory>
#include <mutex>
#include <set>
#include <unordered_set>
#include <thread>

#include <unistd.h>

#include "mkp_ctx.hpp"
#include "mkp_ingest_host.hpp"
#include "mkp_focus.hpp"
#include "mkp_writer.hpp"
#include "mkp_rand.hpp"

using namespace mkp;

namespace {

std::string f32_display(float v);   // (Rust's `{}` of an f32; defined with the extract-calls writer below)

// --bedgraph (BedGraphWriter, writers.rs:264-381): one file per (partition key, strand, mod code[, motif]) in the output directory,
// `[<prefix>_][<key>_]<code>[_<motif label without commas>]_<positive|negative|combined>.bedgraph`, rows
// `chrom <tab> pos <tab> pos + 1 <tab> fraction_modified <tab> filtered_coverage` with the fraction as an f32 through `{}`
// (n_modified as f32 / filtered_coverage as f32, pileup/mod.rs:327-331).  A projection of the bedMethyl rows: same rows, same order per file.
struct BedGraphOut {
  std::string dir, prefix; bool groupings = false; std::vector<std::string> labels; uint64_t n = 0;
  struct File { FILE* f = nullptr; std::string buf; };
  std::map<std::string, File> files;
  void write(const std::string& chrom, const mkp_rows& r) {
    char line[160];
    for (uint64_t i = 0; i < r.n_rows; i++) {
      std::string name;
      if (!prefix.empty()) name = prefix + "_";
      if (groupings) { const uint32_t k = r.partition_key ? r.partition_key[i] : 0u;
        name += (k < r.n_partition_keys ? r.partition_key_names[k] : "not_found"); name += "_"; }
      const uint32_t code = r.code_repr[i];
      if (code & 0x80000000u) name += std::to_string(code & 0x7fffffffu); else name += (char)code;
      if (r.motif_idx[i] >= 0 && (size_t)r.motif_idx[i] < labels.size()) { name += "_";
        for (char ch : labels[(size_t)r.motif_idx[i]]) if (ch != ',') name += ch;
        }
      name += r.strand[i] == '+' ? "_positive" : r.strand[i] == '-' ? "_negative" : r.strand[i] == '.' ? "_combined" : "__unknown";
      File& fl = files[name];
      if (!fl.f) { const std::string path = dir + "/" + name + ".bedgraph"; fl.f = fopen(path.c_str(), "w");
        if (!fl.f) throw Error(MKP_E_IO, "failed to make output file " + path);
        }
      const float frac = (float)r.n_mod[i] / (float)r.n_valid[i];
      const int k = snprintf(line, sizeof(line), "\t%u\t%u\t%s\t%u\n", r.pos[i], r.pos[i] + 1, f32_display(frac).c_str(), r.n_valid[i]);
      fl.buf += chrom; fl.buf.append(line, (size_t)k);
      if (fl.buf.size() > ((size_t)1 << 20)) {
        if (fwrite(fl.buf.data(), 1, fl.buf.size(), fl.f) != fl.buf.size()) throw Error(MKP_E_IO, "write error in " + dir);
        fl.buf.clear(); }
      n++;
    }
  }
  void finish() { for (auto& kv : files) { File& fl = kv.second; if (fl.f) {
        if (!fl.buf.empty() && fwrite(fl.buf.data(), 1, fl.buf.size(), fl.f) != fl.buf.size()) throw Error(MKP_E_IO, "write error in " + dir);
        fclose(fl.f); fl.f = nullptr; } } }
  ~BedGraphOut() { for (auto& kv : files) if (kv.second.f) fclose(kv.second.f); }
};

struct Args {
  std::string in_bam, out_bed, region, sample_region, include_bed, ignore, ref_fasta, edge_filter, preset;
  uint32_t max_depth = 8000, interval_size = 100000, sampling_interval_size = 1000000;
  bool have_seed = false; uint64_t seed = 0;   // --seed
  bool serial_sampler = false;   // sample-probs / summary / extract calls on a BAM that has no index file (reads_sampler/mod.rs:129-158)
  size_t threads = 4, num_reads = 10042; bool have_frac = false; double sampling_frac = 0; bool no_filtering = false; float filter_percentile = 0.1f;
  std::vector<std::string> filter_threshold, mod_thresholds, motif_parts, partition_tags; std::string prefix;
  bool include_unmapped = false, force_allow = false, cpg = false, mask = false, combine_mods = false, combine_strands = false, invert_edge = false,
      mixed_delim = false, with_header = false;
  int device = 0; uint32_t rank = 0, world = 1;
      uint64_t shard_bp = 0,
          shard_bytes = 256ull << 20 /* BAM bytes per shard (indexed input): bounds host memory, and the next shard inflates while this one is packed and run */;
      bool no_index = false; uint32_t tile = 0; bool stats = false, plan_only = false; uint32_t rerun = 0, plan_pack_min = 1024;
  bool hemi = false;   /* `pileup-hemi` (DuplexModBamPileup, subcommand.rs:827-1514) */
  mkp_threshold_fn thr_cb = nullptr; void* thr_cb_user = nullptr;
    /* mkp_pileup_run_cb: the pass thresholds come from the caller (multi-GPU: all-reduced histograms / a broadcast) */
  uint64_t hbm_budget_mb = 0;   /* --hbm-budget-mb: device memory the shards ingested ahead may hold (0: 55 % of the device) */
  bool device_inflate = false;
    /* inflate the shards' BGZF windows on the GPU (mkp_inflate_wave4.hip) instead of the host pool, records back to the host packer */
  bool host_ingest = false, shard_bytes_set = false;
    /* --host-ingest (or MKP_HOST_INGEST=1): inflate, cut and pack the shards on the host instead of the device (mkp_ingest.hip) */
  bool bedgraph = false;
    /* --bedgraph: the output path is a directory of <code>[_<motif>]_<strand>.bedgraph files (BedGraphWriter, writers.rs:264-381) */
  bool bgzf = false;   /* write the bedMethyl as BGZF + a .tbi index (what `bgzip` + `tabix -p bed` make of the reference's output) */
};

struct RegionSpec { std::string name; uint32_t start, end; };

uint64_t peak_rss_kb() {   // VmHWM of this process (--stats)
  std::ifstream f("/proc/self/status"); std::string line;
  while (std::getline(f, line)) if (line.compare(0, 6, "VmHWM:") == 0) return strtoull(line.c_str() + 6, nullptr, 10);
  return 0;
}

RegionSpec parse_region(const std::string& raw, const BamSource& bam) {  // Region::parse_str (util.rs:463-524)
  auto bad = [&]() { return Error(MKP_E_INVALID, "invalid region, " + raw + ", should be 'chrom' or 'chrom:start-stop'"); };
  size_t c = raw.find(':');
  if (c == std::string::npos) { int tid = bam.tid_of(raw); if (tid < 0) throw Error(MKP_E_INVALID, "contig-missing");
    return {raw, 0, bam.ref_lens[(size_t)tid]}; }
  if (raw.find(':', c + 1) != std::string::npos) throw bad();
  std::string se = raw.substr(c + 1); std::vector<uint32_t> v; size_t s = 0;
  for (;;) { size_t d = se.find('-', s); std::string part = se.substr(s, d == std::string::npos ? std::string::npos : d - s), cl;
      for (char ch : part) if (ch != ',') cl += ch; if (cl.empty()) throw bad(); uint64_t x = 0; for (char ch : cl) {
        if (ch < '0' || ch > '9') throw bad();
      x = x * 10 + (uint64_t)(ch - '0'); if (x > 0xffffffffull) throw bad(); } v.push_back((uint32_t)x); if (d == std::string::npos) break; s = d + 1;
        }
  if (v.size() != 2 || v[1] <= v[0]) throw bad();
  return {raw.substr(0, c), v[0], v[1]};
}

bool parse_code(const std::string& s, uint32_t* out) {  // ModCodeRepr::parse (mod_base_code.rs:112-122)
  if (s.size() == 1) { *out = (uint32_t)(unsigned char)s[0]; return true; }
  if (s.empty()) return false;
  uint64_t v = 0; for (char c : s) { if (c < '0' || c > '9') return false; v = v * 10 + (uint64_t)(c - '0'); if (v > 0x7fffffffull) return false; }
  *out = 0x80000000u | (uint32_t)v; return true;
}

std::vector<Contig> targets(const BamSource& bam, const RegionSpec* r) {  // get_targets (util.rs:409-446)
  std::vector<Contig> out;
  for (size_t t = 0; t < bam.ref_names.size(); t++) { if (r) {
      if (bam.ref_names[t] == r->name) out.push_back({(uint32_t)t, r->start, r->end - r->start,
      bam.ref_names[t]}); } else out.push_back({(uint32_t)t, 0, bam.ref_lens[t], bam.ref_names[t]}); }
  return out;
}

struct IdxStats { std::map<int64_t, uint64_t> mapped_by_tid; uint64_t mapped = 0, unmapped = 0; };
IdxStats idxstats(const BamSource& bam, const RegionSpec* region, const BedFilter* bf) {  // IdxStats::new_from_reader (sampling_schedule.rs:649-716)
  IdxStats st; int rt = region ? bam.tid_of(region->name) : -1;
  if (region && rt < 0) throw Error(MKP_E_INVALID, "did not find target_id for region");
  std::vector<uint64_t> m, u; uint64_t nocoor = 0;
  bam.counts(&m, &u, &nocoor);   // the index's per-reference counts (htslib idxstats), or a count over the resident file
  auto keep = [&](int64_t t) { if (region) return t == rt; if (bf) return bf->has_chrom(t); return true; };
  for (size_t t = 0; t < m.size(); t++) if (keep((int64_t)t)) { st.mapped += m[t]; st.unmapped += u[t]; st.mapped_by_tid[(int64_t)t] = m[t]; }
  if (keep(-1)) st.unmapped += nocoor;
  return st;
}

struct Quota { bool all = false; size_t n = 0; };
// SamplingSchedule::from_num_reads (sampling_schedule.rs:171-273): per contig ceil(N * its share of the reads), capped by its count; while the
// sum overshoots N by more than half, contigs at or under a rising floor are dropped (the reference walks an FxHashMap there; ascending tid
// here — order unpinned)
std::map<uint32_t, Quota> quota_from_num_reads(const IdxStats& st, size_t num_reads, bool include_unmapped) {
  const uint64_t total_u = include_unmapped ? st.mapped + st.unmapped : st.mapped;
  if (total_u == 0) throw Error(MKP_E_THRESHOLD, "zero reads found in bam index");
  std::map<uint32_t, Quota> quota;
  const float total = (float)total_u; size_t sum = 0;
  for (auto& kv : st.mapped_by_tid) if (kv.second) { Quota q;
    q.n = std::min<size_t>((size_t)ceilf((float)num_reads * ((float)kv.second / total)), (size_t)kv.second);
    sum += q.n; quota[(uint32_t)kv.first] = q; }
  if (include_unmapped) sum += (size_t)ceilf((float)num_reads * ((float)st.unmapped / total));
  size_t floor = 1;
  while ((double)sum / (double)num_reads > 1.5) {
    for (auto& kv : quota) { if (kv.second.n <= floor) { sum -= kv.second.n; kv.second.n = 0; } if (sum <= num_reads) break; }
    sum = 0; for (auto& kv : quota) sum += kv.second.n; floor++;
  }
  for (auto it = quota.begin(); it != quota.end();) { if (!it->second.all && it->second.n == 0) it = quota.erase(it); else ++it; }
  return quota;
}
struct SampleTimes { double fetch_ms = 0, device_ms = 0, decide_ms = 0; uint64_t rounds = 0, reads = 0; };
SampleTimes g_sample_times;   // --stats: where the threshold estimate's time went (last run in this process)

// Default threshold estimation: the reference's deterministic "first N qualifying reads per interval" schedule
// (reads_sampler/mod.rs:30-257, sampling_schedule.rs:171-615); the per-call probabilities come from the decode kernel.
// The values are accumulated in the context's HBM-resident sample (mkp_internal_sample_take); nothing but per-read counts
// comes back to the host.  With --gpus-world W > 1 (full-data mode `-f 1.0` only) a rank walks just its own sampling intervals:
// a read is taken in the first processed interval it overlaps, so the union over ranks is the single-rank sample.
// The records of one fetch as the sampler sees them: a BamBatch of the indexed / whole-file reader, or — `res`: the shard the device ingest
// attached to the context covers the contig — indices into that shard's digest (every kept record is a candidate; names are compared
// through their two 64-bit hashes; the reads' bases and tags are in HBM already, mkp_internal_sample_resident).
struct RecSet {
  // complete: nothing was left out for a later fetch; resident: index < S->hdr.size() = a kept read, above = sampler-only read (index - hdr.size())
  std::unique_ptr<BamBatch> b; bool complete = false; const ShardHost* S = nullptr; std::vector<uint32_t> idx;
  size_t size() const { return S ? idx.size() : b->recs.size(); }
  bool truncated(size_t cap) const { return !S && !complete && b->recs.size() >= cap; }
  std::string name(size_t i) const {
    if (S) { const uint32_t k = idx[i]; const size_t n = S->hdr.size(); char key[16];
      memcpy(key, k < n ? &S->name_hash[k] : &S->so_name_hash[k - n], 8);
      memcpy(key + 8, k < n ? &S->dev_name_hash2[k] : &S->so_name_hash2[k - n], 8);
      return std::string(key, 16); }
    return b->qname(b->recs[i]);
  }
  int32_t pos(size_t i) const { if (S) { const uint32_t k = idx[i]; const size_t n = S->hdr.size();
      return k < n ? S->hdr[k].ref_start : S->so_hdr[k - n].ref_start; } return b->recs[i].pos; }
  bool candidate(size_t i, bool drop_unmapped) const {
    if (S) return true;
    const BamIndexEntry& e = b->recs[i];
    if ((e.flag & (256 | 1024 | 2048)) || b->l_seq(e) == 0) return false;
    return !(drop_unmapped && (e.flag & 4));
  }
};

// what `bam::IndexedReader::from_path` finds (htslib: <bam>.bai, <bam>.csi, or the extension replaced)
bool bam_index_file_exists(const std::string& path) {
  std::vector<std::string> c = {path + ".bai", path + ".csi"};
  if (path.size() > 4 && path.compare(path.size() - 4, 4, ".bam") == 0) { c.push_back(path.substr(0, path.size() - 4) + ".bai");
    c.push_back(path.substr(0, path.size() - 4) + ".csi"); }
  for (auto& f : c) { FILE* p = fopen(f.c_str(), "rb"); if (p) { fclose(p); return true; } }
  return false;
}

// The sampler of a BAM WITHOUT an index (reads_sampler/mod.rs:129-158; `bam::IndexedReader::from_path(..).is_ok()` decides, so an index
// ignored by `extract calls --ignore-index` still counts as one): no schedule — one pass over the file in file order, mapped and unmapped
// records alike, under RecordSampler::new_from_options (record_sampler.rs:51-61): the first --num-reads records that yield values, or with
// --sampling-frac one `gen_bool` per record the tag iterator yields (StdRng, --seed).  A region is an error there, as it is here.
// Records go to the sampling kernels contig by contig (a round has one reference window and one BED mask); the sampler's state — reads
// used, names seen — runs across the whole file.
void sample_serial(mkp_ctx* ctx, const BamSource& bam, const Args& a, const RegionSpec* region, const BedFilter* bf) {
  if (region) throw Error(MKP_E_INVALID, "cannot use region without indexed BAM");
  const bool only_mapped = !a.include_unmapped;
  const bool draws = a.have_frac && a.sampling_frac < 1.0;
  if (a.have_frac && a.sampling_frac > 1.0) throw Error(MKP_E_INVALID, "sample fraction must be <= 1");
  if (draws && !a.have_seed) throw Error(MKP_E_UNSUPPORTED,
      "--sampling-frac < 1 on a BAM without an index draws from an entropy-seeded rand::StdRng (record_sampler.rs:29-38): give --seed");
  // a draw is taken for every record whose tags parse and hold a position, whether or not a value survives the filters; the kernels report
  // surviving values only — the two coincide while nothing can remove a record's every position
  if (draws && (only_mapped || bf || !a.edge_filter.empty())) throw Error(MKP_E_UNSUPPORTED,
      "--sampling-frac < 1 on a BAM without an index together with --only-mapped / --include-bed / --edge-filter: which records consume a draw is not reproduced");
  SeededSampler rng(a.seed);
  const long limit = a.have_frac ? -1 : (long)a.num_reads;
  std::set<std::string> seen; size_t used = 0;
  std::map<uint32_t, std::vector<uint8_t>> masks;   // --include-bed: bit 0 / 1 = the position is listed for the '+' / '-' strand
  auto mask_of = [&](uint32_t tid) -> const uint8_t* {
    if (!bf) return nullptr;
    auto it = masks.find(tid); if (it != masks.end()) return it->second.data();
    std::vector<uint8_t> m(bam.ref_lens[tid], 0);
    auto mark = [&](const std::map<uint32_t, std::vector<Span>>& mp, uint8_t bit) { auto f = mp.find(tid); if (f == mp.end()) return;
        for (auto& x : f->second) for (uint64_t q = x.s; q < std::min<uint64_t>(x.e, m.size()); q++) m[q] |= bit; };
    mark(bf->pos, 1); mark(bf->neg, 2);
    return masks.emplace(tid, std::move(m)).first->second.data();
  };
  auto round = [&](const RecSet& set, int64_t tid) {
    const BamBatch& b = *set.b;
    std::vector<size_t> cand;
    for (size_t i = 0; i < set.size(); i++) if (set.candidate(i, only_mapped || !a.edge_filter.empty())) cand.push_back(i);
    const bool mapped = tid >= 0;
    const uint8_t* mask = mapped ? mask_of((uint32_t)tid) : nullptr;
    for (size_t next = 0; next < cand.size() && (limit < 0 || used < (size_t)limit);) {
      const size_t want = limit < 0 ? std::min<size_t>(cand.size() - next, 1u << 18) : std::max<size_t>(256, 2 * ((size_t)limit - used));
      const size_t hi = std::min(cand.size(), next + want);
      std::vector<mkp_record> recs; recs.reserve(hi - next); for (size_t i = next; i < hi; i++) recs.push_back(b.view(b.recs[cand[i]]));
      std::vector<uint32_t> nv;
      int rc = mkp_internal_sample(ctx, mapped ? (int32_t)tid : -1, 0, mapped ? bam.ref_lens[(size_t)tid] : 1, mask, recs.data(),
          (uint32_t)recs.size(),
          only_mapped, &nv);
      if (rc != MKP_OK) throw Error(rc, mkp_last_error(ctx));
      std::vector<uint8_t> keep(recs.size(), 0);
      for (size_t i = next; i < hi; i++) {
        if (limit >= 0 && used >= (size_t)limit) break;                               // RecordSampler::ask -> Done
        const size_t k = i - next;
        if (draws && nv[k] != 0 && !rng.keep(a.sampling_frac)) continue;               // check_sample_frac -> Skip
        std::string name = set.name(cand[i]);
        if (seen.count(name) || nv[k] == 0) continue;                                 // seen(); a record that keeps no value is not recorded
        seen.insert(name); used++; keep[k] = 1;
      }
      rc = mkp_internal_sample_take(ctx, keep);
      if (rc != MKP_OK) throw Error(rc, mkp_last_error(ctx));
      next = hi;
    }
  };
  for (size_t t = 0; t < bam.ref_names.size() && (limit < 0 || used < (size_t)limit); t++) {
    RecSet set; set.b.reset(new BamBatch()); bam.fetch((uint32_t)t, 0, std::max<uint32_t>(bam.ref_lens[t], 1u), set.b.get());
    round(set, (int64_t)t);
  }
  if (limit < 0 || used < (size_t)limit) { RecSet set; set.b.reset(new BamBatch()); bam.fetch_unmapped(set.b.get()); round(set, -1); }
}

// `resident_of` (optional): the device-packed shard holding contig `tid`, bound to the context (mkp_internal_sample_bind) — the estimate then
// samples from HBM; called whenever the schedule moves to another contig.
using ResidentOf = std::function<const ShardHost*(uint32_t tid)>;
void sample_probabilities(mkp_ctx* ctx, const BamSource& bam, const Args& a, const RegionSpec* region, const BedFilter* bf,
    const ResidentOf& resident_of = ResidentOf()) {
  const bool only_mapped = !a.include_unmapped;
  const bool sharded = a.world > 1;
  const bool resident_mode = (bool)resident_of;
  if (resident_mode && (!only_mapped || sharded)) throw Error(MKP_E_INVALID, "internal: resident sampling needs mapped-only, single-rank sampling");
  const ShardHost* res = nullptr; uint32_t res_tid = 0xffffffffu;
  std::vector<int32_t> res_pmax;   // resident shard: prefix maximum of the alignment ends (first record that can reach an interval)
  auto use_contig = [&](uint32_t tid) {   // the shard of contig `tid` becomes the one sampled from
    if (res && res_tid == tid) return;
    res = resident_of(tid); res_tid = tid;
    if (!res || (int32_t)tid != res->tid) throw Error(MKP_E_INVALID, "internal: resident sampling outside the ingested contigs");
    res_pmax.resize(res->hdr.size()); int32_t m = INT32_MIN; for (size_t i = 0; i < res->hdr.size(); i++) {
      m = std::max(m, std::max(res->hdr[i].ref_end, res->hdr[i].ref_start + 1)); res_pmax[i] = m; }
  };
  auto fetch_set = [&](uint32_t tid, uint32_t s, uint32_t e, size_t cap) {
    RecSet r;
    if (resident_mode) {
      use_contig(tid);
      r.S = res;
      const size_t first = (size_t)(std::upper_bound(res_pmax.begin(), res_pmax.end(),
          (int32_t)std::min<uint32_t>(s, 0x7fffffffu)) - res_pmax.begin());
      for (size_t i = first; i < res->hdr.size() && (int64_t)res->hdr[i].ref_start < (int64_t)e; i++) if ((int64_t)std::max(res->hdr[i].ref_end,
          res->hdr[i].ref_start + 1) > (int64_t)s) r.idx.push_back((uint32_t)i);
      // the sampler-only records of the window (QC-fail ...: few), merged in by their place in the file
      bool any_so = false;
      for (size_t k = 0; k < res->so_hdr.size(); k++) { const MkpReadHdr& h = res->so_hdr[k];
        if ((int64_t)h.ref_start < (int64_t)e && (int64_t)std::max(h.ref_end, h.ref_start + 1) > (int64_t)s) {
          r.idx.push_back((uint32_t)(res->hdr.size() + k)); any_so = true; } }
      if (any_so) { const size_t n = res->hdr.size(); auto wi = [&](uint32_t k) { return k < n ? res->dev_win_idx[k] : res->so_win_idx[k - n]; };
        std::stable_sort(r.idx.begin(), r.idx.end(), [&](uint32_t x, uint32_t y) { return wi(x) < wi(y); }); }
      return r;
    }
    r.b.reset(new BamBatch());
    if (bf) {
      // under --include-bed a read counts only through calls on BED positions (the kernel masks the rest): a read that meets no BED span
      // yields nothing, is not counted and not recorded — so only the records that can meet one are fetched (a sparse BED used to make
      // the estimate inflate every sampling interval whole), all of them: there is no head to extend afterwards.  A record of the interval
      // meets a span inside it, or reaches a span outside it — then it crosses the interval's first or last position.
      std::vector<Span> sp;
      for (auto* m : {&bf->pos, &bf->neg}) { auto it = m->find(tid); if (it == m->end()) continue;
        for (auto& x : it->second) if (x.e > s && x.s < e) sp.push_back({std::max<uint64_t>(x.s, s), std::min<uint64_t>(x.e, e)});
        }
      sp.push_back({s, (uint64_t)s + 1}); if (e > s + 1) sp.push_back({(uint64_t)e - 1, e});
      std::sort(sp.begin(), sp.end(), [](const Span& x, const Span& y) { return x.s < y.s; });
      merge_spans(sp);
      FetchParts parts; for (auto& x : sp) parts.push_back({(int64_t)x.s, (int64_t)x.e});
      if (!parts.empty()) bam.fetch_parts(tid, parts, r.b.get());
      r.complete = true; return r;
    }
    bam.fetch(tid, s, e, r.b.get(), cap); return r;
  };
  mkp_internal_bedmask_reset(ctx);
  struct MaskSession { mkp_ctx* c; ~MaskSession() { mkp_internal_bedmask_reset(c); } } mask_session{ctx};   // the host masks below die with this call
  if (a.serial_sampler) { sample_serial(ctx, bam, a, region, bf); return; }
  if (sharded && !(a.have_frac && a.sampling_frac >= 1.0)) throw Error(MKP_E_UNSUPPORTED,
      "rank-sharded threshold sampling needs the full-data mode (-f 1.0): the count-based schedule carries quotas from interval to interval");
  IdxStats st = idxstats(bam, region, bf);
  const uint64_t total_u = only_mapped ? st.mapped : st.mapped + st.unmapped;
  if (total_u == 0) throw Error(MKP_E_THRESHOLD, "zero reads found in bam index");
  std::map<uint32_t, Quota> quota; bool sched_unmapped = !only_mapped;
  if (a.have_frac) {  // from_sample_frac (321-381)
    if (a.sampling_frac > 1.0) throw Error(MKP_E_INVALID, "sample fraction must be <= 1");
    const float f = (float)a.sampling_frac;
    for (auto& kv : st.mapped_by_tid) if (kv.second) { Quota q; if (f == 1.0f) q.all = true; else q.n = (size_t)ceilf((float)kv.second * f);
        quota[(uint32_t)kv.first] = q; }
  } else {  // from_num_reads (171-273)
    const float total = (float)total_u; size_t sum = 0;
    for (auto& kv : st.mapped_by_tid) if (kv.second) { Quota q;
      q.n = std::min<size_t>((size_t)ceilf((float)a.num_reads * ((float)kv.second / total)), (size_t)kv.second);
        sum += q.n; quota[(uint32_t)kv.first] = q; }
    if (!only_mapped) sum += (size_t)ceilf((float)a.num_reads * ((float)st.unmapped / total));
    size_t floor = 1;
    while ((double)sum / (double)a.num_reads > 1.5) {  // pruning walks an FxHashMap in the reference; ascending tid here (order unpinned)
      for (auto& kv : quota) { if (kv.second.n <= floor) { sum -= kv.second.n; kv.second.n = 0; } if (sum <= a.num_reads) break; }
      sum = 0; for (auto& kv : quota) sum += kv.second.n; floor++;
    }
    for (auto it = quota.begin(); it != quota.end();) { if (!it->second.all && it->second.n == 0) it = quota.erase(it); else ++it; }
  }
  const size_t batch_size = (size_t)floorf((float)a.threads * 1.5f);
  std::vector<Contig> contigs; for (auto& c : targets(bam, region)) if (quota.count(c.tid)) contigs.push_back(c);
  std::map<uint32_t, uint32_t> contig_size; for (auto& c : contigs) contig_size[c.tid] = c.length;
  std::map<uint32_t, uint64_t> contig_base, contig_start; uint64_t grid_bp = 0; for (auto& c : contigs) { contig_base[c.tid] = grid_bp;
    contig_start[c.tid] = c.start;
      grid_bp += c.length; }
  std::set<std::string> taken; std::map<uint32_t, size_t> sampled_so_far;
  std::map<uint32_t, std::vector<uint8_t>> bedmasks;
  auto bedmask_for = [&](uint32_t tid) -> const uint8_t* {
    if (!bf) return nullptr;
    auto it = bedmasks.find(tid); if (it != bedmasks.end()) return it->second.data();
    std::vector<uint8_t> m(bam.ref_lens[tid], 0);
    auto mark = [&](const std::map<uint32_t, std::vector<Span>>& mp, uint8_t bit) { auto f = mp.find(tid); if (f == mp.end()) return;
        for (auto& s : f->second) for (uint64_t p = s.s; p < std::min<uint64_t>(s.e, m.size()); p++) m[p] |= bit; };
    mark(bf->pos, 1); mark(bf->neg, 2);
    return bedmasks.emplace(tid, std::move(m)).first->second.data();
  };
  // process_records (read_ids_to_base_mod_probs.rs:223-362) over the candidate records, first-N semantics
  // One fetched batch of an interval: `cand` = indices into batch.recs that pass `candidates`, processed from cand[from] on.
  // `skip` (rank-sharded mode): candidates an earlier interval already took.  State across calls: used / n_reads_out.
  struct TakeState { size_t used = 0, n_reads_out = 0; };
  // the sampler's verdict on candidates cand[lo, hi) whose value counts are nv[0 ..): mask[k] = 1 where the read's values enter the sample
  auto decide = [&](const RecSet& batch, const std::vector<size_t>& cand, size_t lo, size_t hi, const uint32_t* nv, long limit,
      std::set<std::string>* interval_seen,
      TakeState* ts, const std::vector<uint8_t>* skip, uint8_t* mask, SeededSampler* draws = nullptr, double frac = 1.0) {
    for (size_t i = lo; i < hi; i++) {
      if (limit >= 0 && ts->used >= (size_t)limit) break;   // RecordSampler::ask -> Done
      const size_t k = i - lo;
      if (skip && (*skip)[i]) continue;
      // check_sample_frac (record_sampler.rs:80-86): one draw per record the iterator yields, before anything else is looked at.  Only the
      // unmapped leg draws, and there (no edge filter, no BED, no alignment) "the tags parse and hold a position" == "a value is left"
      if (draws && nv[k] != 0 && !draws->keep(frac)) continue;
      std::string name = batch.name(cand[i]);
      // with_mod_base_info drops reads whose tags fail or are empty before the sampler is asked; a read that parses
      // but keeps no position is asked, not counted, and not recorded
      if (interval_seen->count(name)) continue;
      if (nv[k] == 0) continue;
      interval_seen->insert(name); ts->used++; ts->n_reads_out++;
      if (taken.count(name)) continue;  // Moniod::op_mut keeps the first occurrence of a read id
      taken.insert(name);
      mask[k] = 1;
    }
  };
  // decode the records `which` (indices into the sets' records, set by set) in sampling mode: one device round
  auto sample_round = [&](const std::vector<std::pair<const RecSet*, size_t>>& which, uint32_t tid, bool mapped_contig, std::vector<uint32_t>* nv) {
    const uint32_t ws = 0, we = mapped_contig ? bam.ref_lens[tid] : 1; const uint8_t* mask = mapped_contig ? bedmask_for(tid) : nullptr;
    int rc;
    if (resident_mode) { std::vector<uint32_t> ids; ids.reserve(which.size()); for (auto& w : which) ids.push_back(w.first->idx[w.second]);
      rc = mkp_internal_sample_resident(ctx, ws, we, mask, ids.data(), (uint32_t)ids.size(), only_mapped, nv); }
    else { std::vector<mkp_record> recs; recs.reserve(which.size());
      for (auto& w : which) recs.push_back(w.first->b->view(w.first->b->recs[w.second]));
      rc = mkp_internal_sample(ctx, mapped_contig ? (int32_t)tid : -1, ws, we, mask, recs.data(), (uint32_t)recs.size(), only_mapped, nv); }
    if (rc != MKP_OK) throw Error(rc, mkp_last_error(ctx));
  };
  auto take = [&](const RecSet& batch, const std::vector<size_t>& cand, size_t from, long limit, uint32_t tid, bool mapped_contig,
      std::set<std::string>* interval_seen,
      TakeState* ts, const std::vector<uint8_t>* skip = nullptr, SeededSampler* draws = nullptr, double frac = 1.0) {
    size_t next = from;
    while (next < cand.size() && (limit < 0 || ts->used < (size_t)limit)) {
      size_t want = limit < 0 ? std::min<size_t>(cand.size() - next, 1u << 18) : std::max<size_t>(256, 2 * ((size_t)limit - ts->used));
      size_t hi = std::min(cand.size(), next + want);
      std::vector<std::pair<const RecSet*, size_t>> which; which.reserve(hi - next);
        for (size_t i = next; i < hi; i++) which.push_back({&batch, cand[i]});
      std::vector<uint32_t> nv; sample_round(which, tid, mapped_contig, &nv);
      std::vector<uint8_t> mask(which.size(), 0);
      decide(batch, cand, next, hi, nv.data(), limit, interval_seen, ts, skip, mask.data(), draws, frac);
      int rc = mkp_internal_sample_take(ctx, mask);
      if (rc != MKP_OK) throw Error(rc, mkp_last_error(ctx));
      next = hi;
    }
  };
  auto candidates = [&](const RecSet& batch, std::vector<size_t>* out) {
    out->clear();
    for (size_t i = 0; i < batch.size(); i++) if (batch.candidate(i, only_mapped || a.edge_filter.size())) out->push_back(i);
  };
  if (!contigs.empty()) {
    // ReferenceIntervalsFeeder over the sampling grid, batch_size super-batches (interval_chunks.rs:563-643)
    struct Iv { uint32_t tid, start, end; };
    std::vector<std::vector<Iv>> groups;  // MultiChromCoordinates in feeder order
    { std::vector<Iv> batch; uint32_t blen = 0;
      for (auto& c : contigs) for (uint32_t p = c.start; p < c.end();) {
        uint32_t e = (uint32_t)std::min<uint64_t>((uint64_t)p + a.sampling_interval_size, c.end());
          batch.push_back({c.tid, p, e}); blen += e - p; if (blen >= a.sampling_interval_size) { groups.push_back(batch); batch.clear(); blen = 0;
            } p = e; }
      if (!batch.empty()) groups.push_back(batch); }
    for (size_t g0 = 0; g0 < groups.size(); g0 += std::max<size_t>(batch_size, 1)) {
      std::vector<Iv> all;
        for (size_t g = g0; g < std::min(groups.size(), g0 + std::max<size_t>(batch_size, 1)); g++) for (auto& iv : groups[g]) all.push_back(iv);
      std::stable_sort(all.begin(), all.end(), [](const Iv& x, const Iv& y) { return x.tid != y.tid ? x.tid < y.tid : x.start < y.start; });
      // accumulate_sample_counts (sampling_schedule.rs:440-615)
      std::map<uint32_t, uint32_t> len_per; for (auto& iv : all) len_per[iv.tid] += iv.end - iv.start;
      std::map<uint32_t, Quota> per_chrom;
      for (auto& kv : len_per) { auto cs = contig_size.find(kv.first); auto q = quota.find(kv.first);
        if (cs == contig_size.end() || q == quota.end()) continue;
          size_t so_far = sampled_so_far.count(kv.first) ? sampled_so_far[kv.first] : 0; float f = (float)kv.second / (float)cs->second;
        if (q->second.all) per_chrom[kv.first] = q->second; else if (q->second.n > so_far) { Quota x;
          x.n = (size_t)ceilf(f * (float)(q->second.n - so_far));
            per_chrom[kv.first] = x; } }
      struct G { Iv iv; Quota q; }; std::vector<G> grouped; bool have_slack = false; Iv slack{0, 0, 0}; size_t slack_n = 0;
      auto merged = [](const Iv& x, const Iv& y) { return Iv{x.tid, std::min(x.start, y.start), std::max(x.end, y.end)}; };
      for (auto& iv : all) {
        auto pc = per_chrom.find(iv.tid); if (pc == per_chrom.end()) continue;
        if (pc->second.all) { grouped.push_back({iv, pc->second}); continue; }
        float f = (float)(iv.end - iv.start) / (float)len_per[iv.tid]; size_t x = (size_t)ceilf((float)pc->second.n * f); Quota qx; qx.n = x;
        if (x < 50) {
          if (have_slack) { if (slack.tid == iv.tid) { Iv m = merged(slack, iv); size_t tot = x + slack_n; if (tot < 50) { slack = m; slack_n = tot;
              } else { Quota q;
              q.n = tot; grouped.push_back({m, q}); have_slack = false; } } else { Quota q; q.n = slack_n; grouped.push_back({slack, q}); slack = iv;
                slack_n = x; } }
          else { have_slack = true; slack = iv; slack_n = x; }
        } else if (have_slack) { have_slack = false; if (slack.tid == iv.tid) { Quota q; q.n = slack_n + x; grouped.push_back({merged(slack, iv), q});
          } else { Quota q;
            q.n = slack_n; grouped.push_back({slack, q}); grouped.push_back({iv, qx}); } }
        else grouped.push_back({iv, qx});
      }
      if (have_slack) { Quota q; q.n = slack_n; grouped.push_back({slack, q}); }
      std::map<uint32_t, size_t> batch_counts;
      // intervals this rank samples, in order; the head of the next one is fetched while the current one is decoded
      std::vector<size_t> mine;
      for (size_t gi = 0; gi < grouped.size(); gi++) {
        const G& g = grouped[gi];
        if (bf && !bf->overlaps(g.iv.tid, g.iv.start, g.iv.end)) continue;
        if (sharded) {
          // owner of the interval: contiguous runs of the sampling grid by base pairs (as the pileup shards are dealt)
          const uint64_t mid = contig_base[g.iv.tid] + (g.iv.start - contig_start[g.iv.tid]) + (g.iv.end - g.iv.start) / 2;
          if (std::min<uint64_t>(a.world - 1, mid * a.world / std::max<uint64_t>(grid_bp, 1)) != a.rank) continue;
        }
        mine.push_back(gi);
      }
      auto cap_of = [&](const G& g) { return g.q.all ? SIZE_MAX : 2 * g.q.n + 128; };
      auto head_of = [&](size_t gi) {
        return std::unique_ptr<RecSet>(new RecSet(fetch_set(grouped[gi].iv.tid, grouped[gi].iv.start, grouped[gi].iv.end, cap_of(grouped[gi])))); };
      auto skip_for = [&](const G& g, const RecSet& batch, const std::vector<size_t>& cand, std::vector<uint8_t>* skip) {
        // rank-sharded mode: a read that reaches back into an earlier processed interval of this contig was taken there
        skip->assign(cand.size(), 0);
        for (size_t i = 0; i < cand.size(); i++) {
          const int64_t e_pos = batch.pos(cand[i]);
          for (int64_t s1 = g.iv.start; s1 > (int64_t)contig_start[g.iv.tid] && e_pos < s1;) {   // grid intervals before this one, nearest first
            const int64_t s0 = std::max<int64_t>((int64_t)contig_start[g.iv.tid], s1 - (int64_t)a.sampling_interval_size);
            if (!bf || bf->overlaps(g.iv.tid, (uint64_t)s0, (uint64_t)s1)) { (*skip)[i] = 1; break; }
            s1 = s0;
          }
        }
      };
      // the rest of an interval after its head (or all of it in full-data mode): sequential device rounds
      auto finish_interval = [&](const G& g, std::unique_ptr<RecSet>& head, const std::vector<size_t>& head_cand, size_t head_done,
          const std::vector<uint8_t>& head_skip, std::set<std::string>* seen, TakeState* ts) {
        const long limit = g.q.all ? -1 : (long)g.q.n;
        take(*head, head_cand, head_done, limit, g.iv.tid, true, seen, ts, sharded ? &head_skip : nullptr);
        const bool truncated = head->truncated(cap_of(g));
        if (!truncated || (limit >= 0 && ts->used >= (size_t)limit)) return;
        const size_t done = head->size();
        head.reset();
        // the head was not enough: the whole interval, records already seen skipped
        const RecSet whole = fetch_set(g.iv.tid, g.iv.start, g.iv.end, SIZE_MAX);
        std::vector<size_t> cand; candidates(whole, &cand);
        std::vector<uint8_t> skip; if (sharded) skip_for(g, whole, cand, &skip);
        size_t from = 0; while (from < cand.size() && cand[from] < done) from++;
        take(whole, cand, from, limit, g.iv.tid, true, seen, ts, sharded ? &skip : nullptr);
      };
      // run_batch (reads_sampler/mod.rs:259-338).  The count-based schedule needs only the head of every interval: the heads of up to
      // 8 consecutive intervals of one contig (32 when they are scans of a resident shard's digest) are fetched concurrently and decoded in ONE
      // device round; the sampler's first-N logic
      // then runs over them in interval order.  An interval its head does not satisfy is finished sequentially before the
      // following ones are judged (their reads may already be taken by it), and the remaining heads are decoded again.
      struct Pending { size_t gi; std::unique_ptr<RecSet> head; std::vector<size_t> cand; std::vector<uint8_t> skip; size_t n_first = 0; std::set<std::string> seen;
          TakeState ts; };
      for (size_t mi = 0; mi < mine.size();) {
        const G& g0 = grouped[mine[mi]];
        size_t mj = mi + 1;
        // (resident: a head is a scan of the digest — more intervals per device round)
        if (!g0.q.all) while (mj < mine.size() && mj - mi < (resident_mode ? 32u : 8u) && !grouped[mine[mj]].q.all
            && grouped[mine[mj]].iv.tid == g0.iv.tid) mj++;
        std::vector<std::future<std::unique_ptr<RecSet>>> futs;
        // (resident: a scan of the digest, no fetch to overlap)
        for (size_t k = mi; k < mj; k++) futs.push_back(std::async(resident_mode ? std::launch::deferred : std::launch::async, head_of, mine[k]));
        std::vector<Pending> pend(mj - mi);
        for (size_t k = mi; k < mj; k++) {
          Pending& P = pend[k - mi]; P.gi = mine[k];
          { auto t_f = std::chrono::steady_clock::now(); P.head = futs[k - mi].get(); g_sample_times.fetch_ms += ms_since(t_f); }
          candidates(*P.head, &P.cand); if (sharded) skip_for(grouped[P.gi], *P.head, P.cand, &P.skip);
          const G& g = grouped[P.gi];
          P.n_first = g.q.all ? 0 : std::min(P.cand.size(), std::max<size_t>(256, 2 * g.q.n));
        }
        if (g0.q.all) { finish_interval(g0, pend[0].head, pend[0].cand, 0, pend[0].skip, &pend[0].seen, &pend[0].ts);
          batch_counts[g0.iv.tid] += pend[0].ts.n_reads_out;
            mi = mj; continue; }
        for (size_t k0 = 0; k0 < pend.size();) {
          std::vector<std::pair<const RecSet*, size_t>> recs; std::vector<size_t> at(pend.size() + 1, 0);
          for (size_t k = k0; k < pend.size(); k++) { at[k] = recs.size();
              for (size_t i = 0; i < pend[k].n_first; i++) recs.push_back({pend[k].head.get(), pend[k].cand[i]}); }
          at[pend.size()] = recs.size();
          std::vector<uint32_t> nv;
          auto t_d = std::chrono::steady_clock::now();
          sample_round(recs, g0.iv.tid, true, &nv);
          g_sample_times.device_ms += ms_since(t_d); g_sample_times.rounds++; g_sample_times.reads += recs.size();
          auto t_h = std::chrono::steady_clock::now();
          std::vector<uint8_t> mask(recs.size(), 0);
          size_t unsatisfied = pend.size();
          for (size_t k = k0; k < pend.size(); k++) {
            Pending& P = pend[k]; const G& g = grouped[P.gi];
            decide(*P.head, P.cand, 0, P.n_first, nv.data() + at[k], (long)g.q.n, &P.seen, &P.ts, sharded ? &P.skip : nullptr, mask.data() + at[k]);
            const bool more = P.n_first < P.cand.size() || P.head->truncated(cap_of(g));
            if (P.ts.used < g.q.n && more) { unsatisfied = k; break; }   // the judgement of the later heads waits for this interval
          }
          g_sample_times.decide_ms += ms_since(t_h);
          t_d = std::chrono::steady_clock::now();
          { const int rc = mkp_internal_sample_take(ctx, mask); if (rc != MKP_OK) throw Error(rc, mkp_last_error(ctx)); }
          g_sample_times.device_ms += ms_since(t_d);
          if (unsatisfied == pend.size()) break;
          Pending& P = pend[unsatisfied];
          finish_interval(grouped[P.gi], P.head, P.cand, P.n_first, P.skip, &P.seen, &P.ts);
          k0 = unsatisfied + 1;
        }
        for (auto& P : pend) batch_counts[grouped[P.gi].iv.tid] += P.ts.n_reads_out;
        mi = mj;
      }
      for (auto& kv : batch_counts) sampled_so_far[kv.first] += kv.second;
    }
  }
  // reads_sampler/mod.rs:89-125 (rank-sharded: rank 0 takes the unmapped reads)
  if ((sched_unmapped || taken.size() < 100) && !only_mapped && a.rank == 0) {
    RecSet batch; batch.b.reset(new BamBatch()); bam.fetch_unmapped(batch.b.get());
    std::vector<size_t> cand; candidates(batch, &cand);
    long limit;
    if (!a.have_frac) limit = (long)(a.num_reads > taken.size() ? a.num_reads - taken.size() : 0);
    else if (a.sampling_frac >= 1.0) limit = -1;
    else limit = -1;
    // RecordSampler::new_from_options (record_sampler.rs:51-61): a fraction < 1 is a Bernoulli draw per record from StdRng — seeded by
    // --seed, else from entropy (no two runs of the reference agree: refused, --seed makes it a function of the input)
    const bool draws = a.have_frac && a.sampling_frac < 1.0;
    if (draws && !cand.empty() && !a.have_seed) throw Error(MKP_E_UNSUPPORTED,
        "unmapped-read sampling with --sampling-frac < 1 draws from an entropy-seeded rand::StdRng (record_sampler.rs:29-38): give --seed");
    SeededSampler rng(a.seed);
    std::set<std::string> seen; TakeState ts; take(batch, cand, 0, limit, 0, false, &seen, &ts, nullptr, draws ? &rng : nullptr, a.sampling_frac);
  }
}

// Full-data mode (`-f 1.0`, thresholds.rs:121-159) over shards that are resident in HBM.  sample_probabilities' schedule degenerates there
// to "every candidate read whose tags yield a value, once" — each sampling interval takes all of its reads, a read counts in the first
// interval it overlaps — so the shards are sampled directly, read by read in file order, without the interval machinery (and without its
// per-read std::set<std::string> lookups: 2.0 s of a 2.4 s run on the C4 scale model in round 4).  A read that lies in two shards of a
// contig belongs to the first one (`own_from`: the end of the previous shard's fetch), so the union over the shards — of one rank or of
// all ranks — is the single-rank sample; names seen before are skipped as the reference's Moniod keeps the first occurrence of a read id.
struct FullShard { uint32_t tid; int64_t own_from, ext_lo, ext_hi; std::function<const ShardHost*()> bind; };
struct NameKey { uint64_t a, b; bool operator==(const NameKey& o) const { return a == o.a && b == o.b; } };
struct NameKeyHash { size_t operator()(const NameKey& k) const { return (size_t)(k.a ^ (k.b * 0x9e3779b97f4a7c15ull)); } };
void sample_resident_full(mkp_ctx* ctx, const BamSource& bam, const BedFilter* bf, std::vector<FullShard>& shards) {
  mkp_internal_bedmask_reset(ctx);
  struct MaskSession { mkp_ctx* c; ~MaskSession() { mkp_internal_bedmask_reset(c); } } mask_session{ctx};
  std::map<uint32_t, std::vector<uint8_t>> bedmasks;
  auto bedmask_for = [&](uint32_t tid) -> const uint8_t* {
    if (!bf) return nullptr;
    auto it = bedmasks.find(tid); if (it != bedmasks.end()) return it->second.data();
    std::vector<uint8_t> m(bam.ref_lens[tid], 0);
    auto mark = [&](const std::map<uint32_t, std::vector<Span>>& mp, uint8_t bit) { auto f = mp.find(tid); if (f == mp.end()) return;
        for (auto& sp : f->second) for (uint64_t p = sp.s; p < std::min<uint64_t>(sp.e, m.size()); p++) m[p] |= bit; };
    mark(bf->pos, 1); mark(bf->neg, 2);
    return bedmasks.emplace(tid, std::move(m)).first->second.data();
  };
  std::unordered_set<NameKey, NameKeyHash> taken;
  for (auto& fs : shards) {
    auto t_f = std::chrono::steady_clock::now();
    const ShardHost* S = fs.bind();
    g_sample_times.fetch_ms += ms_since(t_f);
    if (!S || (int32_t)fs.tid != S->tid) throw Error(MKP_E_INVALID, "internal: resident sampling outside the ingested contigs");
    const size_t n = S->hdr.size(), n_so = S->so_hdr.size();
    auto owned = [&](const MkpReadHdr& h) { const int64_t pos = h.ref_start, end = std::max(h.ref_end, h.ref_start + 1);
      return pos >= fs.own_from && pos < fs.ext_hi && end > fs.ext_lo; };
    std::vector<uint32_t> ids; ids.reserve(n + n_so);
    for (size_t i = 0; i < n; i++) if (owned(S->hdr[i])) ids.push_back((uint32_t)i);
    { bool any_so = false; for (size_t k = 0; k < n_so; k++) if (owned(S->so_hdr[k])) { ids.push_back((uint32_t)(n + k)); any_so = true; }
      if (any_so) { auto wi = [&](uint32_t k) { return k < n ? S->dev_win_idx[k] : S->so_win_idx[k - n]; };
        std::stable_sort(ids.begin(), ids.end(), [&](uint32_t x, uint32_t y) { return wi(x) < wi(y); }); } }
    const uint8_t* mask = bedmask_for(fs.tid);
    for (size_t at = 0; at < ids.size();) {
      const size_t hi = std::min(ids.size(), at + ((size_t)1 << 18));
      std::vector<uint32_t> nv;
      auto t_d = std::chrono::steady_clock::now();
      int rc = mkp_internal_sample_resident(ctx, 0, bam.ref_lens[fs.tid], mask, ids.data() + at, (uint32_t)(hi - at), true, &nv);
      if (rc != MKP_OK) throw Error(rc, mkp_last_error(ctx));
      g_sample_times.device_ms += ms_since(t_d); g_sample_times.rounds++; g_sample_times.reads += hi - at;
      auto t_h = std::chrono::steady_clock::now();
      std::vector<uint8_t> take(hi - at, 0);
      for (size_t i = at; i < hi; i++) {
        if (nv[i - at] == 0) continue;   // a read that keeps no position is asked, not counted and not recorded
        const uint32_t k = ids[i];
          NameKey key{k < n ? S->name_hash[k] : S->so_name_hash[k - n], k < n ? S->dev_name_hash2[k] : S->so_name_hash2[k - n]};
        if (taken.insert(key).second) take[i - at] = 1;
      }
      g_sample_times.decide_ms += ms_since(t_h);
      t_d = std::chrono::steady_clock::now();
      rc = mkp_internal_sample_take(ctx, take); if (rc != MKP_OK) throw Error(rc, mkp_last_error(ctx));
      g_sample_times.device_ms += ms_since(t_d);
      at = hi;
    }
  }
}

// per-base pass thresholds from the context's resident sample: two-level histogram -> the two order statistics -> interpolation
void thresholds_from_sample(mkp_ctx* ctx, float q, float thr[4], uint8_t has[4], bool verbose, uint64_t* n_out = nullptr) {
  std::vector<uint64_t> h0(65536), h1(65536);
  for (uint32_t b = 0; b < 4; b++) {
    thr[b] = 0.f; has[b] = 0;
    int rc = mkp_histogram_get(ctx, b, 0, 0, h0.data()); if (rc != MKP_OK) throw Error(rc, mkp_last_error(ctx));
    uint32_t bins[2]; uint64_t rk[2], n = 0;
    rc = mkp_histogram_locate(h0.data(), q, bins, rk, &n);
    if (n_out) n_out[b] = n;
    if (n == 0) continue;   // no calls on this base
    if (rc != MKP_OK) throw Error(MKP_E_THRESHOLD, "not enough datapoints, got " + std::to_string(n));
    float y[2];
    for (int k = 0; k < 2; k++) {
      if (k == 1 && bins[1] == bins[0] && rk[1] == rk[0]) { y[1] = y[0]; break; }
      if (k == 0 || bins[1] != bins[0]) { rc = mkp_histogram_get(ctx, b, 1, bins[k], h1.data());
        if (rc != MKP_OK) throw Error(rc, mkp_last_error(ctx));
        }
      rc = mkp_histogram_resolve(bins[k], h1.data(), rk[k], &y[k]);
        if (rc != MKP_OK) throw Error(MKP_E_THRESHOLD, "internal: histogram levels disagree");
    }
    float t; rc = mkp_percentile_from_histogram(n, q, y[0], y[1], &t);
      if (rc != MKP_OK) throw Error(MKP_E_THRESHOLD, "not enough datapoints, got " + std::to_string(n));
    thr[b] = t; has[b] = 1;
    if (verbose) fprintf(stderr, "[mkpileup] threshold %c %.9g (n=%llu)\n", "ACGT"[b], (double)t, (unsigned long long)n);
  }
}

void parse_base_thresholds(const std::vector<std::string>& raws, mkp_caller* k) {  // parse_per_base_thresholds (command_utils.rs:136-206)
  bool have_default = false;
  for (auto& raw : raws) {
    size_t c = raw.find(':');
    if (c != std::string::npos) {
      if (raw.find(':', c + 1) != std::string::npos || c == 0) throw Error(MKP_E_INVALID, "encountered illegal per-base threshold " + raw);
      int b = (int)std::string("ACGT").find(raw[0]); if (b < 0 || b > 3) throw Error(MKP_E_INVALID, "failed to parse base in " + raw);
      if (k->has_per_base[b]) throw Error(MKP_E_INVALID, "repeated threshold for base");
      k->has_per_base[b] = 1; k->per_base_threshold[b] = strtof(raw.c_str() + c + 1, nullptr);
    } else { if (have_default) throw Error(MKP_E_INVALID, "default threshold encountered more than once"); have_default = true;
        k->default_threshold = strtof(raw.c_str(), nullptr); }
  }
}

int run(const Args& a, mkp_ctx* ext_ctx, mkp_run_report* rep) {
  auto t_all = std::chrono::steady_clock::now();
  // BAI next to the BAM: only the blocks of the shards (and sampling intervals) this run touches are read and inflated; otherwise
  // the whole file is loaded once.  (inflate threads: --threads only steers the sampling schedule)
  std::unique_ptr<BamSource> src = BamSource::open(a.in_bam, std::max(std::max<unsigned>((unsigned)a.threads, 1u), HostPool::host_cpus()),
      !a.no_index);
  const BamSource& bam = *src;
  double load_ms = ms_since(t_all);
  const bool trace = getenv("MKP_TRACE_PLAN") != nullptr;   // wall-clock marks of the subcommand's phases on stderr
  auto mark = [&](const char* what) { if (trace) fprintf(stderr, "[mkpileup run] %-36s at %.1f ms\n", what, ms_since(t_all)); };
  mark("BAM opened (header, index)");
  RegionSpec region, sregion; const bool have_region = !a.region.empty(), have_sregion = !a.sample_region.empty();
  if (have_region) region = parse_region(a.region, bam);
  if (have_sregion) sregion = parse_region(a.sample_region, bam);
  mkp_caller kc; memset(&kc, 0, sizeof(kc)); kc.max_depth = a.max_depth; kc.force_allow_implicit = a.force_allow;
  if (!a.edge_filter.empty()) {  // parse_edge_filter_input (command_utils.rs:243-277)
    kc.edge_filter = 1; kc.edge_inverted = a.invert_edge; size_t c = a.edge_filter.find(',');
    if (c != std::string::npos) { kc.edge_start = (uint32_t)strtoul(a.edge_filter.c_str(), nullptr, 10);
        kc.edge_end = (uint32_t)strtoul(a.edge_filter.c_str() + c + 1, nullptr, 10); }
    else kc.edge_start = kc.edge_end = (uint32_t)strtoul(a.edge_filter.c_str(), nullptr, 10);
  }
  std::vector<mkp_mod_threshold> per_mod;
  for (auto& raw : a.mod_thresholds) { size_t c = raw.find(':'); uint32_t code;
      if (c == std::string::npos || raw.find(':', c + 1) != std::string::npos || !parse_code(raw.substr(0, c), &code)) throw Error(MKP_E_INVALID,
      "encountered illegal per-mod threshold: " + raw);
      per_mod.push_back({code, strtof(raw.c_str() + c + 1, nullptr)}); }
  std::vector<Contig> records = targets(bam, have_region ? &region : nullptr);
  BedFilter bed_store; const BedFilter* bf = nullptr;
  if (!a.include_bed.empty()) { std::map<std::string, uint32_t> c2t; for (auto& r : records) c2t[r.name] = r.tid;
    bed_store = BedFilter::load(a.include_bed, c2t);
      bf = &bed_store; }
  if (idxstats(bam, have_region ? &region : nullptr, bf).mapped == 0) throw Error(MKP_E_INVALID,
      "did not find any mapped reads, perform alignment first or use modkit extract and/or modkit summary to inspect unaligned modBAMs");
  if (a.filter_percentile > 1.0f) throw Error(MKP_E_INVALID, "filter percentile must be <= 1.0");
  if (a.combine_strands && !(a.cpg || !a.motif_parts.empty())) throw Error(MKP_E_INVALID,
      "need to specify either --motif or --cpg to combine strands");
  // subcommand.rs:1247-1276: one motif, --cpg or --motif (a clap argument group: not both), palindromic; the reference FASTA is required
  if (a.hemi) {
    if (!a.cpg && a.motif_parts.empty()) throw Error(MKP_E_INVALID, "either --cpg or a --motif must be provided for pileup-hemi");
    if (a.cpg && !a.motif_parts.empty()) throw Error(MKP_E_INVALID, "the argument '--cpg' cannot be used with '--motif'");
    if (a.motif_parts.size() > 2) throw Error(MKP_E_INVALID, "motif arg should be length 2, eg. CG 0");
    if (a.ref_fasta.empty()) throw Error(MKP_E_INVALID, "the following required arguments were not provided: --ref <REFERENCE_FASTA>");
  }
  bool combine_strands = a.combine_strands;  // option resolution (subcommand.rs:484-523)
  if (a.preset == "traditional") { kc.numeric_mode = 2; kc.collapse_code = 'h'; combine_strands = true; }
  else if (!a.preset.empty()) throw Error(MKP_E_INVALID, "unknown preset " + a.preset);
  else if (a.combine_mods) kc.numeric_mode = 1;
  else if (!a.ignore.empty()) { uint32_t code; if (!parse_code(a.ignore, &code)) throw Error(MKP_E_INVALID, "failed to parse mod code " + a.ignore);
    kc.numeric_mode = 2;
      kc.collapse_code = code; }
  kc.combine_strands = combine_strands;
  if (a.hemi) combine_strands = true;
      // the interval feeder runs with combine_strands = true ("must be true for duplex", subcommand.rs:1383-1390); the caller's flag stays off
  FocusBuilder fb; fb.combine = combine_strands; fb.mask = a.mask; fb.bed = bf;
  Fasta fasta; std::future<Fasta> fasta_load;
  struct JoinFasta { std::future<Fasta>* f; ~JoinFasta() { if (f->valid()) f->wait(); } } join_fasta{&fasta_load};
  if (!a.motif_parts.empty()) {  // RegexMotif::from_raw_parts (motif_bed.rs:152-195)
    if (!a.preset.empty()) throw Error(MKP_E_INVALID, "cannot use presets and motifs together");
    std::vector<std::string> parts = a.motif_parts;
    for (size_t i = 0; i + 1 < parts.size(); i += 2) for (size_t j = i + 2; j + 1 < parts.size(); j += 2) if (parts[i] == parts[j]
        && parts[i + 1] == parts[j + 1]) throw Error(MKP_E_INVALID,
        "cannot have the same motif more than once");
    if (a.cpg) { bool has = false; for (size_t i = 0; i + 1 < parts.size(); i += 2) if (parts[i] == "CG" && parts[i + 1] == "0") has = true;
        if (!has) { parts.push_back("CG"); parts.push_back("0"); } }
    for (size_t i = 0; i + 1 < parts.size(); i += 2) fb.motifs.push_back(Motif::parse(parts[i], strtoul(parts[i + 1].c_str(), nullptr, 10)));
  } else if (a.preset == "traditional" || a.cpg) fb.motifs.push_back(Motif::parse("CG", 0));
  RowWriter wr; wr.mixed = a.mixed_delim; for (auto& m : fb.motifs) wr.labels.push_back(m.label());
  if (!fb.motifs.empty()) {
    if (a.ref_fasta.empty()) throw Error(MKP_E_INVALID, "reference fasta is required for using --motif or --cpg options");
    if (combine_strands) for (auto& m : fb.motifs) if (!m.palindrome) throw Error(MKP_E_INVALID,
        a.hemi ? "motif must be palindromic for pileup-hemi" : "cannot combine strands with a motif that is not a palindrome");
    // joined where the grid walk needs it: the device ingest of the first shard starts meanwhile
    fasta_load = std::async(std::launch::async, [&]() { return Fasta::load(a.ref_fasta); });
  }
  mkp_config cfg; memset(&cfg, 0, sizeof(cfg)); cfg.device = a.device; cfg.tile_positions = a.tile;
  mkp_ctx* ctx = ext_ctx;
  int rc = (a.plan_only || ext_ctx) ? MKP_OK : mkp_ctx_create(&cfg, &ctx);   // --plan-only: shard plan of this rank, no device work
  if (rc != MKP_OK) throw Error(rc, "no usable gfx950 device (libmkpileup has no CPU path)");
  struct Guard { mkp_ctx* c; ~Guard() { if (c) mkp_ctx_destroy(c); } } guard{ext_ctx ? nullptr : ctx};
  auto must = [&](int r) { if (r != MKP_OK) throw Error(r, mkp_last_error(ctx)); };
  // --device-inflate (or MKP_DEVICE_INFLATE=1): the shard windows' BGZF blocks are inflated on the GPU; destroyed after the last fetch (declared
  // before everything that fetches, so it outlives the prefetch threads on every path out of here)
  struct InflaterGuard { mkp_dev_inflater* d = nullptr; BamSource* src = nullptr; ~InflaterGuard() { if (src) { src->dev_inflate = nullptr;
        src->dev_inflate_user = nullptr; } mkp_internal_inflater_destroy(d); } } inflater;
  if ((a.device_inflate || (getenv("MKP_DEVICE_INFLATE") && !strcmp(getenv("MKP_DEVICE_INFLATE"), "1"))) && !a.plan_only && bam.indexed()) {
    inflater.d = mkp_internal_inflater_create(a.device);
    if (!inflater.d) throw Error(MKP_E_DEVICE, "--device-inflate: cannot create a stream on the device");
    inflater.src = src.get(); src->dev_inflate = mkp_internal_device_inflate; src->dev_inflate_user = inflater.d;
  }
  // The interval grid and the focus bytes need the reference and the BED only: with one rank they are built on a second thread
  // while the thresholds are being estimated (both are all-cores work in short bursts; neither waits for the other's results).
  if (bf) records = bed_contigs(*bf, records, a.interval_size);
  std::vector<std::vector<uint8_t>> focus_of(records.size()); std::vector<char> focus_done(records.size(), 0);
  std::vector<std::vector<Interval>> grid_of(records.size()); std::vector<char> grid_done(records.size(), 0);
  double focus_ms = 0;
  // shards: pieces of the contig records cut at interval boundaries, bounded in positions (tally / focus buffers) and in BAM bytes
  // (host memory: a shard's blocks are inflated and packed as a unit)
  uint64_t total_bp = 0; for (auto& r : records) total_bp += r.length;
  const uint64_t shard_bp = a.shard_bp ? a.shard_bp : (a.world > 1 ? std::max<uint64_t>(a.interval_size, std::min<uint64_t>(1ull << 27,
      (total_bp + a.world * 8 - 1) / (a.world * 8))) : (1ull << 27));
  // 2^27 positions per shard keeps the per-shard focus / slot buffers small
  // BAM bytes per shard: the device ingest's inflate pays a fixed latency per launch and HBM holds the window many times over — 1 GiB of
  // compressed blocks at a time; the host path keeps 256 MiB (its inflated window lives in host memory)
  const bool dev_ingest_plan = !a.no_index && !a.plan_only && !a.host_ingest && a.partition_tags.empty() && !a.device_inflate
      && !(getenv("MKP_HOST_INGEST") && !strcmp(getenv("MKP_HOST_INGEST"), "1"));
  const uint64_t shard_bytes = (a.shard_bytes_set || !dev_ingest_plan) ? a.shard_bytes : (1ull << 30);
  auto shard_cut = [&](const Contig& rec, const std::vector<Interval>& ivs, size_t i0, uint64_t* bp_out) {   // -> one past the shard's last interval
    size_t i1 = i0; uint64_t bp = 0; const uint64_t o0 = bam.offset_at(rec.tid, ivs[i0].start);
    while (i1 < ivs.size() && (bp == 0 || (bp + (ivs[i1].end - ivs[i1].start) <= shard_bp && (!bam.indexed() || bam.offset_at(rec.tid,
        ivs[i1].end) - o0 <= shard_bytes)))) { bp += ivs[i1].end - ivs[i1].start; i1++; }
    *bp_out = bp; return i1;
  };
  // Shard records come from the device ingest (mkp_ingest_host.cpp: compressed blocks up, inflate + record cut + tag tokeniser + packing in
  // HBM, a digest back) whenever the BAM is indexed; --host-ingest / MKP_HOST_INGEST=1, --partition-tag (keys are read from aux fields on
  // the host) and --plan-only keep the host reader + packer.
  struct ShardInput { std::unique_ptr<BamBatch> batch; std::unique_ptr<DevShard> dev; };
  const bool host_ingest_env = getenv("MKP_HOST_INGEST") && !strcmp(getenv("MKP_HOST_INGEST"), "1");
  const bool dev_ingest = bam.indexed() && !a.plan_only && !a.host_ingest && a.partition_tags.empty() && !inflater.d && !host_ingest_env;
  if (dev_ingest && !ctx->ingest) { ctx->ingest = mkp_internal_ingest_create(ctx->device);
    if (!ctx->ingest) throw Error(MKP_E_DEVICE, "device ingest: cannot create streams on the device");
    }
  double ingest_ms[5] = {0, 0, 0, 0, 0}, ingest_kernel_ms = 0; uint64_t ingest_blocks = 0, ingest_records = 0, ingest_comp = 0, ingest_raw = 0;
  // the records of a shard: those overlapping any of its windows (one window, or the BED spans of a merged shard), each +- the halo
  auto fetch_windows = [&](uint32_t tid, const std::vector<std::pair<uint32_t, uint32_t>>& wins, mkp_dev_ingest* ing = nullptr) {
    ShardInput in; FetchParts parts;
    for (auto& w : wins) { const int64_t lo = w.first > MKP_HALO ? (int64_t)w.first - MKP_HALO : 0, hi = (int64_t)w.second + MKP_HALO;
      if (!parts.empty() && lo <= parts.back().second) parts.back().second = std::max(parts.back().second, hi); else parts.push_back({lo, hi}); }
    // foreground: the upload feeds the GPU's longest job of the run
    if (dev_ingest) { in.dev = mkp_internal_ingest_run(ing ? ing : ctx->ingest, bam, tid, parts); return in; }
    HostPool::background() = true; in.batch.reset(new BamBatch()); bam.fetch_parts(tid, parts, in.batch.get()); return in; };
  auto fetch_range = [&](uint32_t tid, uint32_t s0, uint32_t s1) { return fetch_windows(tid, {{s0, s1}}); };
  // compressed bytes a set of fetch windows stands for (the index's 16 kb granularity: a short window costs at least its blocks)
  auto win_bytes = [&](uint32_t tid, const std::vector<std::pair<uint32_t, uint32_t>>& wins) { uint64_t b = 0;
    for (auto& w : wins) b += bam.offset_at(tid, (uint64_t)w.second + 16384) - bam.offset_at(tid, w.first > 16384 ? w.first - 16384 : 0) + (1u << 16);
    return b; };
  // The first shard's blocks are read and inflated behind the threshold estimate (background priority on the host pool: the estimate's
  // own bursts go first), as soon as the first contig's grid is known.
  std::future<ShardInput> early_fetch; uint32_t early_s0 = 0, early_s1 = 0; bool early_set = false;
  // One target contig that fits one shard whatever its grid turns out to be: the device ingest of that window starts now, before the
  // reference is read and the grid walked (if the grid's one shard is not this window after all, the ingest is repeated for the plan's)
  bool early_whole = false;
  if (dev_ingest && a.world == 1 && records.size() == 1 && records[0].length > 0 && !getenv("MKP_NO_EARLY_FETCH")) {
    const Contig& r0 = records[0];
    if (r0.length <= shard_bp && bam.offset_at(r0.tid, r0.end()) - bam.offset_at(r0.tid, r0.start) <= shard_bytes) {
      early_s0 = r0.start; early_s1 = r0.end(); early_set = true; early_whole = true;
      early_fetch = std::async(std::launch::async, fetch_range, r0.tid, early_s0, early_s1);
    }
  }
  struct JoinFetch { std::future<ShardInput>* f; ~JoinFetch() { if (f->valid()) f->wait(); } } join_fetch{&early_fetch};
  // Several target contigs (or the BED records of several contigs), each fitting one shard: all of them are ingested AHEAD, one after the
  // other on a worker thread, and stay packed in HBM (288 GB hold a 30x genome's packed reads) — the threshold estimate then samples from
  // them (every contig's interval heads are scans of a digest, no second read of the file), and the pileup pass finds its shards already
  // there.  Budget: the compressed bytes under the shards; beyond it the shards are fetched as the loop reaches them (host sampler).
  struct Ahead { size_t rec0 = 0, rec1 = 0; uint32_t tid = 0, s0 = 0, s1 = 0; std::vector<std::pair<uint32_t,
      uint32_t>> wins; ShardInput in; bool ready = false; std::exception_ptr err; uint64_t bytes = 0, est = 0; };
  // Several shards are in flight at once, each on an ingest object of its own (its streams, staging and scratch): one shard's inflate is a
  // launch that fills a fraction of the chip for most of its time, and its host half (pread into staging) leaves the GPU idle —
  // shards side by side fill both.  What they may hold is bounded by an HBM BUDGET (round 5; round 4 had an all-or-nothing 24 GiB gate on
  // the compressed size): a shard is admitted — in plan order — while the estimated bytes of the admitted, not yet consumed shards fit the
  // budget (one shard is always admitted), and the shard loop gives a shard's bytes back when it is through with it.  A run whose shards
  // fit together keeps them all (and may sample its threshold estimate from them); a larger one streams.
  std::vector<Ahead> ahead; std::mutex amu; std::condition_variable acv; std::vector<std::thread> aworkers; std::atomic<bool> astop{false};
    std::atomic<size_t> anext{0};
  size_t admit_next = 0; uint64_t hbm_used = 0, hbm_budget = 0, ahead_est_total = 0;   // (under amu)
  std::vector<mkp_dev_ingest*> aingest;
  struct FreeIngest { std::vector<mkp_dev_ingest*>* v; ~FreeIngest() { for (auto* d : *v) mkp_internal_ingest_destroy(d); } } free_ingest{&aingest};
  struct JoinAhead { std::vector<std::thread>* t; std::atomic<bool>* stop; std::condition_variable* cv; ~JoinAhead() { stop->store(true);
      cv->notify_all(); for (auto& x : *t) if (x.joinable()) x.join(); } } join_ahead{&aworkers, &astop, &acv};
  if (dev_ingest) {
    uint64_t mb = a.hbm_budget_mb; if (const char* e = getenv("MKP_HBM_BUDGET_MB")) mb = strtoull(e, nullptr, 10);
    if (mb) hbm_budget = mb << 20;
    else { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess || !tot) tot = (size_t)64 << 30;
      hbm_budget = (uint64_t)((double)tot * 0.55); }
  }
  // packed arrays of a shard ~ 1.9 x its compressed blocks on ONT-like data (SEQ nibbles + CIGAR words + 5 bytes per call; names and
  // qualities are not kept), the inflated window of the ingest object in flight on top: 2.5 x as the estimate
  auto est_of = [](uint64_t comp_bytes) { return comp_bytes * 5 / 2 + (64ull << 20); };
  auto start_ahead = [&]() {
    if (ahead.empty() || !aworkers.empty()) return;
    ahead_est_total = 0; for (auto& A : ahead) { A.est = est_of(A.bytes); ahead_est_total += A.est; }
    // two shards in flight: a shard's own upload, block table and inflate overlap since the staged ingest, a second shard fills the gaps, and
    // more only crowd the short kernels of the threshold estimate beside them (C4 scale model, CLI: 1 worker 1.42 s, 2 1.22, 4 1.35, 8 1.34)
    size_t nw = 2; if (const char* e = getenv("MKP_AHEAD_WORKERS")) nw = (size_t)std::max(1, atoi(e));
    nw = std::min(nw, ahead.size());
    for (size_t w = 1; w < nw; w++) { mkp_dev_ingest* d = mkp_internal_ingest_create(ctx->device); if (!d) break; aingest.push_back(d); }
    nw = aingest.size() + 1;
    for (size_t w = 0; w < nw; w++) aworkers.emplace_back([&, w]() {
      mkp_dev_ingest* ing = w == 0 ? ctx->ingest : aingest[w - 1];
      for (;;) {
        const size_t k = anext.fetch_add(1); if (k >= ahead.size()) break;
        { std::unique_lock<std::mutex> lk(amu);   // admission in plan order, under the budget
          acv.wait(lk, [&] { return astop.load() || (admit_next == k && (hbm_used == 0 || hbm_used + ahead[k].est <= hbm_budget)); });
          if (!astop.load()) { hbm_used += ahead[k].est; admit_next = k + 1; } }
        acv.notify_all();
        ShardInput in; std::exception_ptr err;
        if (astop.load()) err = std::make_exception_ptr(Error(MKP_E_INVALID, "internal: shard ingest cancelled"));
        else try { in = fetch_windows(ahead[k].tid, ahead[k].wins, ing); } catch (...) { err = std::current_exception(); astop.store(true); }
        { std::lock_guard<std::mutex> g(amu); ahead[k].in = std::move(in); ahead[k].err = err; ahead[k].ready = true; }
        acv.notify_all();
      }
    });
  };
  // the shard loop is through with shard k
  auto ahead_release = [&](size_t k) { { std::lock_guard<std::mutex> g(amu); hbm_used -= std::min(hbm_used, ahead[k].est); } acv.notify_all(); };
  // the fetch windows of a shard made of BED records [r0, r1) of one contig: the BED spans inside them (rows exist at BED positions only,
  // so only records reaching a span matter), not the records, which run from one span to the next
  auto bed_windows = [&](size_t r0, size_t r1) {
    std::vector<std::pair<uint32_t, uint32_t>> w; const uint32_t tid = records[r0].tid; std::vector<Span> sp;
    for (auto* m : {&bf->pos, &bf->neg}) { auto it = m->find(tid); if (it == m->end()) continue;
      for (auto& x : it->second) {
        const uint64_t lo = std::max<uint64_t>(x.s, records[r0].start), hi = std::min<uint64_t>(x.e, records[r1 - 1].end());
        if (lo < hi) sp.push_back({lo, hi});
        } }
    merge_spans(sp);
    // keep what lies inside the records (sorted, disjoint): one sweep
    { std::vector<Span> in; size_t r = r0;
      for (auto& x : sp) { while (r < r1 && records[r].end() <= x.s) r++;
        for (size_t q = r; q < r1 && records[q].start < x.e; q++) {
          const uint64_t lo = std::max<uint64_t>(x.s, records[q].start), hi = std::min<uint64_t>(x.e, records[q].end());
          if (lo < hi) in.push_back({lo, hi});
          } }
      sp.swap(in); merge_spans(sp); } for (auto& x : sp) w.push_back({(uint32_t)x.s, (uint32_t)x.e});
    return w;
  };
  if (dev_ingest && !early_whole && a.world == 1 && !records.empty() && !getenv("MKP_NO_EARLY_FETCH") && !getenv("MKP_NO_AHEAD")) {
    uint64_t total = 0; bool fits = true;
    for (size_t r0 = 0; r0 < records.size() && fits;) {
      size_t r1 = r0 + 1; if (bf) while (r1 < records.size() && records[r1].tid == records[r0].tid) r1++;
      Ahead A; A.rec0 = r0; A.rec1 = r1; A.tid = records[r0].tid; A.s0 = records[r0].start; A.s1 = records[r1 - 1].end();
      if (bf) A.wins = bed_windows(r0, r1); else A.wins.push_back({A.s0, A.s1});
      A.bytes = bf ? win_bytes(A.tid, A.wins) : bam.offset_at(A.tid, A.s1) - bam.offset_at(A.tid, A.s0);
      if ((uint64_t)A.s1 - A.s0 > shard_bp || A.bytes > shard_bytes || (bf && r1 - r0 > 65536) || records[r0].length == 0
          || A.wins.empty()) fits = false;
      total += A.bytes; ahead.push_back(std::move(A)); r0 = r1;
    }
    (void)total;
    if (!fits) ahead.clear();   // a contig larger than a shard: the shards are cut on the grid and ingested ahead once the plan is known (below)
    start_ahead();
  }
  auto ahead_wait = [&](size_t k) -> Ahead& { std::unique_lock<std::mutex> lk(amu); acv.wait(lk,
      [&] { return ahead[k].ready; }); if (ahead[k].err) std::rethrow_exception(ahead[k].err); return ahead[k]; };
  if (fasta_load.valid()) { fasta = fasta_load.get(); fb.fasta = &fasta; mark("reference FASTA loaded"); }
  std::future<void> early_walk;
  if (fb.has_focus() && a.world == 1 && !a.plan_only && a.filter_threshold.empty() && !a.no_filtering)
    early_walk = std::async(std::launch::async, [&]() {
      auto t_focus = std::chrono::steady_clock::now();
      for (size_t ri = 0; ri < records.size(); ri++) {
        grid_of[ri] = fb.walk(records[ri], a.interval_size, &focus_of[ri]); grid_done[ri] = 1; focus_done[ri] = 1;
        // (--include-bed: the first shard is a merge of records, known only with the plan)
        if (ri == 0 && !early_whole && !bf && ahead.empty() && !grid_of[0].empty() && !getenv("MKP_NO_EARLY_FETCH")) {
          uint64_t bp; const size_t i1 = shard_cut(records[0], grid_of[0], 0, &bp);
          early_s0 = grid_of[0][0].start; early_s1 = grid_of[0][i1 - 1].end; early_set = true;
          early_fetch = std::async(std::launch::async, fetch_range, records[0].tid, early_s0, early_s1);
        }
      }
      focus_ms += ms_since(t_focus);
    });
  struct JoinWalk { std::future<void>* f; ~JoinWalk() { if (f->valid()) f->wait(); } } join_walk{&early_walk};
      // (an exception below must not leave the walker running on dead locals)
  // ---- shard plan (shard_cut above); ranks take contiguous runs, balanced by the bytes the index puts under them (by length
  // without an index)
  struct ShardPart { size_t rec; uint32_t s0, s1; };
  struct ShardPlan { size_t rec; uint32_t s0, s1; uint64_t bp; std::vector<uint32_t> iv_starts; /* pileup-hemi: starts of the shard's intervals */
                     std::vector<ShardPart> parts; /* --include-bed: the BED-span records merged into this shard (empty: one window) */
                     int64_t own_from = INT64_MIN;
                       /* full-data sampling from the shards: reads starting before this position lie in the previous shard of the contig too, and are sampled there */ };
  std::vector<ShardPlan> plan;
  // what every rank's shards would hold in HBM if they were all ingested ahead (est_of of each shard's bytes under the index), computed by
  // every rank for every rank: the one thing a rank of a multi-GPU run may base a choice on that the other ranks must make the same way
  std::vector<uint64_t> est_by_rank(std::max<uint32_t>(a.world, 1u), 0);
  const bool hf = fb.has_focus();
  uint64_t positions = 0, processed = 0, skipped = 0, n_shards = 0;
      double kernel_ms = 0, pack_ms = 0, h2d_ms = 0, d2h_ms = 0, dec_ms = 0, pil_ms = 0, row_ms = 0, write_ms = 0, fetch_wait_ms = 0;
  bool plan_built = false;
  auto build_plan = [&]() {
    if (plan_built) return; plan_built = true;
    if (early_walk.valid()) early_walk.get();   // (the walker owns grid_of / focus_of until it is done; rethrows what it threw)
  {
      const uint64_t off_lo = records.empty() ? 0 : bam.offset_at(records.front().tid,
          records.front().start), off_hi = records.empty() ? 0 : bam.offset_at(records.back().tid, records.back().end());
      bool have_prev = false; uint32_t prev_tid = 0; int64_t prev_fetch_hi = 0;   // the shard before, over ALL ranks' shards: where its fetch ends
      // end of the last fetch window of a shard [w0, w1) (plan_windows below, before the halo)
      auto last_window_end = [&](uint32_t tid, uint32_t w0, uint32_t w1) -> int64_t {
        if (!bf) return w1;
        uint64_t last = 0; for (auto* m : {&bf->pos, &bf->neg}) { auto it = m->find(tid); if (it == m->end()) continue;
          for (auto& x : it->second) { const uint64_t lo = std::max<uint64_t>(x.s, w0), hi = std::min<uint64_t>(x.e, w1);
            if (lo < hi) last = std::max(last, hi);
            } }
        return last ? (int64_t)last : (int64_t)w0 + 1;
      };
      for (size_t ri = 0; ri < records.size(); ri++) {
        const Contig& rec = records[ri];
        auto t_focus = std::chrono::steady_clock::now();
        // the grid; with one rank the focus bytes are filled in the same walk, otherwise only for the contigs this rank owns (below)
        std::vector<Interval> ivs;
        if (grid_done[ri]) ivs.swap(grid_of[ri]);
        else { ivs = fb.walk(rec, a.interval_size, (hf && a.world == 1) ? &focus_of[ri] : nullptr); if (hf && a.world == 1) focus_done[ri] = 1;
            focus_ms += ms_since(t_focus); }
        size_t i0 = 0;
        while (i0 < ivs.size()) {
          uint64_t bp = 0; const size_t i1 = shard_cut(rec, ivs, i0, &bp); const uint64_t o0 = bam.offset_at(rec.tid, ivs[i0].start);
          const uint32_t s0 = ivs[i0].start, s1 = ivs[i1 - 1].end;
          const uint64_t mid = (o0 + bam.offset_at(rec.tid, s1)) / 2;
          const uint32_t owner = off_hi > off_lo ? (uint32_t)std::min<uint64_t>(a.world - 1,
              (mid > off_lo ? mid - off_lo : 0) * a.world / (off_hi - off_lo)) : 0;
          // the shard's intervals (pileup-hemi: per-interval NoCalls; always: the duplicate-name rule)
          std::vector<uint32_t> iv_starts; for (size_t k = i0; k < i1; k++) iv_starts.push_back(ivs[k].start);
          i0 = i1;
          const int64_t own_from = have_prev && prev_tid == rec.tid ? prev_fetch_hi : INT64_MIN;
          have_prev = true; prev_tid = rec.tid; prev_fetch_hi = last_window_end(rec.tid, s0, s1) + MKP_HALO;
          est_by_rank[owner] += est_of(bam.indexed() ? bam.offset_at(rec.tid, s1) - o0 + (1u << 16) : 0);
          if (owner == a.rank) { plan.push_back({ri, s0, s1, bp, std::move(iv_starts)}); plan.back().own_from = own_from; }
        }
      }
    }
    // --include-bed: optimize_reference_records (position_filter.rs:103-210) turns every run of BED spans into a reference record of its own —
    // hundreds per contig for a sparse BED.  Rows only depend on the focus bytes (BED positions inside the records), not on how records are
    // grouped, so consecutive records of one contig share a shard: one hull window whose focus is zero between the records, fed by a
    // multi-window fetch of the records' spans only (what lies between them is neither read nor inflated).
    if (bf && plan.size() > 1 && !getenv("MKP_NO_BED_MERGE")) {
      std::vector<ShardPlan> merged; uint64_t bytes = 0;
      for (auto& sp : plan) {
        const uint32_t tid = records[sp.rec].tid;
        const uint64_t sp_bytes = bam.indexed() ? bam.offset_at(tid, sp.s1) - bam.offset_at(tid, sp.s0) + (1u << 16) : 0;
        const bool join = !merged.empty() && records[merged.back().rec].tid == tid && sp.s0 >= merged.back().s1
            && (uint64_t)sp.s1 - merged.back().s0 <= shard_bp
            && (bytes + sp_bytes <= shard_bytes || !ahead.empty() /* its spans were sized when it was ingested ahead */) &&
                          merged.back().parts.size() < 65536;
        if (!join) { ShardPlan m = sp; m.parts.assign(1, {sp.rec, sp.s0, sp.s1}); merged.push_back(std::move(m)); bytes = sp_bytes; continue; }
        ShardPlan& m = merged.back(); m.s1 = sp.s1; m.bp += sp.bp; m.iv_starts.insert(m.iv_starts.end(), sp.iv_starts.begin(), sp.iv_starts.end());
          m.parts.push_back({sp.rec, sp.s0, sp.s1}); bytes += sp_bytes;
      }
      for (auto& m : merged) if (m.parts.size() == 1) m.parts.clear();
      plan.swap(merged);
    }
    mark("shard plan done");
  };
  // the fetch windows of a shard: its window — or, under --include-bed, the BED spans inside its pieces (rows exist at BED positions only;
  // the records of optimize_reference_records run from one span to the next, most of what lies under them is never looked at)
  auto plan_windows = [&](const ShardPlan& sp) {
    std::vector<std::pair<uint32_t, uint32_t>> w;
    if (!bf) { w.push_back({sp.s0, sp.s1}); return w; }
    const uint32_t tid = records[sp.rec].tid; std::vector<Span> spn;
    auto clip = [&](uint32_t a0, uint32_t a1) { for (auto* m : {&bf->pos, &bf->neg}) { auto it = m->find(tid); if (it == m->end()) continue;
        for (auto& x : it->second) { const uint64_t lo = std::max<uint64_t>(x.s, a0), hi = std::min<uint64_t>(x.e, a1);
          if (lo < hi) spn.push_back({lo, hi});
          } } };
    if (sp.parts.empty()) clip(sp.s0, sp.s1); else for (auto& pt : sp.parts) clip(pt.s0, pt.s1);
    merge_spans(spn); for (auto& x : spn) w.push_back({(uint32_t)x.s, (uint32_t)x.e});
    if (w.empty()) w.push_back({sp.s0, sp.s0 + 1});   // (no BED position inside: nothing to fetch but an empty window)
    return w;
  };
  auto fetch_shard = [&](const ShardPlan& sp) { return fetch_windows(records[sp.rec].tid, plan_windows(sp)); };
  // the plan's shards as the list ingested ahead (several ranks, contigs larger than a shard, full-data sampling from the shards): same
  // workers, same budget as the contig list above
  auto ahead_from_plan = [&]() {
    if (!dev_ingest || !ahead.empty() || plan.empty()) return;
    for (auto& sp : plan) { Ahead A; A.rec0 = sp.rec; A.rec1 = sp.rec + 1; A.tid = records[sp.rec].tid; A.s0 = sp.s0; A.s1 = sp.s1;
      A.wins = plan_windows(sp);
      A.bytes = bf ? win_bytes(A.tid, A.wins) : bam.offset_at(A.tid, A.s1) - bam.offset_at(A.tid, A.s0) + (1u << 16); ahead.push_back(std::move(A)); }
    start_ahead();
  };
  // thresholds (subcommand.rs:615-638)
  kc.per_mod = per_mod.data(); kc.n_per_mod = (uint32_t)per_mod.size();
  double thr_ms = 0, fetch_wait_early_ms = 0, grid_wait_ms = 0, callback_ms = 0;
  // the first shard's records, taken before the loop
  ShardInput early_in; bool early_in_ready = false, pre_attached = false; std::unique_ptr<DevShard> pre_dev;
  const bool full_mode = a.have_frac && a.sampling_frac >= 1.0;
  const bool want_estimate = a.filter_threshold.empty() && !a.no_filtering && !a.plan_only;
  const bool resident_ok = !have_sregion && !a.include_unmapped && !getenv("MKP_NO_RESIDENT_SAMPLING");
  const bool ahead_whole_contigs = !ahead.empty();   // (the list made above holds whole contigs; the plan's shards may be pieces of them)
  bool resident_used = pre_attached;
  // The plan comes before the thresholds — and its shards go ahead — when nothing is gained by waiting for them: several ranks (a rank needs
  // its own shards whatever the thresholds turn out to be, and the others' estimate or all-reduce is time to ingest in), thresholds given,
  // or a full-data estimate, which samples from the shards themselves.
  if (dev_ingest && !early_whole && ahead.empty() && !getenv("MKP_NO_AHEAD") && (a.world > 1 || !want_estimate || (full_mode && resident_ok))) {
    build_plan(); ahead_from_plan(); }
  if (!a.filter_threshold.empty()) parse_base_thresholds(a.filter_threshold, &kc);
  else if (a.no_filtering || a.plan_only) { kc.n_per_mod = 0; }  // MultipleThresholdModCaller::new_passthrough
  else {
    auto t0 = std::chrono::steady_clock::now();
    must(mkp_set_caller(ctx, &kc));  // collapse + edge filter apply to the sampled probabilities too
    const RegionSpec* sr = have_sregion ? &sregion : (have_region ? &region : nullptr);
    // a rank of a multi-GPU run that was not handed thresholds (--filter-threshold from the all-reduced histograms, see
    // modkit_amd.distributed) estimates them alone over the whole file: correct, but every rank repeats the work
    Args as = a; as.world = 1; as.rank = 0;
    must(mkp_histogram_begin(ctx));
    g_sample_times = SampleTimes();
    // mkp_pileup_run_cb: in the count-based modes the caller supplies the thresholds (one rank estimates, all ranks receive: the schedule
    // carries quotas from interval to interval and does not shard) — nothing is sampled here, and this rank's shards are being ingested
    // meanwhile; in the full-data mode this rank samples its own shards and the caller reduces the histograms over the ranks
    const bool cb_supplies = a.thr_cb && !full_mode;
    if (!cb_supplies) {
    // The one shard of the run is being ingested on the device and every read the schedule can ask for lies in it: the estimate waits for
    // it and samples from HBM (the heads are scans of the shard's digest, a round is one launch over reads that are already packed)
    // instead of fetching, inflating and packing interval heads on the host next to the ingest.
    const ShardHost* resident = nullptr;
    if (early_whole && !have_sregion && !a.include_unmapped && !getenv("MKP_NO_RESIDENT_SAMPLING")) {
      bool only_this = true; { const IdxStats stx = idxstats(bam, sr, bf);
        for (auto& kv : stx.mapped_by_tid) if (kv.second && kv.first != (int64_t)records[0].tid) only_this = false;
        }
      if (only_this) {
        mark("resident sampling: waiting for the grid");
        auto t_gw = std::chrono::steady_clock::now();
        if (early_walk.valid()) early_walk.get();
        else if (!grid_done[0]) { auto t_focus = std::chrono::steady_clock::now();
          grid_of[0] = fb.walk(records[0], a.interval_size, fb.has_focus() ? &focus_of[0] : nullptr); grid_done[0] = 1;
          if (fb.has_focus()) focus_done[0] = 1; focus_ms += ms_since(t_focus); }
        grid_wait_ms += ms_since(t_gw);
        const std::vector<Interval>& ivs = grid_of[0]; uint64_t bp = 0;
        if (!ivs.empty() && shard_cut(records[0], ivs, 0, &bp) == ivs.size() && ivs.front().start == early_s0 && ivs.back().end == early_s1) {
          // the shard is begun and everything of its plan that needs the window alone (slot bitmap, slot positions, their uploads) is
          // made now, while the ingest is still running
          mkp_shard sh; memset(&sh, 0, sizeof(sh)); sh.tid = (int32_t)records[0].tid; sh.start = early_s0; sh.end = early_s1;
          if (fb.has_focus()) { sh.focus = focus_of[0].data() + (early_s0 - records[0].start); sh.combos = fb.combos.data();
            sh.n_combos = (uint32_t)fb.combos.size(); }
          must(mkp_shard_begin(ctx, &sh));
          { std::vector<uint32_t> st; st.reserve(ivs.size()); for (auto& iv : ivs) st.push_back(iv.start);
            must(mkp_shard_set_intervals(ctx, st.data(), (uint32_t)st.size())); }
          if (fb.has_focus() && !a.hemi) must(mkp_internal_shard_preplan(ctx));
          auto t_w = std::chrono::steady_clock::now();
          mark("resident sampling: waiting for the ingest");
          early_in = early_fetch.get(); early_in_ready = true;
          fetch_wait_early_ms = ms_since(t_w);
          mark("resident sampling: ingest in hand");
          if (early_in.dev) {
            pre_dev = std::move(early_in.dev); early_in_ready = false;
            must(mkp_internal_shard_attach(ctx, pre_dev.get())); mkp_internal_ingest_recycle(ctx->ingest, pre_dev.get());
            pre_attached = true; resident = &ctx->shard;
            mark("first shard attached (resident sampling)");
          }
        }
      }
    }
    ResidentOf resident_of;
    DevShard* bound = nullptr;   // multi-shard resident sampling: the shard currently swapped into the context
    struct Unbind { mkp_ctx* c; DevShard** b; ~Unbind() { if (*b) { (void)mkp_internal_sample_bind(c, *b); *b = nullptr; } } } unbind{ctx, &bound};
    auto bind_ahead = [&](size_t k) -> const ShardHost* {
      auto t_w = std::chrono::steady_clock::now();
      Ahead& A = ahead_wait(k);
      fetch_wait_early_ms += ms_since(t_w);
      if (!A.in.dev) throw Error(MKP_E_INVALID, "internal: shard ingested ahead without device records");
      if (bound != A.in.dev.get()) { if (bound) must(mkp_internal_sample_bind(ctx, bound)); bound = nullptr;
        must(mkp_internal_sample_bind(ctx, A.in.dev.get())); bound = A.in.dev.get(); }
      return &ctx->shard;
    };
    // every shard stays in HBM until the loop takes it: the sample can come from them
    const bool ahead_fits = !ahead.empty() && ahead_est_total <= hbm_budget;
    // the extent the sampler covers on a contig: the region, or all of it
    auto ext_of = [&](uint32_t tid, int64_t* lo, int64_t* hi) { if (have_region) { *lo = region.start; *hi = region.end; } else { *lo = 0;
        *hi = bam.ref_lens[tid]; } };
    // Several ranks in the full-data mode (mkp_pileup_run_cb): the ranks' samples are summed by the caller, so every rank must cut the reads the
    // same way — by shard ownership (own_from) when the shards are the source, by sampling intervals when the host reader is.  The two cuts do
    // not coincide, so the choice is one every rank arrives at alike without talking: the shards are the source iff EVERY rank's shards fit
    // its budget (est_by_rank: the same arithmetic on the same index on every rank; the budget and MKP_NO_AHEAD must be the ranks' common
    // setting, like the flags).  A rank without shards then contributes an empty sample (it used to sample its bp-share of the
    // sampling intervals, reads the other ranks' shards had already counted).
    bool all_ranks_resident = dev_ingest && plan_built && !getenv("MKP_NO_AHEAD") && !early_whole;
    for (uint64_t e : est_by_rank) if (e > hbm_budget) all_ranks_resident = false;
    const bool shards_are_the_sample = a.world > 1 ? (a.thr_cb && all_ranks_resident) : ahead_fits;
    if (full_mode && resident_ok && (resident || shards_are_the_sample)) {
      // full-data mode: the shards themselves are the sample's source, whatever their cut — with several ranks (mkp_pileup_run_cb) each
      // rank its own, the caller sums the histograms
      std::vector<FullShard> fs;
      if (resident) { FullShard f; f.tid = records[0].tid; f.own_from = INT64_MIN; ext_of(f.tid, &f.ext_lo, &f.ext_hi); f.bind = [&]() {
          return resident; }; fs.push_back(std::move(f)); }
      else for (size_t k = 0; k < ahead.size(); k++) { FullShard f; f.tid = ahead[k].tid;
        f.own_from = ahead_whole_contigs ? INT64_MIN : plan[k].own_from; ext_of(f.tid, &f.ext_lo, &f.ext_hi);
          f.bind = [&, k]() { return bind_ahead(k); }; fs.push_back(std::move(f)); }
      sample_resident_full(ctx, bam, bf, fs);
      resident_used = true;
      if (a.stats) fprintf(stderr, "[mkpileup] full-data threshold sample taken from %zu resident shard(s)%s\n", fs.size(),
          a.world > 1 ? " of this rank" : "");
    } else {
    if (resident) resident_of = [&](uint32_t) { return resident; };
    else if (ahead_fits && ahead_whole_contigs && resident_ok && a.world == 1) {
      // every contig the schedule can visit must be among the shards ingested ahead
      std::map<uint32_t, size_t> by_tid; for (size_t k = 0; k < ahead.size(); k++) by_tid[ahead[k].tid] = k;
      bool covered = true; { const IdxStats stx = idxstats(bam, sr, bf);
        for (auto& kv : stx.mapped_by_tid) if (kv.second && !by_tid.count((uint32_t)kv.first)) covered = false;
        }
      // (a region run: the one record is the region; a BED run: a contig's shard holds the records reaching its BED spans, which are the only ones
      // that can yield a value)
      if (covered) resident_of = [&, by_tid](uint32_t tid) -> const ShardHost* {
        auto it = by_tid.find(tid); if (it == by_tid.end()) throw Error(MKP_E_INVALID, "internal: resident sampling outside the ingested contigs");
        return bind_ahead(it->second);
      };
    }
    // (a rank of mkp_pileup_run_cb's full-data mode whose shards are not resident — an unindexed BAM, a budget too small — samples its own
    // sampling intervals through the host reader, as round 4 did)
    sample_probabilities(ctx, bam, (a.thr_cb && full_mode) ? a : as, sr, bf, resident_of);
    if (resident_of) resident_used = true;
    if (a.stats && resident_of && !resident) fprintf(stderr, "[mkpileup] threshold estimate sampled from %zu shards ingested ahead (resident)\n",
        ahead.size());
    }
    if (bound) { must(mkp_internal_sample_bind(ctx, bound)); bound = nullptr; }
    mark("schedule walked, sample in HBM");
    if (a.stats) fprintf(stderr,
        "[mkpileup] threshold sampling: head fetch wait %.1f ms, device rounds %llu (%llu reads) %.1f ms, first-N logic %.1f ms\n",
        g_sample_times.fetch_ms, (unsigned long long)g_sample_times.rounds, (unsigned long long)g_sample_times.reads, g_sample_times.device_ms,
        g_sample_times.decide_ms);
    }
    { float thr[4] = {0, 0, 0, 0}; uint8_t has[4] = {0, 0, 0, 0};
      if (a.thr_cb) { auto t_cb = std::chrono::steady_clock::now(); const int rc = a.thr_cb(a.thr_cb_user, ctx, cb_supplies ? 0 : 1, thr, has);
        callback_ms = ms_since(t_cb);
        if (rc != MKP_OK) throw Error(rc, "the threshold callback failed"); mark("thresholds from the callback"); }
      else thresholds_from_sample(ctx, a.filter_percentile, thr, has, a.stats);
      for (int b = 0; b < 4; b++) if (has[b]) { kc.has_per_base[b] = 1; kc.per_base_threshold[b] = thr[b]; } }
    thr_ms = ms_since(t0) - fetch_wait_early_ms - grid_wait_ms - callback_ms;
  }
  mark("thresholds done");
  if (!a.plan_only) must(mkp_set_caller(ctx, &kc));
  if (early_walk.valid()) early_walk.get();   // rethrows what the walk threw
  // --partition-tag: the output path is a directory with one bedMethyl per key, `[<prefix>_]<key>.bed` (PartitioningBedMethylWriter,
  // writers.rs:1005-1082)
  const bool partitioned = !a.partition_tags.empty();
  std::map<std::string, std::unique_ptr<RowWriter>> key_writers;
  BedGraphOut bg;
  if (a.bedgraph) {   // (subcommand.rs:328-363: no header, no mixed delimiters; the path is a directory)
    if (a.with_header || a.mixed_delim || a.bgzf || a.hemi || a.plan_only) throw Error(MKP_E_INVALID,
        "--bedgraph cannot be combined with --with-header, --mixed-delim, --bgzf or --plan-only");
    if (a.out_bed.empty() || a.out_bed == "-" || a.out_bed == "stdout") throw Error(MKP_E_INVALID, "--bedgraph needs an output directory");
    if (partitioned) { std::vector<const char*> tp; for (auto& t : a.partition_tags) tp.push_back(t.c_str());
      must(mkp_set_partition_tags(ctx, tp.data(), (uint32_t)tp.size())); }
    if (mkdir(a.out_bed.c_str(), 0777) != 0 && errno != EEXIST) throw Error(MKP_E_IO, "failed to make output directory " + a.out_bed);
    bg.dir = a.out_bed; bg.prefix = a.prefix; bg.groupings = partitioned; bg.labels = wr.labels;
  } else if (partitioned) {
    if (a.with_header) throw Error(MKP_E_INVALID, "--with-header cannot be combined with --partition-tag");
    if (a.plan_only) throw Error(MKP_E_INVALID, "--plan-only has no partitioned form");
    { std::vector<const char*> tp; for (auto& t : a.partition_tags) tp.push_back(t.c_str());
      must(mkp_set_partition_tags(ctx, tp.data(), (uint32_t)tp.size())); }
    if (mkdir(a.out_bed.c_str(), 0777) != 0 && errno != EEXIST) throw Error(MKP_E_IO, "failed to make output directory " + a.out_bed);
  } else {
  wr.f = (a.out_bed == "-" || a.out_bed == "stdout" || (a.hemi && a.out_bed.empty())) ? stdout : fopen(a.out_bed.c_str(), "w+");
  if (!wr.f) throw Error(MKP_E_IO, "failed to make output file " + a.out_bed);
  if (a.bgzf) {
    if (wr.f == stdout || a.with_header || a.plan_only) throw Error(MKP_E_INVALID,
        "--bgzf writes a file and its index: it needs an output path and goes without --with-header");
    wr.bz.reset(new BgzfTabixSink()); wr.bz->f = wr.f; wr.bz->index_path = a.out_bed + ".tbi";
  }
  }
  if (a.bgzf && partitioned) throw Error(MKP_E_INVALID, "--bgzf has no partitioned form");
  auto writer_for = [&](const std::string& key) -> RowWriter& {
    auto it = key_writers.find(key);
    if (it == key_writers.end()) {
      std::unique_ptr<RowWriter> w(new RowWriter()); w->mixed = a.mixed_delim; w->labels = wr.labels;
      const std::string path = a.out_bed + "/" + (a.prefix.empty() ? key : a.prefix + "_" + key) + ".bed";
      w->f = fopen(path.c_str(), "w+"); if (!w->f) throw Error(MKP_E_IO, "failed to make output file " + path);
      it = key_writers.emplace(key, std::move(w)).first;
    }
    return *it->second;
  };
  if (a.with_header) fputs("chrom\tchromStart\tchromEnd\tname\tscore\tstrand\tthickStart\tthickEnd\tcolor\tvalid_coverage\tpercent_modified\tcount_modified\tcount_canonical\tcount_other_mod\tcount_delete\tcount_fail\tcount_diff\tcount_nocall\n",
      wr.f);
  build_plan();
  // (thresholds known before the plan: the shards go ahead of the loop from here)
  if (ahead.empty() && plan.size() > 1 && !getenv("MKP_NO_AHEAD")) ahead_from_plan();
  // shards that were ingested ahead: same contig, same hull, same windows
  std::vector<long> plan_ahead(plan.size(), -1);
  for (size_t pi = 0; pi < plan.size(); pi++) for (size_t k = 0; k < ahead.size(); k++)
    if (ahead[k].tid == records[plan[pi].rec].tid && ahead[k].s0 == plan[pi].s0 && ahead[k].s1 == plan[pi].s1
        && ahead[k].wins == plan_windows(plan[pi])) {
      plan_ahead[pi] = (long)k; break; }
  // double buffering: the next shard's blocks are read and inflated while this one is packed, run and written
  std::future<ShardInput> next_batch;
  const bool early_match = early_set && !plan.empty() && plan[0].rec == 0 && plan[0].s0 == early_s0 && plan[0].s1 == early_s1;
  if (pre_attached && !(early_match && plan.size() == 1)) throw Error(MKP_E_INVALID,
      "internal: the shard attached for resident sampling is not the plan's");
  // (what the estimate waited for shards ingested ahead is load time too)
  if (pre_attached || !ahead.empty()) { fetch_wait_ms += fetch_wait_early_ms; }
  else if (early_match && early_in_ready) { fetch_wait_ms += fetch_wait_early_ms; next_batch = std::async(std::launch::deferred, [&]() {
      return std::move(early_in); }); }
  else if (early_match && early_fetch.valid()) next_batch = std::move(early_fetch);
  else { if (early_fetch.valid()) early_fetch.wait();
    if (!plan.empty() && plan_ahead[0] < 0) next_batch = std::async(std::launch::async, fetch_shard, plan[0]);
    }
  for (size_t pi = 0; pi < plan.size(); pi++) {
    const ShardPlan& sp = plan[pi]; const Contig& rec = records[sp.rec]; const uint32_t s0 = sp.s0, s1 = sp.s1; const uint64_t bp = sp.bp;
    std::unique_ptr<BamBatch> batch; std::unique_ptr<DevShard> dev;
    const bool attached_already = pre_attached && pi == 0;
    // the shard's bytes go back to the budget when this iteration is over
    struct Release { decltype(ahead_release)& rel; long k; ~Release() { if (k >= 0) rel((size_t)k); } } release_shard{ahead_release, plan_ahead[pi]};
    if (attached_already) dev = std::move(pre_dev);
    else if (plan_ahead[pi] >= 0) { auto t_f = std::chrono::steady_clock::now(); Ahead& A = ahead_wait((size_t)plan_ahead[pi]);
      batch = std::move(A.in.batch); dev = std::move(A.in.dev); fetch_wait_ms += ms_since(t_f); }
    else { auto t_f = std::chrono::steady_clock::now(); ShardInput in = next_batch.get(); batch = std::move(in.batch); dev = std::move(in.dev);
      fetch_wait_ms += ms_since(t_f); }
    if (dev) { ingest_ms[0] += dev->ms_plan; ingest_ms[1] += dev->ms_upload; ingest_ms[2] += dev->ms_inflate; ingest_ms[3] += dev->ms_pack;
      ingest_ms[4] += dev->ms_digest; ingest_blocks += dev->n_blocks;
               ingest_records += dev->n_records; ingest_kernel_ms += dev->ms_kernel; ingest_comp += dev->comp_bytes; ingest_raw += dev->raw_bytes; }
    if (pi + 1 < plan.size() && plan_ahead[pi + 1] < 0) next_batch = std::async(std::launch::async, fetch_shard, plan[pi + 1]);
    std::vector<uint8_t> merged_focus;   // a merged shard: its records' focus bytes at their places in the hull, zero in between
    if (hf) for (size_t k = 0; k < std::max<size_t>(sp.parts.size(), 1); k++) { const size_t ri = sp.parts.empty() ? sp.rec : sp.parts[k].rec;
      if (!focus_done[ri]) { auto t_focus = std::chrono::steady_clock::now(); fb.walk(records[ri], a.interval_size, &focus_of[ri]);
        focus_done[ri] = 1; focus_ms += ms_since(t_focus); } }
    if (sp.rec > 0 && (pi == 0 || plan[pi - 1].rec != sp.rec)) for (size_t r2 = 0; r2 < sp.rec; r2++) { std::vector<uint8_t>().swap(focus_of[r2]);
        }   // earlier contigs are done
    if (hf && !sp.parts.empty()) { merged_focus.assign((size_t)(s1 - s0), 0);
      for (auto& pt : sp.parts) memcpy(merged_focus.data() + (pt.s0 - s0), focus_of[pt.rec].data() + (pt.s0 - records[pt.rec].start),
          (size_t)(pt.s1 - pt.s0));
        }
    const std::vector<uint8_t>& focus = focus_of[sp.rec];
    std::vector<mkp_record> recs; if (batch) { recs.reserve(batch->recs.size()); for (auto& e : batch->recs) recs.push_back(batch->view(e)); }
    {
      if (a.plan_only) {  // host-only dry run: the shard plan, plus the packer over the shard's records (no device)
        static std::vector<ShardHost> kept_pieces;   // the all-cores pack's per-thread buffers persist across shards, as they do in a context
        Packer pk; pk.pieces.swap(kept_pieces); ShardHost S; S.tid = (int32_t)rec.tid; S.win_start = (int32_t)s0; S.win_end = (int32_t)s1;
        const int32_t tid = (int32_t)rec.tid; auto t_pk = std::chrono::steady_clock::now();
        // --plan-pack-min N: records from which the packer runs on all cores (0 = always; default as in mkp_shard_add_records)
        pack_records(pk, S, recs.data(), (uint32_t)recs.size(), [tid](const mkp_record& r) { return r.tid == tid && Packer::keep(r);
          }, a.plan_pack_min);
        pack_ms += ms_since(t_pk); kept_pieces.swap(pk.pieces);
        // digest of everything the packer hands to the device: a parallel pack must equal the sequential one byte for byte
        uint64_t dg = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; i++) { dg ^= b[i];
            dg *= 1099511628211ull; } };
        mix(S.hdr.data(), S.hdr.size() * sizeof(MkpReadHdr)); mix(S.cigar.data(), S.cigar.size() * 4);
          mix(S.chunk_pfx.data(), S.chunk_pfx.size() * 4);
            mix(S.seq.data(), S.seq.size());
        mix(S.tagref.data(), S.tagref.size() * sizeof(MkpTagRef)); mix(S.ranks.data(), S.ranks.size() * 4); mix(S.ml.data(), S.ml.size());
            mix(S.name_hash.data(), S.name_hash.size() * 8);
        for (auto& k : pk.layout_keys) mix(k.data(), k.size());
        fprintf(wr.f, "%s\t%u\t%u\t%zu\t%llu\t%016llx\n", rec.name.c_str(), s0, s1, S.hdr.size(), (unsigned long long)S.n_calls,
            (unsigned long long)dg);
          positions += bp;
            continue;
      }
      mkp_shard sh; memset(&sh, 0, sizeof(sh)); sh.tid = (int32_t)rec.tid; sh.start = s0; sh.end = s1;
      if (hf) { sh.focus = sp.parts.empty() ? focus.data() + (s0 - rec.start) : merged_focus.data(); sh.combos = fb.combos.data();
        sh.n_combos = (uint32_t)fb.combos.size(); }
      mark("shard blocks in hand");
      if (attached_already) dev.reset();   // begun and attached before the threshold estimate, which sampled from it
      else {
        must(mkp_shard_begin(ctx, &sh));
        must(mkp_shard_set_intervals(ctx, sp.iv_starts.data(), (uint32_t)sp.iv_starts.size()));
        if (dev) { must(mkp_internal_shard_attach(ctx, dev.get())); mkp_internal_ingest_recycle(ctx->ingest, dev.get()); dev.reset(); }
        else must(mkp_shard_add_records(ctx, recs.data(), (uint32_t)recs.size()));
      }
      mark("shard packed");
      batch.reset();   // packed: the inflated blocks are no longer needed (their mappings are parked for the next fetch, ByteBuf::spares)
      mkp_rows rows; memset(&rows, 0, sizeof(rows));
      if (a.hemi) {
        int hoff = 0; fb.motifs[0].neg_delta(&hoff);
        mkp_hemi_rows hrows; must(mkp_hemi_shard_run(ctx, hoff, sp.iv_starts.data(), (uint32_t)sp.iv_starts.size(), &hrows));
        if (a.rerun) must(mkp_shard_rerun(ctx, a.rerun, nullptr));
        auto t_w = std::chrono::steady_clock::now();
        wr.write_hemi(rec.name, hrows);
        write_ms += ms_since(t_w);
        rows.processed_records = hrows.processed_records; rows.skipped_records = hrows.skipped_records;
      } else {
      must(mkp_shard_run(ctx, &rows));
      mark("  mkp_shard_run returned");
      if (a.rerun) must(mkp_shard_rerun(ctx, a.rerun, &rows));   // measurement aid: warm, averaged kernel times in --stats
      { auto t_w = std::chrono::steady_clock::now();
        if (a.bedgraph) { bg.write(rec.name, rows); wr.n += rows.n_rows; }
        else if (!partitioned) wr.write(rec.name, rows);
        else for (uint64_t i0r = 0; i0r < rows.n_rows;) {   // rows come grouped by key: one slice per key
          uint64_t i1r = i0r; while (i1r < rows.n_rows && rows.partition_key[i1r] == rows.partition_key[i0r]) i1r++;
          mkp_rows v = rows; v.n_rows = i1r - i0r; v.pos += i0r; v.strand += i0r; v.code_repr += i0r; v.motif_idx += i0r; v.n_valid += i0r;
            v.n_mod += i0r;
              v.n_canonical += i0r; v.n_other += i0r;
          v.n_delete += i0r; v.n_fail += i0r; v.n_diff += i0r; v.n_nocall += i0r; v.partition_key += i0r;
          const uint32_t k = rows.partition_key[i0r];
          RowWriter& kw = writer_for(k < rows.n_partition_keys ? rows.partition_key_names[k] : "not_found"); kw.write(rec.name, v); wr.n += v.n_rows;
          i0r = i1r;
        }
        write_ms += ms_since(t_w); }
      }
      n_shards++;
      mark("shard run, rows with the writer");
      mkp_stats st; mkp_get_stats(ctx, &st); kernel_ms += st.kernel_ms; dec_ms += st.decode_kernel_ms; pil_ms += st.pileup_kernel_ms;
        row_ms += st.rows_kernel_ms;
          pack_ms += st.pack_ms; h2d_ms += st.h2d_ms; d2h_ms += st.d2h_ms;
      positions += bp; processed += rows.processed_records; skipped += rows.skipped_records;
    }
  }
  load_ms += fetch_wait_ms;   // what the shard loop waited for blocks to be read and inflated (the rest overlapped with pack / run / write)
  { auto t_w = std::chrono::steady_clock::now(); bg.finish(); wr.finish(); for (auto& kv : key_writers) { kv.second->finish(); fclose(kv.second->f);
    } write_ms += ms_since(t_w); }
  if (wr.f && wr.f != stdout) fclose(wr.f);
  mark("output closed");
  if (rep) {
    memset(rep, 0, sizeof(*rep));
    rep->load_ms = load_ms; rep->threshold_ms = thr_ms; rep->focus_ms = focus_ms; rep->pack_ms = pack_ms; rep->h2d_ms = h2d_ms;
      rep->kernel_ms = kernel_ms;
        rep->d2h_ms = d2h_ms; rep->write_ms = write_ms;
    rep->total_ms = ms_since(t_all); rep->n_rows = wr.n; rep->n_positions = positions; rep->n_shards = n_shards; rep->processed_records = processed;
        rep->skipped_records = skipped;
    for (int b = 0; b < 4; b++) { rep->threshold[b] = kc.per_base_threshold[b]; rep->has_threshold[b] = kc.has_per_base[b]; }
    rep->grid_wait_ms = grid_wait_ms; rep->callback_ms = callback_ms; rep->ingest_kernel_ms = ingest_kernel_ms; rep->ingest_upload_ms = ingest_ms[1];
      rep->ingest_table_ms = ingest_ms[0];
    rep->ingest_pack_ms = ingest_ms[3]; rep->ingest_comp_bytes = ingest_comp; rep->ingest_raw_bytes = ingest_raw; rep->ingest_blocks = ingest_blocks;
      rep->ingest_records = ingest_records;
  }
  if (a.stats) fprintf(stderr,
      "[mkpileup] rows=%llu positions=%llu processed~%llu skipped~%llu load_ms=%.1f threshold_ms=%.1f focus_ms=%.1f pack_ms=%.1f h2d_ms=%.1f kernel_ms=%.3f (decode %.3f pileup %.3f rows %.3f) d2h_ms=%.1f write_ms=%.1f total_ms=%.1f shards=%llu indexed=%d bam_bytes_read=%llu bam_bytes_inflated=%llu (on the device %llu) peak_rss_kb=%llu\n",
                       (unsigned long long)wr.n, (unsigned long long)positions, (unsigned long long)processed, (unsigned long long)skipped, load_ms,
                           thr_ms, focus_ms, pack_ms, h2d_ms, kernel_ms, dec_ms, pil_ms, row_ms, d2h_ms, write_ms, ms_since(t_all),
                       (unsigned long long)n_shards, bam.indexed() ? 1 : 0, (unsigned long long)bam.bytes_read.load(),
                           (unsigned long long)bam.bytes_inflated.load(), (unsigned long long)bam.bytes_inflated_device.load(),
                           (unsigned long long)peak_rss_kb());
  if (a.stats) {   // every MKP_* variable that is set: they pick kernels and paths and would otherwise leave no trace in a measurement
    std::string ov; for (char** e = ::environ; e && *e; e++) if (!strncmp(*e, "MKP_", 4)) { ov += ' '; ov += *e; }
    fprintf(stderr,
        "[mkpileup] ingest=%s resident_sampling=%d ahead=%zu shards (estimated %.0f MB, HBM budget %.0f MB) rank=%u/%u env overrides:%s\n",
        dev_ingest ? "device" : "host", (pre_attached || resident_used) ? 1 : 0,
        ahead.size(), (double)ahead_est_total / 1048576.0, (double)hbm_budget / 1048576.0, a.rank, a.world, ov.empty() ? " none" : ov.c_str());
  }
  if (a.stats
      && dev_ingest) fprintf(stderr,
      "[mkpileup] device ingest: %llu BGZF blocks, %llu records; block plan %.1f ms, upload %.1f, inflate + CRC + chains %.1f, parse + pack %.1f, digest %.1f (overlapped with the threshold estimate / the shard in hand)\n",
      (unsigned long long)ingest_blocks, (unsigned long long)ingest_records, ingest_ms[0], ingest_ms[1], ingest_ms[2], ingest_ms[3], ingest_ms[4]);
  return MKP_OK;
}

void parse_args(int argc, const char* const* argv, Args* out, bool need_positional, bool hemi = false) {
  Args& a = *out; std::vector<std::string> pos;
  a.hemi = hemi;
  for (int i = 0; i < argc; i++) {
    std::string s = argv[i];
    auto val = [&]() { if (i + 1 >= argc) throw Error(MKP_E_INVALID, "missing value for " + s); return std::string(argv[++i]); };
    if (hemi && (s == "--preset" || s == "--combine-strands" || s == "--with-header" || s == "--header" || s == "--partition-tag" || s == "--prefix"
        || s == "--bedgraph" || s == "--plan-only"))
      throw Error(MKP_E_INVALID, "unexpected argument '" + s + "' for pileup-hemi");
    if (hemi && (s == "-o" || s == "--out-bed")) { a.out_bed = val(); continue; }
    if (s == "--region") a.region = val(); else if (s == "--max-depth") a.max_depth = (uint32_t)std::stoul(val());
    else if (s == "-t" || s == "--threads") a.threads = std::stoul(val());
      else if (s == "-i" || s == "--interval-size") a.interval_size = (uint32_t)std::stoul(val());
    else if (s == "--chunk-size" || s == "--queue-size" || s == "--log-filepath") val();
    else if (s == "--seed") { a.have_seed = true; a.seed = std::stoull(val()); }
    else if (s == "-n" || s == "--num-reads") a.num_reads = std::stoul(val()); else if (s == "-f" || s == "--sampling-frac") { a.have_frac = true;
        a.sampling_frac = std::stod(val()); }
    else if (s == "--no-filtering") a.no_filtering = true; else if (s == "-p" || s == "--filter-percentile") a.filter_percentile = std::stof(val());
    else if (s == "--filter-threshold") a.filter_threshold.push_back(val());
        else if (s == "--mod-thresholds" || s == "--mod-threshold") a.mod_thresholds.push_back(val());
    else if (s == "--sample-region") a.sample_region = val();
      else if (s == "--sampling-interval-size") a.sampling_interval_size = (uint32_t)std::stoul(val());
    else if (s == "--include-bed" || s == "--include-positions") a.include_bed = val(); else if (s == "--include-unmapped") a.include_unmapped = true;
    else if (s == "--ignore") a.ignore = val(); else if (s == "--force-allow-implicit") a.force_allow = true;
    else if (s == "--motif") { a.motif_parts.push_back(val()); a.motif_parts.push_back(val()); } else if (s == "--cpg") a.cpg = true;
    else if (s == "--ref" || s == "-r") a.ref_fasta = val(); else if (s == "--mask" || s == "-k") a.mask = true;
      else if (s == "--preset") a.preset = val();
    else if (s == "--combine-mods") a.combine_mods = true; else if (s == "--combine-strands") a.combine_strands = true;
    else if (s == "--edge-filter") a.edge_filter = val(); else if (s == "--invert-edge-filter") a.invert_edge = true;
    else if (s == "--only-tabs" || s == "--suppress-progress") {} else if (s == "--mixed-delim") a.mixed_delim = true;
        else if (s == "--with-header" || s == "--header") a.with_header = true;
    else if (s == "--device") a.device = std::stoi(val()); else if (s == "--gpus-rank") a.rank = (uint32_t)std::stoul(val());
        else if (s == "--gpus-world") a.world = (uint32_t)std::stoul(val());
    else if (s == "--plan-only") a.plan_only = true; else if (s == "--plan-pack-min") a.plan_pack_min = (uint32_t)std::stoul(val());
        else if (s == "--rerun") a.rerun = (uint32_t)std::stoul(val()); else if (s == "--shard-bp") a.shard_bp = std::stoull(val());
        else if (s == "--shard-bytes") { a.shard_bytes = std::max<uint64_t>(1, std::stoull(val())); a.shard_bytes_set = true;
          } else if (s == "--no-index") a.no_index = true;
        else if (s == "--device-inflate") a.device_inflate = true; else if (s == "--host-ingest") a.host_ingest = true;
        else if (s == "--hbm-budget-mb") a.hbm_budget_mb = std::stoull(val());
        else if (s == "--tile") a.tile = (uint32_t)std::stoul(val()); else if (s == "--stats") a.stats = true;
    else if (s == "--partition-tag") a.partition_tags.push_back(val()); else if (s == "--prefix") a.prefix = val();
    else if (s == "--bgzf") a.bgzf = true;
    else if (s == "--bedgraph") a.bedgraph = true;
    else if (!s.empty() && s[0] == '-' && s != "-") throw Error(MKP_E_INVALID, "unknown flag " + s);
    else pos.push_back(s);
  }
  if (need_positional && hemi) { if (pos.size() != 1) throw Error(MKP_E_INVALID, "usage: <in.bam> -o <out.bed> [flags of `modkit pileup-hemi`]");
    a.in_bam = pos[0]; }
  else if (need_positional) { if (pos.size() != 2) throw Error(MKP_E_INVALID, "usage: <in.bam> <out.bed> [flags of `modkit pileup`]");
    a.in_bam = pos[0];
      a.out_bed = pos[1]; }
  else if (!pos.empty()) throw Error(MKP_E_INVALID, "unexpected positional argument " + pos[0]);
  if (a.world == 0 || a.rank >= a.world) throw Error(MKP_E_INVALID, "bad --gpus-rank/--gpus-world");
}

}  // namespace

// (tests) the --bedgraph writer alone: rows in, files out — no device involved
extern "C" int mkp_internal_bedgraph_write(const char* dir, const char* prefix, int groupings, const char* const* motif_labels, uint32_t n_labels,
    const char* chrom, const mkp_rows* rows) {
  if (!dir || !chrom || !rows) return MKP_E_INVALID;
  try {
    BedGraphOut bg; bg.dir = dir; bg.prefix = prefix ? prefix : ""; bg.groupings = groupings != 0;
    for (uint32_t k = 0; k < n_labels; k++) bg.labels.push_back(motif_labels[k]);
    bg.write(chrom, *rows); bg.finish();
    return MKP_OK;
  } catch (const Error& e) { return e.status; } catch (const std::exception&) { return MKP_E_INVALID; }
}

extern "C" int mkp_pileup_main(int argc, const char* const* argv, char* errbuf, size_t errbuf_len) {
  auto fail = [&](int st, const std::string& m) { if (errbuf && errbuf_len) { snprintf(errbuf, errbuf_len, "%s", m.c_str()); } return st; };
  try {
    Args a; parse_args(argc, argv, &a, true);
    return run(a, nullptr, nullptr);
  } catch (const Error& e) { return fail(e.status, e.what()); }
  catch (const std::exception& e) { return fail(MKP_E_INVALID, e.what()); }
}

extern "C" int mkp_pileup_hemi_main(int argc, const char* const* argv, char* errbuf, size_t errbuf_len) {
  auto fail = [&](int st, const std::string& m) { if (errbuf && errbuf_len) { snprintf(errbuf, errbuf_len, "%s", m.c_str()); } return st; };
  try {
    Args a; parse_args(argc, argv, &a, true, true);
    return run(a, nullptr, nullptr);
  } catch (const Error& e) { return fail(e.status, e.what()); }
  catch (const std::exception& e) { return fail(MKP_E_INVALID, e.what()); }
}

extern "C" int mkp_pileup_hemi_run(mkp_ctx* ctx, int argc, const char* const* argv, mkp_run_report* report) {
  if (!ctx) return MKP_E_INVALID;
  try {
    Args a; parse_args(argc, argv, &a, true, true);
    return run(a, ctx, report);
  } catch (const Error& e) { ctx->err = e.what(); return e.status; }
  catch (const std::exception& e) { ctx->err = e.what(); return MKP_E_INVALID; }
}

// The same subcommand on a context the caller owns: the last shard stays resident in HBM afterwards (mkp_shard_rerun
// re-launches the kernels on it) and the per-stage wall times come back in `report`.
extern "C" int mkp_pileup_run(mkp_ctx* ctx, int argc, const char* const* argv, mkp_run_report* report) {
  if (!ctx) return MKP_E_INVALID;
  try {
    Args a; parse_args(argc, argv, &a, true);
    if (a.plan_only) throw Error(MKP_E_INVALID, "--plan-only needs no context: use mkp_pileup_main");
    return run(a, ctx, report);
  } catch (const Error& e) { ctx->err = e.what(); return e.status; }
  catch (const std::exception& e) { ctx->err = e.what(); return MKP_E_INVALID; }
}

extern "C" int mkp_pileup_run_cb(mkp_ctx* ctx, int argc, const char* const* argv, mkp_threshold_fn fn, void* user, mkp_run_report* report) {
  if (!ctx) return MKP_E_INVALID;
  try {
    Args a; parse_args(argc, argv, &a, true);
    if (a.plan_only) throw Error(MKP_E_INVALID, "--plan-only needs no context: use mkp_pileup_main");
    a.thr_cb = fn; a.thr_cb_user = user;
    return run(a, ctx, report);
  } catch (const Error& e) { ctx->err = e.what(); return e.status; }
  catch (const std::exception& e) { ctx->err = e.what(); return MKP_E_INVALID; }
}

// get_threshold_from_options (command_utils.rs:74-134) as a call: per-base pass thresholds from the
// reference's sampling schedule; argv takes the sampling flags of `modkit pileup`
// (-n -f -p -t --sampling-interval-size --region --sample-region --include-bed --include-unmapped --edge-filter --ignore --preset).
namespace {
// the sampling half of get_threshold_from_options: parse the sampling flags, set the caller's collapse / edge filter, walk the schedule
void sample_bam(mkp_ctx* ctx, const char* bam_path, int argc, const char* const* argv, float* q_out, const mkp_caller* thresholds = nullptr,
    bool serial_without_index = false /* sample-probs / summary / extract calls: the reference's own choice (reads_sampler/mod.rs:47) */) {
  Args a; parse_args(argc, argv, &a, false); a.in_bam = bam_path;
  std::unique_ptr<BamSource> src = BamSource::open(a.in_bam,
      std::max(std::max<unsigned>((unsigned)a.threads, 1u), std::min(32u, HostPool::host_cpus())),
      !a.no_index);   // inflate threads: --threads only steers the sampling schedule
  const BamSource& bam = *src;
  RegionSpec region, sregion; const bool hr = !a.region.empty(), hs = !a.sample_region.empty();
  if (hr) region = parse_region(a.region, bam);
  if (hs) sregion = parse_region(a.sample_region, bam);
  if (!(a.filter_percentile >= 0.0f) || a.filter_percentile > 1.0f) throw Error(MKP_E_INVALID, "filter percentile must be in [0, 1]");
  mkp_caller kc; memset(&kc, 0, sizeof(kc)); kc.max_depth = a.max_depth;
  if (!a.edge_filter.empty()) { kc.edge_filter = 1; kc.edge_inverted = a.invert_edge; size_t c = a.edge_filter.find(',');
      if (c != std::string::npos) { kc.edge_start = (uint32_t)strtoul(a.edge_filter.c_str(), nullptr, 10);
      kc.edge_end = (uint32_t)strtoul(a.edge_filter.c_str() + c + 1, nullptr, 10);
        } else kc.edge_start = kc.edge_end = (uint32_t)strtoul(a.edge_filter.c_str(), nullptr,
      10); }
  if (a.preset == "traditional") { kc.numeric_mode = 2; kc.collapse_code = 'h'; }
  else if (!a.ignore.empty()) { uint32_t code; if (!parse_code(a.ignore, &code)) throw Error(MKP_E_INVALID, "failed to parse mod code " + a.ignore);
    kc.numeric_mode = 2;
      kc.collapse_code = code; }
  std::vector<Contig> records = targets(bam, hr ? &region : nullptr);
  BedFilter bed_store; const BedFilter* bf = nullptr;
  if (!a.include_bed.empty()) { std::map<std::string, uint32_t> c2t; for (auto& r : records) c2t[r.name] = r.tid;
    bed_store = BedFilter::load(a.include_bed, c2t);
      bf = &bed_store; }
  if (thresholds) { kc.default_threshold = thresholds->default_threshold; kc.per_mod = thresholds->per_mod; kc.n_per_mod = thresholds->n_per_mod;
                    for (int b = 0; b < 4; b++) { kc.per_base_threshold[b] = thresholds->per_base_threshold[b];
                      kc.has_per_base[b] = thresholds->has_per_base[b]; } }
  int rc = mkp_set_caller(ctx, &kc); if (rc != MKP_OK) throw Error(rc, mkp_last_error(ctx));
  a.serial_sampler = serial_without_index && a.world <= 1 && !bam_index_file_exists(a.in_bam);
  sample_probabilities(ctx, bam, a, hs ? &sregion : (hr ? &region : nullptr), bf);
  if (q_out) *q_out = a.filter_percentile;
}
}  // namespace

// get_threshold_from_options (command_utils.rs:74-134) as a call: per-base pass thresholds from the
// reference's sampling schedule; argv takes the sampling flags of `modkit pileup`
// (-n -f -p -t --sampling-interval-size --region --sample-region --include-bed --include-unmapped --edge-filter --ignore --preset).
extern "C" int mkp_estimate_thresholds(mkp_ctx* ctx, const char* bam_path, int argc, const char* const* argv, float thr[4], uint8_t has[4]) {
  if (!ctx || !bam_path || !thr || !has) return MKP_E_INVALID;
  try {
    int rc = mkp_histogram_begin(ctx); if (rc != MKP_OK) return rc;
    float q = 0.1f; sample_bam(ctx, bam_path, argc, argv, &q);
    thresholds_from_sample(ctx, q, thr, has, false);
    return MKP_OK;
  } catch (const Error& e) { ctx->err = e.what(); return e.status; }
  catch (const std::exception& e) { ctx->err = e.what(); return MKP_E_INVALID; }
}

// The sampling half alone, for multi-GPU runs: add this rank's share of the sample (argv carries --gpus-rank R --gpus-world W
// next to the sampling flags; W > 1 needs -f 1.0) to the context's histograms.  The caller then sums mkp_histogram_get's arrays
// over the ranks (RCCL all-reduce) and finishes with mkp_histogram_locate / _resolve / mkp_percentile_from_histogram.
// `modkit sample-probs` (SampleModBaseProbs::run, src/commands.rs:680-887), the percentiles table: the schedule's sample, per canonical
// base the requested percentiles of the argmax probabilities (Percentiles::new -> percentile_linear_interp).  The subcommand's own
// flag names: -i is the sampling interval (default 1 000 000), unmapped reads are sampled unless --only-mapped (or --include-bed).
extern "C" int mkp_sample_probs(mkp_ctx* ctx, const char* bam_path, int argc, const char* const* argv, const float* percentiles,
    uint32_t n_percentiles,
                                float* values /* [4][n_percentiles], bases A,C,G,T */, uint8_t has[4], uint64_t n_values[4]) {
  if (!ctx || !bam_path || (!percentiles && n_percentiles) || !values || !has || !n_values) return MKP_E_INVALID;
  try {
    std::vector<std::string> tr; bool only_mapped = false, have_i = false;
    for (int i = 0; i < argc; i++) {
      const std::string s = argv[i];
      if (s == "-i" || s == "--interval-size") { tr.push_back("--sampling-interval-size"); have_i = true; }
      else if (s == "--no-sampling") { tr.push_back("-f"); tr.push_back("1.0"); }
      else if (s == "--only-mapped") only_mapped = true;
      else if (s == "--include-bed" || s == "--include-positions") { only_mapped = true; tr.push_back(s); }
      else if (s == "-p" || s == "--percentiles" || s == "--filter-percentile") throw Error(MKP_E_INVALID,
          "percentiles are an argument of this call, not a flag");
      else tr.push_back(s);
    }
    if (!have_i) { tr.push_back("--sampling-interval-size"); tr.push_back("1000000"); }
    if (!only_mapped) tr.push_back("--include-unmapped");
    std::vector<const char*> av; for (auto& x : tr) av.push_back(x.c_str());
    int rc = mkp_histogram_begin(ctx); if (rc != MKP_OK) return rc;
    sample_bam(ctx, bam_path, (int)av.size(), av.data(), nullptr, nullptr, true);
    for (int b = 0; b < 4; b++) { has[b] = 0; n_values[b] = 0; }
    for (uint32_t k = 0; k < n_percentiles; k++) {
      const float q = percentiles[k];
      if (!(q >= 0.0f) || q > 1.0f) throw Error(MKP_E_INVALID, "percentiles must be in [0, 1]");
      float thr[4]; uint8_t h[4]; uint64_t n[4] = {0, 0, 0, 0};
      thresholds_from_sample(ctx, q, thr, h, false, n);
      for (int b = 0; b < 4; b++) { values[(size_t)b * n_percentiles + k] = thr[b]; has[b] = h[b]; n_values[b] = n[b]; }
    }
    if (n_percentiles == 0) { float thr[4]; uint64_t n[4] = {0, 0, 0, 0}; try { thresholds_from_sample(ctx, 0.5f, thr, has, false, n);
        } catch (const Error&) {} for (int b = 0; b < 4; b++) n_values[b] = n[b]; }
    return MKP_OK;
  } catch (const Error& e) { ctx->err = e.what(); return e.status; }
  catch (const std::exception& e) { ctx->err = e.what(); return MKP_E_INVALID; }
}

// `modkit summary` (ModSummarize::run, src/commands.rs:1035-1189): ModSummary as counts.  Two walks of the same sampling schedule on the
// device: the first estimates the pass thresholds (calc_thresholds_per_base) unless the flags give them, the second counts every sampled
// call under its thresholded call, or under its argmax call when that is Filtered (sampled_reads_to_summary, src/summarize.rs:117-262).
extern "C" int mkp_summary(mkp_ctx* ctx, const char* bam_path, int argc, const char* const* argv, mkp_summary_out* out) {
  if (!ctx || !bam_path || !out) return MKP_E_INVALID;
  struct ModeGuard { mkp_ctx* c; ~ModeGuard() { c->summary_mode = false; } } guard{ctx};
  try {
    std::vector<std::string> tr; bool only_mapped = false, have_i = false;
    for (int i = 0; i < argc; i++) {
      const std::string s = argv[i];
      if (s == "-i" || s == "--interval-size") { tr.push_back("--sampling-interval-size"); have_i = true; }
      else if (s == "--no-sampling") { tr.push_back("-f"); tr.push_back("1.0"); }
      else if (s == "--only-mapped") only_mapped = true;
      else if (s == "--include-bed" || s == "--include-positions") { only_mapped = true; tr.push_back(s); }
      else if (s == "--tsv" || s == "--table") {}
      else tr.push_back(s);
    }
    if (!have_i) { tr.push_back("--sampling-interval-size"); tr.push_back("1000000"); }
    if (!only_mapped) tr.push_back("--include-unmapped");
    std::vector<const char*> av; for (auto& x : tr) av.push_back(x.c_str());
    Args a; parse_args((int)av.size(), av.data(), &a, false);
    mkp_caller kt; memset(&kt, 0, sizeof(kt));
    std::vector<mkp_mod_threshold> per_mod;
    for (auto& raw : a.mod_thresholds) { size_t c = raw.find(':'); uint32_t code;
        if (c == std::string::npos || raw.find(':', c + 1) != std::string::npos || !parse_code(raw.substr(0, c), &code)) throw Error(MKP_E_INVALID,
        "encountered illegal per-mod threshold: " + raw);
        per_mod.push_back({code, strtof(raw.c_str() + c + 1, nullptr)}); }
    if (!a.filter_threshold.empty()) { parse_base_thresholds(a.filter_threshold, &kt); kt.per_mod = per_mod.data();
      kt.n_per_mod = (uint32_t)per_mod.size(); }
    else if (a.no_filtering) { /* MultipleThresholdModCaller::new_passthrough */ }
    else {
      int rc = mkp_histogram_begin(ctx); if (rc != MKP_OK) return rc;
      float q = 0.1f; sample_bam(ctx, bam_path, (int)av.size(), av.data(), &q, nullptr, true);
      float thr[4]; uint8_t has[4]; thresholds_from_sample(ctx, q, thr, has, false);
      for (int b = 0; b < 4; b++) if (has[b]) { kt.has_per_base[b] = 1; kt.per_base_threshold[b] = thr[b]; }
      kt.per_mod = per_mod.data(); kt.n_per_mod = (uint32_t)per_mod.size();
    }
    ctx->summary_mode = true;
    int rc = mkp_internal_summary_begin(ctx); if (rc != MKP_OK) return rc;
    sample_bam(ctx, bam_path, (int)av.size(), av.data(), nullptr, &kt, true);
    uint64_t t[134]; std::vector<MkpSlot> slots;
    rc = mkp_internal_summary_get(ctx, t, &slots); if (rc != MKP_OK) return rc;
    ctx->h_sum_base.clear(); ctx->h_sum_code.clear(); ctx->h_sum_pass.clear(); ctx->h_sum_fail.clear();
    const uint64_t obs = t[128 + 5];
    for (uint32_t b = 0; b < 4; b++) {
      out->reads_with_mod_calls[b] = t[128 + b]; out->threshold[b] = kt.per_base_threshold[b]; out->has_threshold[b] = kt.has_per_base[b];
      if (!t[128 + b]) continue;
      auto row = [&](uint32_t code, uint32_t cls) { ctx->h_sum_base.push_back((uint8_t)b); ctx->h_sum_code.push_back(code);
        ctx->h_sum_pass.push_back(t[b * 32 + cls]);
          ctx->h_sum_fail.push_back(t[b * 32 + 16 + cls]); };
      row(MKP_HEMI_CANONICAL, 1);
      std::vector<std::pair<uint32_t, uint32_t>> codes;
      for (size_t si = 0; si < slots.size() && si < 14; si++) if (slots[si].pb == b
          && ((obs >> si) & 1ull)) codes.push_back({slots[si].code_repr, (uint32_t)si});
      std::sort(codes.begin(), codes.end());
      for (auto& cs : codes) row(cs.first, 2 + cs.second);
    }
    out->total_reads_used = t[128 + 4];
    out->n_rows = (uint32_t)ctx->h_sum_base.size(); out->base = ctx->h_sum_base.data(); out->code_repr = ctx->h_sum_code.data();
      out->pass_count = ctx->h_sum_pass.data();
        out->fail_count = ctx->h_sum_fail.data();
    return MKP_OK;
  } catch (const Error& e) { ctx->err = e.what(); return e.status; }
  catch (const std::exception& e) { ctx->err = e.what(); return MKP_E_INVALID; }
}

extern "C" int mkp_histogram_add_bam(mkp_ctx* ctx, const char* bam_path, int argc, const char* const* argv) {
  if (!ctx || !bam_path) return MKP_E_INVALID;
  try { sample_bam(ctx, bam_path, argc, argv, nullptr); return MKP_OK; }
  catch (const Error& e) { ctx->err = e.what(); return e.status; }
  catch (const std::exception& e) { ctx->err = e.what(); return MKP_E_INVALID; }
}

// process_region_batch stand-in reading the BAM itself: an indexed BAM goes through the device ingest (compressed blocks up, records cut and
// packed in HBM — what mkp_pileup_run does per shard), anything else through the host reader + packer
extern "C" int mkp_process_region(mkp_ctx* ctx, const char* bam_path, const mkp_shard* shard, mkp_rows* out) {
  if (!ctx || !bam_path || !shard || !out) return MKP_E_INVALID;
  try {
    std::unique_ptr<BamSource> src = BamSource::open(bam_path, 0);
    const bool host_ingest_env = getenv("MKP_HOST_INGEST") && !strcmp(getenv("MKP_HOST_INGEST"), "1");
    if (src->indexed() && ctx->partition_tags.empty() && !host_ingest_env && shard->tid >= 0 && shard->end > shard->start) {
      if (!ctx->ingest) { ctx->ingest = mkp_internal_ingest_create(ctx->device);
        if (!ctx->ingest) throw Error(MKP_E_DEVICE, "device ingest: cannot create streams on the device");
        }
      std::unique_ptr<DevShard> dev = mkp_internal_ingest_run(ctx->ingest, *src, (uint32_t)shard->tid,
          shard->start > MKP_HALO ? (uint32_t)shard->start - MKP_HALO : 0u, (uint32_t)shard->end + MKP_HALO);
      int rc = mkp_shard_begin(ctx, shard); if (rc != MKP_OK) return rc;
      rc = mkp_internal_shard_attach(ctx, dev.get()); mkp_internal_ingest_recycle(ctx->ingest, dev.get()); if (rc != MKP_OK) return rc;
      return mkp_shard_run(ctx, out);
    }
    int rc = mkp_shard_begin(ctx, shard); if (rc != MKP_OK) return rc;
    BamBatch batch; src->fetch((uint32_t)shard->tid, shard->start > MKP_HALO ? shard->start - MKP_HALO : 0, shard->end + MKP_HALO, &batch);
    std::vector<mkp_record> recs; for (auto& e : batch.recs) recs.push_back(batch.view(e));
    rc = mkp_shard_add_records(ctx, recs.data(), (uint32_t)recs.size()); if (rc != MKP_OK) return rc;
    return mkp_shard_run(ctx, out);
  } catch (const Error& e) { ctx->err = e.what(); return e.status; }
  catch (const std::exception& e) { ctx->err = e.what(); return MKP_E_INVALID; }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// `modkit extract calls` (EntryExtractCalls::run, src/extract/subcommand.rs:452-761): the per-read call table.  The reads are decoded
// on the device by the sampling kernels in per-call mode (MM / ML -> BaseModProbs, edge filter, ReDistribute collapse, the argmax call
// and the thresholded call — mkp_kernels.hip, sample_mode 3); what is left for the host is what the table is made of besides the
// call: read ids, soft clips, reference positions (aligned_pairs_full), base qualities, k-mers, and the text
// (PositionModCalls::to_row, src/extract/writer.rs:46-132).  Records are taken in FILE order — the reference's serial path
// (process_records_to_chan, src/extract/util.rs:519-575), which its golden tests pin; its indexed path hands interval batches to the
// writer in whatever order the Rayon pool finishes them.
namespace {
// f32 through Rust's Display: the shortest decimal that parses back to the same f32, positional notation
std::string f32_display(float v) {
  if (v == 0.0f) return std::signbit(v) ? "-0" : "0";
  char buf[64]; int prec = 1;
  for (; prec <= 9; prec++) { snprintf(buf, sizeof(buf), "%.*e", prec - 1, (double)v); if (strtof(buf, nullptr) == v) break; }
  std::string s(buf); const size_t e = s.find('e'); const int ex = atoi(s.c_str() + e + 1);
  std::string mant = s.substr(0, e); bool neg = false;
  if (!mant.empty() && mant[0] == '-') { neg = true; mant.erase(0, 1); }
  std::string digits; for (char c : mant) if (c != '.') digits.push_back(c);
  while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
  std::string out;
  if (ex >= 0) { if ((int)digits.size() <= ex + 1) out = digits + std::string((size_t)(ex + 1 - (int)digits.size()), '0');
    else out = digits.substr(0, (size_t)ex + 1) + "." + digits.substr((size_t)ex + 1);
    }
  else out = "0." + std::string((size_t)(-ex - 1), '0') + digits;
  return neg ? "-" + out : out;
}
char comp_char(char c) {   // bio::alphabets::dna::revcomp
  switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'N': return 'N';
    case 'R': return 'Y'; case 'Y': return 'R'; case 'K': return 'M'; case 'M': return 'K'; case 'B': return 'V'; case 'V': return 'B';
      case 'D': return 'H'; case 'H': return 'D';
    default: return c;
  }
}
// Kmer::new (util.rs:750-777) as text; '-' where the sequence has no base
std::string kmer_at(const char* seq, size_t n, size_t position, size_t size) {
  const size_t before = size % 2 == 0 ? size / 2 - 1 : size / 2, after = size / 2;
  std::string k;
  for (size_t off = before; off >= 1; off--) k.push_back(position >= off && position - off < n ? seq[position - off] : '-');
  k.push_back(position < n ? seq[position] : '-');
  for (size_t off = 1; off <= after; off++) k.push_back(position + off < n ? seq[position + off] : '-');
  return k;
}
}  // namespace

extern "C" int mkp_extract_calls_main(int argc, const char* const* argv, char* errbuf, size_t errbuf_len) {
  auto fail = [&](int st, const std::string& m) { if (errbuf && errbuf_len) { snprintf(errbuf, errbuf_len, "%s", m.c_str()); } return st; };
  mkp_ctx* ctx = nullptr;
  struct Guard { mkp_ctx** c; ~Guard() { if (*c) mkp_ctx_destroy(*c); } } guard{&ctx};
  try {
    std::string ref_path, exclude_bed; std::vector<std::string> motif_parts; bool cpg = false, bgzf = false;
      bool allow_np = false, mapped_only = false, pass_only = false, no_headers = false, stats = false, ignore_index = false, ignore_implicit = false;
      size_t kmer = 5; int device = 0; long num_reads = -1;
    std::vector<std::string> rest;
    for (int i = 0; i < argc; i++) {
      const std::string s = argv[i];
      auto val = [&]() { if (i + 1 >= argc) throw Error(MKP_E_INVALID, "missing value for " + s); return std::string(argv[++i]); };
      if (s == "--ref" || s == "--reference") ref_path = val(); else if (s == "--allow-non-primary") allow_np = true;
        else if (s == "--mapped-only") mapped_only = true;
      else if (s == "--pass-only" || s == "--pass") pass_only = true; else if (s == "--no-headers") no_headers = true;
        else if (s == "--kmer-size") kmer = std::stoul(val());
      else if (s == "--force" || s == "--suppress-progress") {} else if (s == "--device") device = std::stoi(val());
        else if (s == "--stats") stats = true;
      else if (s == "--num-reads") num_reads = std::stol(val()); else if (s == "--ignore-index") ignore_index = true;
      else if (s == "--ignore-implicit") ignore_implicit = true;
      else if (s == "--exclude-bed" || s == "-v" || s == "--exclude-positions") exclude_bed = val();
      else if (s == "--motif") { motif_parts.push_back(val()); motif_parts.push_back(val()); } else if (s == "--cpg") cpg = true;
      else if (s == "--bgzf") bgzf = true; else if (s == "--out-threads") val();
      else rest.push_back(s);
    }
    if (kmer == 0 || kmer > 50) throw Error(MKP_E_INVALID, "kmer size must be less than or equal to 50");
    if (pass_only && std::find(rest.begin(), rest.end(), "--no-filtering") != rest.end()) throw Error(MKP_E_INVALID,
        "the argument '--no-filtering' cannot be used with '--pass-only'");
    std::vector<const char*> av; for (auto& x : rest) av.push_back(x.c_str());
    Args a; parse_args((int)av.size(), av.data(), &a, true);
    // the flags of the threshold estimate (get_threshold_from_options, command_utils.rs:74-134): calls without a reference position count
    // unless --mapped-only (ReferencePositionFilter::only_mapped_positions, src/extract/util.rs:39-41)
    const BamData bd = load_bam(a.in_bam, 0, false);
    Fasta fasta; if (!ref_path.empty()) fasta = Fasta::load(ref_path);
    BedFilter bed; bool have_bed = !a.include_bed.empty();
    if (have_bed) { std::map<std::string, uint32_t> c2t; for (size_t t = 0; t < bd.ref_names.size(); t++) c2t[bd.ref_names[t]] = (uint32_t)t;
      bed = BedFilter::load(a.include_bed, c2t); }
    // --motif / --cpg (load_regions, src/extract/util.rs:157-277): the include filter becomes the motif hits over every contig of the FASTA that
    // the header names — the whole sequence, lower case matching unless --mask — one position per hit and strand, intersected with the
    // --include-bed positions when both are given.  From there on it IS the include filter: rows, the estimate (handed over as a BED file of
    // its own), the schedule.
    std::string motif_bed_path;
    struct Unlink { std::string* p; ~Unlink() { if (!p->empty()) unlink(p->c_str()); } } unlink_motif_bed{&motif_bed_path};
    if (cpg || !motif_parts.empty()) {
      if (ref_path.empty()) throw Error(MKP_E_INVALID, "--motif / --cpg need --ref");
      std::vector<std::string> parts = motif_parts;   // RegexMotif::from_raw_parts (motif_bed.rs:152-195)
      if (cpg) { bool has = false; for (size_t i = 0; i + 1 < parts.size(); i += 2) if (parts[i] == "CG" && parts[i + 1] == "0") has = true;
        if (!has) { parts.push_back("CG"); parts.push_back("0"); } }
      std::vector<Motif> motifs;
      for (size_t i = 0; i + 1 < parts.size(); i += 2) motifs.push_back(Motif::parse(parts[i], std::stoul(parts[i + 1])));
      BedFilter mf;
      for (size_t t = 0; t < bd.ref_names.size(); t++) {
        const FastaSeq* sq = fasta.get(bd.ref_names[t]); if (!sq) continue;
        std::map<uint32_t, Rule> hits;
        for (auto& m : motifs) motif_hits(sq->data(), sq->size(), m, 0, (uint32_t)t, have_bed ? &bed : nullptr, &hits, !a.mask);
        auto& P = mf.pos[(uint32_t)t]; auto& N = mf.neg[(uint32_t)t];   // (a searched contig is in the filter even without a hit)
        for (auto& kv : hits) { if (kv.second & R_POS) P.push_back({kv.first, (uint64_t)kv.first + 1});
          if (kv.second & R_NEG) N.push_back({kv.first, (uint64_t)kv.first + 1}); }
        merge_spans(P); merge_spans(N);
      }
      bed = std::move(mf); have_bed = true;
      char tmpl[] = "/tmp/mkp_motif_bed_XXXXXX"; const int fd = mkstemp(tmpl);
      if (fd < 0) throw Error(MKP_E_IO, "cannot create a temporary BED for the motif positions");
      motif_bed_path = tmpl; FILE* bf_out = fdopen(fd, "w");
      for (auto* mp : {&bed.pos, &bed.neg}) for (auto& kv : *mp) for (auto& x : kv.second)
        fprintf(bf_out, "%s\t%llu\t%llu\t.\t0\t%c\n", bd.ref_names[kv.first].c_str(), (unsigned long long)x.s, (unsigned long long)x.e,
            mp == &bed.pos ? '+' : '-');
      fclose(bf_out);
    }
    std::vector<std::string> sf;
    for (size_t i = 0; i < rest.size(); i++) {
      if (rest[i] == a.in_bam || rest[i] == a.out_bed) continue;
      // (the motif positions already hold the intersection with the BED: the estimate gets them as its only --include-bed)
      if (!motif_bed_path.empty() && (rest[i] == "--include-bed" || rest[i] == "--include-positions")) { i++; continue; }
      sf.push_back(rest[i]);
    }
    if (!motif_bed_path.empty()) { sf.push_back("--include-bed"); sf.push_back(motif_bed_path); }
    // (--include-bed: "specifying include-only BED outputs only mapped sites", util.rs:136-142 — positions without a reference position do not count)
    if (!mapped_only && !have_bed) sf.push_back("--include-unmapped");
    std::vector<const char*> sav; for (auto& x : sf) sav.push_back(x.c_str());
    mkp_config cfg; memset(&cfg, 0, sizeof(cfg)); cfg.device = device;
    int rc = mkp_ctx_create(&cfg, &ctx);
    if (rc != MKP_OK) throw Error(rc, "no usable gfx950 device (libmkpileup has no CPU path)");
    auto must = [&](int r) { if (r != MKP_OK) throw Error(r, mkp_last_error(ctx)); };
    mkp_caller kc; memset(&kc, 0, sizeof(kc)); kc.max_depth = a.max_depth;
    std::vector<mkp_mod_threshold> per_mod;
    for (auto& raw : a.mod_thresholds) { size_t c = raw.find(':'); uint32_t code;
      if (c == std::string::npos || raw.find(':', c + 1) != std::string::npos || !parse_code(raw.substr(0, c), &code)) throw Error(MKP_E_INVALID,
          "encountered illegal per-mod threshold: " + raw);
      per_mod.push_back({code, strtof(raw.c_str() + c + 1, nullptr)}); }
    if (!a.filter_threshold.empty()) { parse_base_thresholds(a.filter_threshold, &kc); kc.per_mod = per_mod.data();
      kc.n_per_mod = (uint32_t)per_mod.size(); }
    else if (a.no_filtering) { /* MultipleThresholdModCaller::new_passthrough */ }
    else {
      must(mkp_histogram_begin(ctx));
      float q = 0.1f; sample_bam(ctx, a.in_bam.c_str(), (int)sav.size(), sav.data(), &q, nullptr, true);
      float thr[4]; uint8_t has[4]; thresholds_from_sample(ctx, q, thr, has, false);
      for (int b = 0; b < 4; b++) if (has[b]) { kc.has_per_base[b] = 1; kc.per_base_threshold[b] = thr[b]; }
      kc.per_mod = per_mod.data(); kc.n_per_mod = (uint32_t)per_mod.size();
    }
    if (!a.edge_filter.empty()) { kc.edge_filter = 1; kc.edge_inverted = a.invert_edge; size_t c = a.edge_filter.find(',');
      if (c != std::string::npos) { kc.edge_start = (uint32_t)strtoul(a.edge_filter.c_str(), nullptr, 10);
        kc.edge_end = (uint32_t)strtoul(a.edge_filter.c_str() + c + 1, nullptr, 10); }
      else kc.edge_start = kc.edge_end = (uint32_t)strtoul(a.edge_filter.c_str(), nullptr, 10); }
    if (!a.ignore.empty()) { uint32_t code; if (!parse_code(a.ignore, &code)) throw Error(MKP_E_INVALID, "failed to parse mod code " + a.ignore);
      kc.numeric_mode = 2; kc.collapse_code = code; }
    must(mkp_set_caller(ctx, &kc));
    must(mkp_internal_set_extract(ctx, true));
    // --include-bed (ReferencePositionFilter::keep, src/extract/util.rs:44-69), --region, --num-reads (src/extract/util.rs:126-160, 329-575).
    // With an index (and without --ignore-index) the reference walks interval chunks of the targets: --region then selects the records its
    // fetches return — every record overlapping the region, once — where the serial scan looks at every record of the file.  The reference's
    // rows leave in the order its pool finishes the intervals; here in file order.  --num-reads: the serial path's "first N records that
    // reach process_record"; on an indexed BAM it follows the sampling schedule (below).
    // --exclude-bed (load_regions, util.rs:177-187; ReferencePositionFilter::keep = include hit && !exclude hit): a row filter only
    BedFilter exbed; const bool have_ex = !exclude_bed.empty();
    if (have_ex) { std::map<std::string, uint32_t> c2t; for (size_t t = 0; t < bd.ref_names.size(); t++) c2t[bd.ref_names[t]] = (uint32_t)t;
      exbed = BedFilter::load(exclude_bed, c2t); }
    bool use_index = false; { FILE* probe = fopen((a.in_bam + ".bai").c_str(), "rb"); if (probe) { fclose(probe); use_index = !ignore_index; } }
    const bool scheduled = use_index && num_reads >= 0;
    // --ignore-implicit: ReadBaseModProfile::remove_inferred runs in the reference's interval path only (src/extract/util.rs:413-419); its serial
    // scan (no index, --ignore-index) takes the flag and never looks at it
    const bool remove_inferred = ignore_implicit && use_index;
    int64_t reg_tid = -1, reg_s = 0, reg_e = 0; const bool have_region = !a.region.empty();
    if (have_region) {
      std::unique_ptr<BamSource> src = BamSource::open(a.in_bam, 0); const RegionSpec rg = parse_region(a.region, *src);
      for (size_t t = 0; t < bd.ref_names.size(); t++) if (bd.ref_names[t] == rg.name) reg_tid = (int64_t)t;
      reg_s = rg.start; reg_e = rg.end;
    }
    // --num-reads with an index (run_extract_reads, src/extract/util.rs:329-470; subcommand.rs:662-683): SamplingSchedule::from_num_reads over the
    // index counts; every interval of the feeder — threads * 1.5 groups of >= --interval-size bases per super batch — has a RecordSampler
    // of its own: ceil(contig count * interval length / length of the whole super batch) records (get_record_sampler, sampling_schedule.rs:
    // 417-438).  An interval's records are the ones its fetch returns that do not start in front of the previous interval's end — in a
    // sorted file: the records that start inside it (the first interval of a region also takes the records reaching into it) — and it takes
    // the first that many whose process_record succeeds.  Then the records without coordinates, unless a region / --mapped-only excludes
    // them: the first (N - used) that reach process_record.  Rows leave in interval order (the reference: in pool order).
    struct IvQuota { uint32_t start, end; long nr; long used; };   // nr == -2: the schedule holds nothing for the contig (never fetched)
    std::map<uint32_t, std::vector<IvQuota>> iv_quota;              // per contig: the feeder's intervals, ascending and disjoint
    const bool include_unmapped_reads = !have_region && !mapped_only && !have_bed;   // load_regions (util.rs:136-155)
    size_t aligned_used = 0;
    if (scheduled) {
      std::unique_ptr<BamSource> src = BamSource::open(a.in_bam, 0);
      RegionSpec rg; if (have_region) rg = parse_region(a.region, *src);
      const IdxStats st = idxstats(*src, have_region ? &rg : nullptr, have_bed ? &bed : nullptr);
      const std::map<uint32_t, Quota> quota = quota_from_num_reads(st, (size_t)num_reads, include_unmapped_reads);
      const size_t batch_size = std::max<size_t>((size_t)floorf((float)a.threads * 1.5f), 1);
      struct Iv { uint32_t tid, start, end; };
      std::vector<std::vector<Iv>> groups;   // MultiChromCoordinates in feeder order (interval_chunks.rs:563-643)
      { std::vector<Iv> g; uint32_t glen = 0;
        // (--include-bed: every run of BED spans is a reference record of its own — optimize_reference_records, util.rs:287-296)
        std::vector<Contig> recs_t = targets(*src, have_region ? &rg : nullptr);
        if (have_bed) recs_t = bed_contigs(bed, recs_t, a.interval_size);
        for (auto& c : recs_t) {
          for (uint32_t p = c.start; p < c.end();) { const uint32_t e = (uint32_t)std::min<uint64_t>((uint64_t)p + a.interval_size, c.end());
            g.push_back({c.tid, p, e}); glen += e - p;
            if (glen >= a.interval_size) { groups.push_back(g); g.clear(); glen = 0; }
            p = e; } }
        if (!g.empty()) groups.push_back(g); }
      for (size_t g0 = 0; g0 < groups.size(); g0 += batch_size) {
        uint64_t total_len = 0;
        for (size_t g = g0; g < std::min(groups.size(), g0 + batch_size); g++) for (auto& iv : groups[g]) total_len += iv.end - iv.start;
        for (size_t g = g0; g < std::min(groups.size(), g0 + batch_size); g++) for (auto& iv : groups[g]) {
          auto q = quota.find(iv.tid);
          const long nr = q == quota.end() ? 0 /* chrom_has_reads: never fetched */
              : (long)std::ceil((double)q->second.n * ((double)(iv.end - iv.start) / (double)(uint32_t)total_len));
          iv_quota[iv.tid].push_back({iv.start, iv.end, q == quota.end() ? -2 : nr, 0});
        }
      }
    }
    long n_sent = 0; bool done = false;
    FILE* out = (a.out_bed == "-" || a.out_bed == "stdout") ? stdout : fopen(a.out_bed.c_str(), "w+");
    if (!out) throw Error(MKP_E_IO, "failed to make output file " + a.out_bed);
    struct Close { FILE* f; ~Close() { if (f && f != stdout) fclose(f); } } closer{out};
    // --bgzf (src/extract/subcommand.rs:629-660): the same table as BGZF blocks (SAM spec 4.1) cut at 0xff00 bytes, closed by the empty EOF block
    std::vector<uint8_t> zpend, zout;
    auto put = [&](const char* p, size_t n) {
      if (!bgzf) { if (n && fwrite(p, 1, n, out) != n) throw Error(MKP_E_IO, "write error on " + a.out_bed); return; }
      zpend.insert(zpend.end(), (const uint8_t*)p, (const uint8_t*)p + n);
      size_t at = 0; zout.clear();
      while (zpend.size() - at >= MKP_BGZF_BLOCK) { bgzf_block(zpend.data() + at, MKP_BGZF_BLOCK, &zout); at += MKP_BGZF_BLOCK; }
      zpend.erase(zpend.begin(), zpend.begin() + (long)at);
      if (!zout.empty() && fwrite(zout.data(), 1, zout.size(), out) != zout.size()) throw Error(MKP_E_IO, "write error on " + a.out_bed);
    };
    static const char EXTRACT_HEADER[] = "read_id\tforward_read_position\tref_position\tchrom\tmod_strand\tref_strand\tref_mod_strand\tfw_soft_clipped_start\tfw_soft_clipped_end\tread_length\tcall_prob\tcall_code\t"
                                         "base_qual\tref_kmer\tquery_kmer\tcanonical_base\tmodified_primary_base\tfail\tinferred\twithin_alignment\tflag\n";
    if (!no_headers) put(EXTRACT_HEADER, sizeof(EXTRACT_HEADER) - 1);
    BamBatch view; view.base = bd.raw.data();
    uint64_t n_used = 0, n_skipped = 0, n_failed = 0, n_rows = 0;
    static const char NT16[] = "=ACMGRSVTWYHKDBN";
    const size_t BATCH = 1 << 15;
    std::vector<MkpEvent> ev; std::vector<float> vals; std::vector<uint32_t> n_vals;
    for (size_t r0 = 0; r0 < bd.recs.size() && !done; r0 += BATCH) {
      const size_t r1 = std::min(bd.recs.size(), r0 + BATCH);
      std::vector<mkp_record> recs;
      for (size_t i = r0; i < r1; i++) {   // TrackingModRecordIter (mod_bam.rs:53-122) + process_records_to_chan's --mapped-only
        const mkp_record r = view.view(bd.recs[i]);
        // IndexedReader::fetch(tid, start, end): the records overlapping the region (one without reference span counts as one base)
        if (use_index && have_region) {
          int64_t span = 0; const uint8_t* cgp = r.data + r.l_qname;
          for (uint32_t k = 0; k < r.n_cigar; k++) { uint32_t w; memcpy(&w, cgp + 4 * (size_t)k, 4); const uint32_t op = w & 15u;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += w >> 4;
            }
          const int64_t e = (int64_t)r.pos + std::max<int64_t>(span, 1);
          if (r.tid != reg_tid || (int64_t)r.pos >= reg_e || e <= reg_s) continue;
        }
        if (scheduled && r.tid < 0) {   // the schedule's last leg: process_records_to_chan(.., only_mapped = false, allow_non_primary = false, ..)
          if (!include_unmapped_reads) continue;
          if (r.flag & (2048 | 256 | 1024)) { n_skipped++; continue; }
          if (r.l_qseq <= 0) { n_failed++; continue; }
          recs.push_back(r); continue;
        }
        if ((r.flag & (2048 | 256 | 1024)) && !allow_np) { n_skipped++; continue; }
        if (r.l_qseq <= 0) { n_failed++; continue; }
        if ((r.flag & 4) && mapped_only && !scheduled) { n_skipped++; continue; }   // (the schedule's samplers never ask about mapping)
        recs.push_back(r);
      }
      if (recs.empty()) continue;
      must(mkp_internal_sample(ctx, 0, 0, 1, nullptr, recs.data(), (uint32_t)recs.size(), false, &n_vals));
      must(mkp_internal_extract_fetch(ctx, &ev, &vals));
      const std::vector<MkpSlot>& slots = ctx->tables.st.slots;
      std::string text;
      for (size_t i = 0; i < recs.size(); i++) {
        const mkp_record& r = recs[i];
        const MkpReadHdr& h = ctx->sample_shard.hdr[i]; const MkpReadOut& ro = ctx->sample_ro[i];
        if (done) break;
        // --num-reads counts what reaches process_record: records whose tags parse and hold something (an unmapped record under --mapped-only
        // never gets that far: dropped above)
        if (scheduled && r.tid >= 0) {
          auto qi = iv_quota.find((uint32_t)r.tid); if (qi == iv_quota.end() || qi->second.empty()) continue;
          // the interval that takes the record: the first one (feeder order) whose fetch returns it and whose `cut` — the previous interval's
          // end — does not lie behind its start: the first interval that ends behind the record's start, if the record reaches it at all
          int64_t rspan = 0; { const uint8_t* cgp = r.data + r.l_qname;
            for (uint32_t k = 0; k < r.n_cigar; k++) { uint32_t w; memcpy(&w, cgp + 4 * (size_t)k, 4); const uint32_t op = w & 15u;
              if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rspan += w >> 4; } }
          const int64_t rend = (int64_t)r.pos + std::max<int64_t>(rspan, 1);
          auto it = std::upper_bound(qi->second.begin(), qi->second.end(), (int64_t)r.pos,
              [](int64_t p, const IvQuota& q) { return p < (int64_t)q.end; });
          if (it == qi->second.end() || (int64_t)it->start >= rend) continue;
          IvQuota& Q = *it;
          if (Q.nr == -2) continue;                                                        // the schedule holds nothing for this contig
          if (!ro.ok) { if (h.flags & MKP_RF_BAD) n_failed++; else n_skipped++; continue; }  // TrackingModRecordIter never offers it
          if (Q.nr >= 0 && Q.used >= Q.nr) continue;                                       // RecordSampler::ask -> Done
          bool clips_ok = true;                                                            // process_record's own failure: get_soft_clipped
          if (!(r.flag & 4)) { const uint8_t* cgp = r.data + r.l_qname; bool other = false;
            for (uint32_t k = 0; k < r.n_cigar; k++) { uint32_t w; memcpy(&w, cgp + 4 * (size_t)k, 4); if ((w & 15u) != 4u) { other = true; break; } }
            clips_ok = other; }
          if (!clips_ok) { n_failed++; continue; }
          Q.used++; aligned_used++;
          if (mapped_only && (r.flag & 4)) continue;                                       // every row of an unmapped record lacks a reference position
          if (!ro.n_events) continue;
        } else if (scheduled) {
          const long n_un = num_reads > (long)aligned_used ? num_reads - (long)aligned_used : 0;
          if (ro.ok) { n_sent++; if (n_sent >= n_un) done = true; }   // (looked at after the record went to the writer: N = 0 lets one through)
          if (!ro.ok || !ro.n_events) { if (h.flags & MKP_RF_BAD) n_failed++; else n_skipped++; continue; }
        } else {
        if (ro.ok) { n_sent++; if (num_reads >= 0 && n_sent >= num_reads) done = true; }
        if (!ro.ok || !ro.n_events) { if (h.flags & MKP_RF_BAD) n_failed++; else n_skipped++; continue; }
        }
        const bool unmapped = (r.flag & 4) != 0, rev = (r.flag & 16) != 0;
        const size_t L = (size_t)r.l_qseq;
        const uint8_t* cg = r.data + r.l_qname; const uint8_t* sq = cg + 4 * (size_t)r.n_cigar; const uint8_t* ql = sq + (L + 1) / 2;
        auto cig = [&](uint32_t k) { uint32_t w; memcpy(&w, cg + 4 * (size_t)k, 4); return w; };
        size_t sc_start = 0, sc_end = 0; bool cigar_ok = true;   // get_soft_clipped (read_ids_to_base_mod_probs.rs:803-824)
        if (!unmapped) {
          bool broke = false; for (uint32_t k = 0; k < r.n_cigar; k++) { const uint32_t w = cig(k); if ((w & 15u) == 4u) sc_start += w >> 4; else {
              broke = true; break; } }
          if (!broke) cigar_ok = false;
          broke = false; for (uint32_t k = r.n_cigar; k-- > 0;) { const uint32_t w = cig(k); if ((w & 15u) == 4u) sc_end += w >> 4; else {
              broke = true; break; } }
          if (!broke) cigar_ok = false;
        }
        if (!cigar_ok) { n_failed++; continue; }
        const size_t clip_start = rev ? sc_end : sc_start, clip_end = rev ? sc_start : sc_end;
        auto within = [&](size_t qp) { return L >= clip_end && qp >= clip_start && qp < L - clip_end; };
        // forward sequence and qualities
        std::string fwd(L, 'N');
        for (size_t k = 0; k < L; k++) { const uint8_t b = sq[k >> 1]; const char c = NT16[(k & 1) ? (b & 15) : (b >> 4)];
          if (rev) fwd[L - 1 - k] = comp_char(c);
          else fwd[k] = c;
          }
        const std::string chrom = (!unmapped && r.tid >= 0 && (size_t)r.tid < bd.ref_names.size()) ? bd.ref_names[(size_t)r.tid] : std::string();
        const bool have_chrom = !unmapped && r.tid >= 0 && (size_t)r.tid < bd.ref_names.size();
        const FastaSeq* ref_seq = have_chrom ? fasta.get(chrom) : nullptr;
        const bool primary_or_unmapped = r.flag == 0 || r.flag == 16 || r.flag == 4;
        const std::string qname((const char*)r.data, r.l_qname ? (size_t)r.l_qname - 1 : 0);
        // the record's calls come in stored order (ascending stored index): one CIGAR walk gives their reference positions
        struct Row { size_t f; long ref; uint32_t info; float p; };
        std::vector<Row> rows; rows.reserve(ro.n_events);
        uint32_t ck = 0; size_t q_at = 0; long r_at = r.pos;   // op ck starts at stored index q_at / reference position r_at
        for (uint32_t k = 0; k < ro.n_events; k++) {
          const MkpEvent& e = ev[(size_t)h.event_off + k];
          const size_t f = e.pos, q = rev ? L - 1 - f : f;
          long ref = -1;
          if (!unmapped) {
            // an event behind the walk (the kernels emit a record's events in stored order; should that ever change, start over instead of
            // underflowing)
            if (q < q_at) { ck = 0; q_at = 0; r_at = r.pos; }
            while (ck < r.n_cigar) {
              const uint32_t w = cig(ck), op = w & 15u, len = w >> 4;
              const bool cq = op == 0 || op == 1 || op == 4 || op == 7 || op == 8, cr = op == 0 || op == 2 || op == 3 || op == 7 || op == 8;
              if (cq && q < q_at + len) { if (op == 0 || op == 7 || op == 8) ref = r_at + (long)(q - q_at); break; }
              if (cq) q_at += len;
              if (cr) r_at += len;
              ck++;
            }
          }
          rows.push_back({f, ref, e.info, vals[(size_t)h.event_off + k]});
        }
        if (unmapped && rev) std::reverse(rows.begin(), rows.end());   // no alignment strand: ascending forward position
        bool any = false;
        for (const Row& w : rows) {
          if ((mapped_only || have_bed) && (unmapped || w.ref < 0)) continue;   // filter_read_base_mod_probs (src/extract/util.rs:71-124)
          // ... asked with the reference strand of the mod
          if (have_bed && !bed.contains((uint32_t)r.tid, (uint64_t)w.ref, (((w.info >> 2) & 1u) != 0) != rev)) continue;
          if (have_ex && !unmapped && w.ref >= 0 && exbed.contains((uint32_t)r.tid, (uint64_t)w.ref, (((w.info >> 2) & 1u) != 0) != rev)) continue;
          if (!primary_or_unmapped && !within(w.f)) continue;                   // iter_profiles (read_ids_to_base_mod_probs.rs:785-800)
          any = true;
          const uint32_t tb = w.info & 3u, sg = (w.info >> 2) & 1u, inferred = (w.info >> 3) & 1u, thr_cls = (w.info >> 4) & 15u,
              arg_cls = (w.info >> 8) & 15u;
          if (remove_inferred && inferred) continue;
          const bool filtered = thr_cls == 0;
          if (filtered && pass_only) continue;
          std::string code = "-";
          if (arg_cls >= 2) { const uint32_t cr = arg_cls - 2 < slots.size() ? slots[arg_cls - 2].code_repr : 0u;
            code = (cr & 0x80000000u) ? std::to_string(cr & 0x7fffffffu) : std::string(1, (char)cr); }
          std::string qk = kmer_at(fwd.data(), L, w.f, kmer);
          if (sg) { std::string t; for (size_t k = qk.size(); k-- > 0;) t.push_back(qk[k] == '-' ? '-' : comp_char(qk[k])); qk.swap(t); }
          std::string rk = ".";
          if (w.ref >= 0 && ref_seq) rk = kmer_at(ref_seq->data(), ref_seq->size(), (size_t)w.ref, kmer);
          const unsigned bq = ql[rev ? L - 1 - w.f : w.f];
          char line[1200];
          const int ln = snprintf(line, sizeof(line), "%s\t%zu\t%ld\t%s\t%c\t%c\t%c\t%zu\t%zu\t%zu\t%s\t%s\t%u\t%s\t%s\t%c\t%c\t%s\t%s\t%s\t%u\n",
              qname.c_str(), w.f, w.ref >= 0 ? w.ref : -1L,
                   have_chrom ? chrom.c_str() : ".", sg ? '-' : '+', unmapped ? '.' : (rev ? '-' : '+'),
                       unmapped ? '.' : ((sg != 0) != rev ? '-' : '+'), clip_start, clip_end, L,
                   f32_display(w.p).c_str(), code.c_str(), bq, rk.c_str(), qk.c_str(), "ACGT"[tb], "ACGT"[sg ? 3 - tb : tb],
                       filtered ? "true" : "false", inferred ? "true" : "false",
                   (have_chrom && within(w.f)) ? "true" : "false", (unsigned)r.flag);
          if (ln < 0 || (size_t)ln >= sizeof(line)) throw Error(MKP_E_UNSUPPORTED,
              "extract calls: a row is longer than " + std::to_string(sizeof(line)) + " bytes (read or contig name of unusual length)");
          text.append(line, (size_t)ln); n_rows++;
        }
        if (any) n_used++; else n_skipped++;
      }
      put(text.data(), text.size());
    }
    if (bgzf) {
      zout.clear(); if (!zpend.empty()) bgzf_block(zpend.data(), zpend.size(), &zout);
      static const uint8_t eof_block[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      zout.insert(zout.end(), eof_block, eof_block + 28);
      if (fwrite(zout.data(), 1, zout.size(), out) != zout.size()) throw Error(MKP_E_IO, "write error on " + a.out_bed);
    }
    if (stats) fprintf(stderr, "[mkpileup] extract calls: reads=%llu rows=%llu skipped=%llu failed=%llu\n", (unsigned long long)n_used,
        (unsigned long long)n_rows, (unsigned long long)n_skipped, (unsigned long long)n_failed);
    return MKP_OK;
  } catch (const Error& e) { return fail(e.status, e.what()); }
  catch (const std::exception& e) { return fail(MKP_E_INVALID, e.what()); }
}
