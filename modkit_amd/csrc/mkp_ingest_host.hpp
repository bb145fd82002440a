// Internal (not installed): device ingest of one shard window (mkp_ingest_host.cpp) and its hand-over to a context (mkp_api.cpp).
#pragma once
#include "mkp_ctx.hpp"

struct MkpRecInfo;
struct mkp_dev_ingest;

namespace mkp {
// one shard's records as the device ingest leaves them: the big arrays in HBM, the planner's digest on the host
struct DevShard {
  ShardHost S;                 // hdr, tagref (MKP_MAX_TAGS per read), name_hash, extra_spans, dev_sum2; dev_packed = true; hdr[i].layout indexes `layouts`
  Packer layouts;              // MM header structures of this shard, in order of first appearance
  DevBuf d_cigar, d_chunk, d_seq, d_tagref, d_ranks, d_ml;
  std::vector<MkpRecInfo> info_host;   // scratch of the layout interning
  bool layouts_adopted = false, bound = false;   // layout ids already mapped into a context's table; currently swapped into a context for sampling
  uint64_t n_blocks = 0, n_segments = 0, n_records = 0, raw_bytes = 0, comp_bytes = 0;
  double ms_plan = 0, ms_upload = 0, ms_inflate = 0, ms_pack = 0, ms_digest = 0, ms_total = 0, ms_alloc = 0, ms_wait = 0, ms_kernel = 0;
  DevShard() = default; DevShard(const DevShard&) = delete; DevShard& operator=(const DevShard&) = delete;
  ~DevShard();
};
}  // namespace mkp

// chain segments of a plan: from every known record start to the next one (the chain must land on it), the last of a range to the range's limit
template <class Seg> inline std::vector<Seg> mkp_plan_segments(const mkp::BamSource::IngestPlan& plan) {
  std::vector<Seg> segs(plan.entries.size());
  for (auto& rg : plan.ranges) for (size_t k = rg.entry0; k < rg.entry1; k++) { Seg s; s.start = plan.entries[k];
    s.exact = k + 1 < rg.entry1 ? 1u : 0u; s.stop = s.exact ? plan.entries[k + 1] : rg.raw_limit; s.pad = 0; segs[k] = s; }
  return segs;
}

// the BGZF inflate kernel for a launch of n blocks (mkp_api.cpp): one wave per block (speculative token decode) below 28 000 blocks, one thread per
// block (second edition) from there; MKP_INFLATE_KERNEL=wave|thread|thread2 forces one
hipError_t mkp_launch_inflate_auto(hipStream_t st, const uint8_t* in, const void* blks, uint32_t n, uint8_t* out, uint32_t* status);

mkp_dev_ingest* mkp_internal_ingest_create(int device);
void mkp_internal_ingest_destroy(mkp_dev_ingest* d);
// the indexed fetch of [beg, end) on `tid` — every record overlapping it — inflated, cut, filtered and packed on the device; throws mkp::Error
// (several windows — ascending, disjoint: a shard made of BED spans — select the union of their fetches, every record once)
std::unique_ptr<mkp::DevShard> mkp_internal_ingest_run(mkp_dev_ingest* d, const mkp::BamSource& bam, uint32_t tid, const mkp::FetchParts& parts);
inline std::unique_ptr<mkp::DevShard> mkp_internal_ingest_run(mkp_dev_ingest* d, const mkp::BamSource& bam, uint32_t tid, uint32_t beg,
    uint32_t end) {
  return mkp_internal_ingest_run(d, bam, tid, mkp::FetchParts{{(int64_t)beg, (int64_t)end}});
}
// the shard begun with mkp_shard_begin takes these records instead of mkp_shard_add_records: device arrays swapped into the context
// (what the context held before stays in `sh` and goes back to the ingest object with mkp_internal_ingest_recycle)
int mkp_internal_shard_attach(mkp_ctx* c, mkp::DevShard* sh);
void mkp_internal_ingest_recycle(mkp_dev_ingest* d, mkp::DevShard* sh);
// Resident sampling over several shards: swap `sh`'s digest and device arrays into the context (its layouts are mapped into the context's
// table the first time) so that mkp_internal_sample_resident reads them; the same call again swaps them back out.  The shard stays owned
// by `sh` and can be attached for its pileup later.
int mkp_internal_sample_bind(mkp_ctx* c, mkp::DevShard* sh);
