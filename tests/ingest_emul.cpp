// Test harness (CPU): the device ingest's per-thread code (modkit_amd/csrc/mkp_ingest_dev.hpp, compiled for the host with
// MKP_INGEST_HOST_SHIM) driven thread by thread over a BAM's inflated stream, against the host path it replaces — BamSource's record
// index + region test and Packer::add (mkp_bam.hpp, mkp_pack.hpp).  Same kernel order as mkp_ingest.hip: count -> scan -> write ->
// parse -> scan -> pack.  Also checks the sliced CRC-32 join of mkp_crc32_blocks against zlib on every BGZF-sized piece of the stream.
//
//   ingest_emul <in.bam> [entry_every=7]      exit 0 = every contig and region compared equal; prints a summary line
#define MKP_INGEST_HOST_SHIM
#include "../modkit_amd/csrc/mkp_pack.hpp"
#include "../modkit_amd/csrc/mkp_ingest_dev.hpp"
#include "../modkit_amd/csrc/mkp_ingest_host.hpp"

#include <cstdio>
#include <cstdlib>
#include <set>

using namespace mkp;

static uint64_t fnv(const std::string& s) { uint64_t h = 1469598103934665603ull; for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull;
  } return h; }

static int fail(const char* what, size_t i, long long a, long long b) {
  fprintf(stderr, "MISMATCH %s at record %zu: device %lld host %lld\n", what, i, a, b); return 1; }

// the CRC kernel's slicing and join, lane by lane
static uint32_t crc_sliced(const uint8_t* p, uint32_t len) {
  uint32_t tab[256]; for (uint32_t v = 0; v < 256; v++) { uint32_t c = v; for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ MKP_CRC_POLY : c >> 1;
    tab[v] = c; }
  const uint32_t S = (len / 64u) & ~3u, first = len - 63u * S;
  uint32_t c[64];
  for (uint32_t lane = 0; lane < 64; lane++) {
    const uint8_t* q = p + (lane ? first + (lane - 1u) * S : 0u); const uint32_t n = lane ? S : first;
    uint32_t x = lane ? 0u : 0xffffffffu; for (uint32_t k = 0; k < n; k++) x = tab[(x ^ q[k]) & 0xffu] ^ (x >> 8);
    c[lane] = x;
  }
  uint32_t sh = gf2_xpow8n(S);
  for (uint32_t L = 0; L < 6; L++) {
    uint32_t nx[64];
    for (uint32_t lane = 0; lane < 64; lane++) { const uint32_t other = c[lane ^ (1u << L)]; const bool left = ((lane >> L) & 1u) == 0u;
      nx[lane] = gf2_mulmod(left ? c[lane] : other, sh) ^ (left ? other : c[lane]); }
    memcpy(c, nx, sizeof(c)); sh = gf2_mulmod(sh, sh);
  }
  return c[0] ^ 0xffffffffu;
}

// The block table as the device builds it (mkp_ingest_host.cpp): the ranges' bytes laid out as the upload lays them out, one
// ingest_walk_blocks per chain, the host's layout over the result — must be the plan the host's pread walk arrives at.
static int device_block_table(const BamSource& src, uint32_t t, const FetchParts& parts, const BamSource::IngestPlan& want) {
  BamSource::IngestPlan plan; src.ingest_ranges(t, parts, &plan);
  if (plan.ranges.empty()) return want.blks.empty() ? 0 : fail("device block table: ranges", t, 0, (long long)want.blks.size());
  std::vector<uint64_t> zbase(plan.ranges.size()); uint64_t zbytes = 0;
  for (size_t r = 0; r < plan.ranges.size(); r++) { zbase[r] = zbytes; zbytes += (plan.ranges[r].file_len + 63) & ~63ull; }
  std::vector<uint8_t> z(zbytes + 64, 0xA5);
  for (size_t r = 0; r < plan.ranges.size(); r++) if (::pread(src.fd(), z.data() + zbase[r], plan.ranges[r].file_len,
      (off_t)plan.ranges[r].file_off) != (ssize_t)plan.ranges[r].file_len) {
    fprintf(stderr, "pread failed\n"); return 2; }
  std::vector<BamSource::IngestChain> chains; src.ingest_chains(plan, &chains);
  std::vector<std::vector<BamSource::IngestBlk>> bparts(chains.size()); uint32_t err = 0;
  for (size_t i = 0; i < chains.size(); i++) {
    const BamSource::IngestRange& rg = plan.ranges[chains[i].range]; const uint64_t zb = zbase[chains[i].range], fo = rg.file_off;
    MkpZChain c; c.start = zb + (chains[i].start - fo); c.stop = chains[i].stop == UINT64_MAX ? ~0ull : zb + (chains[i].stop - fo);
      c.range_end = zb + rg.file_len;
    c.ce = (rg.vend >> 16) >= fo ? zb + ((rg.vend >> 16) - fo) : 0; c.ue = (uint32_t)(rg.vend & 0xffff); c.pad = 0;
    const uint32_t n = ingest_walk_blocks(z.data(), c, nullptr, &err); std::vector<MkpZBlk> out(n + 1);
    if (ingest_walk_blocks(z.data(), c, out.data(), &err) != n) return fail("device block table: the two passes disagree", i, n, 0);
    for (uint32_t k = 0; k < n; k++) bparts[i].push_back({fo + (out[k].coff - zb), out[k].hdr, out[k].clen, out[k].isize, 0});
  }
  if (err) return fail("device block table: error bits", t, err, 0);
  src.ingest_layout(&plan, chains, bparts);
  if (plan.blks.size() != want.blks.size() || plan.raw_total != want.raw_total
      || plan.entries != want.entries) return fail("device block table: size / entries", t, (long long)plan.blks.size(),
      (long long)want.blks.size());
  for (size_t k = 0; k < plan.blks.size(); k++) { const auto& a = plan.blks[k]; const auto& b = want.blks[k];
    if (a.coff != b.coff || a.hdr != b.hdr || a.clen != b.clen || a.isize != b.isize || a.doff != b.doff) return fail("device block table: block", k,
        (long long)a.coff, (long long)b.coff);
      }
  for (size_t r = 0; r < plan.ranges.size(); r++) { const auto& a = plan.ranges[r]; const auto& b = want.ranges[r];
    if (a.blk0 != b.blk0 || a.blk1 != b.blk1 || a.raw_start != b.raw_start || a.raw_limit != b.raw_limit || a.entry0 != b.entry0
        || a.entry1 != b.entry1) return fail("device block table: range", r, (long long)a.raw_limit, (long long)b.raw_limit);
      }
  return 0;
}

// --plan: the indexed side.  BamSource::ingest_plan (block table, window layout, entry points from the BAI) + the chain walk + the region
// test must select exactly the records BamSource::fetch returns, for whole contigs and for sub-regions, and no chain may miss its entry point.
static int plan_mode(const char* path) {
  std::unique_ptr<BamSource> src = BamSource::open(path, 4, true);
  if (!src->indexed()) { printf("ok (no usable index)\n"); return 0; }
  size_t n_cmp = 0, n_seg = 0;
  for (uint32_t t = 0; t < src->ref_names.size(); t++) {
    const uint32_t L = src->ref_lens[t];
    const uint32_t regions[4][2] = {{0, L + 16}, {L / 3, 2 * L / 3 + 1}, {L / 2, L / 2 + 50}, {L > 20000 ? L - 20000 : 0, L}};
    for (auto& rg : regions) {
      BamSource::IngestPlan plan; src->ingest_plan(t, rg[0], rg[1], &plan);
      if (int rc = device_block_table(*src, t, FetchParts{{(int64_t)rg[0], (int64_t)rg[1]}}, plan)) return rc;
      std::vector<uint8_t> raw(plan.raw_total + 8, 0);
      for (auto& r : plan.ranges) {
        std::vector<uint8_t> comp(r.file_len + 8, 0);
        if (::pread(src->fd(), comp.data(), r.file_len, (off_t)r.file_off) != (ssize_t)r.file_len) { fprintf(stderr, "pread failed\n"); return 2; }
        for (size_t k = r.blk0; k < r.blk1; k++) { const auto& b = plan.blks[k];
          if (b.isize) inflate_block(comp.data() + (b.coff - r.file_off) + b.hdr, b.clen, raw.data() + b.doff, b.isize);
          }
      }
      const std::vector<MkpSeg> segs = mkp_plan_segments<MkpSeg>(plan); n_seg += segs.size();
      MkpIngestParams P; memset(&P, 0, sizeof(P)); P.raw_len = plan.raw_total; P.tid = (int32_t)t; P.beg = (int32_t)rg[0]; P.end = (int32_t)rg[1];
        P.n_ref = (int32_t)src->ref_names.size(); P.n_seg = (uint32_t)segs.size();
      uint32_t err = 0; std::vector<unsigned long long> offs;
      for (auto& sg : segs) { const uint32_t n = ingest_walk_segment(raw.data(), plan.raw_total, sg, nullptr, &err); const size_t at = offs.size();
        offs.resize(at + n); ingest_walk_segment(raw.data(), plan.raw_total, sg, offs.data() + at, &err); }
      if (err) return fail("chain error bits", t, err, 0);
      for (size_t i = 1; i < offs.size(); i++) if (offs[i] <= offs[i - 1]) return fail("record offsets not ascending", i, (long long)offs[i],
          (long long)offs[i - 1]);
      std::vector<std::string> dev_names;
      for (auto o : offs) { MkpRecInfo R; ingest_parse_record(raw.data(), o, P, nullptr, &R, &err);
        if (R.kind == 1) dev_names.push_back(std::string((const char*)raw.data() + R.core + 32) + ":" + std::to_string(R.pos) + ":" + std::to_string(R.flag));
        }
      if (err) return fail("record error bits", t, err, 0);
      BamBatch batch; src->fetch(t, rg[0], rg[1], &batch);
      std::vector<std::string> host_names;
      for (auto& e : batch.recs) { const mkp_record r = batch.view(e);
        if (Packer::keep(r)) host_names.push_back(batch.qname(e) + ":" + std::to_string(e.pos) + ":" + std::to_string(e.flag));
        }
      if (dev_names != host_names) return fail("kept records of a region", t, (long long)dev_names.size(), (long long)host_names.size());
      n_cmp += host_names.size();
    }
  }
  // multi-part fetches (a shard made of BED spans): the union of the windows' fetches, every record once, against fetch_parts and
  // against the windows fetched one by one
  size_t n_parts_cmp = 0;
  for (uint32_t t = 0; t < src->ref_names.size(); t++) {
    const int64_t L = src->ref_lens[t]; if (L < 4000) continue;
    const FetchParts layouts[3] = {{{L / 10, L / 10 + 300}, {L / 2, L / 2 + 1200}, {L - 1500, L - 200}}, {{0, 50}, {60, 61}, {L / 3, 2 * L / 3}},
        {{100, 101}, {L / 4, L / 4 + 16}, {L / 4 + 16, L / 4 + 5000}, {L - 50, L + 16}}};
    for (auto& parts : layouts) {
      BamSource::IngestPlan plan; src->ingest_ranges(t, parts, &plan); src->ingest_blocks(&plan);
      if (int rc = device_block_table(*src, t, parts, plan)) return rc;
      std::vector<uint8_t> raw(plan.raw_total + 8, 0);
      for (auto& r : plan.ranges) { std::vector<uint8_t> comp(r.file_len + 8, 0);
        if (::pread(src->fd(), comp.data(), r.file_len, (off_t)r.file_off) != (ssize_t)r.file_len) return 2;
        for (size_t k = r.blk0; k < r.blk1; k++) { const auto& b = plan.blks[k];
          if (b.isize) inflate_block(comp.data() + (b.coff - r.file_off) + b.hdr, b.clen, raw.data() + b.doff, b.isize);
          } }
      const std::vector<MkpSeg> segs = mkp_plan_segments<MkpSeg>(plan);
      std::vector<int32_t> pv; for (auto& pr : parts) { pv.push_back((int32_t)pr.first); pv.push_back((int32_t)pr.second); }
      MkpIngestParams P; memset(&P, 0, sizeof(P)); P.raw_len = plan.raw_total; P.tid = (int32_t)t; P.beg = (int32_t)parts.front().first;
        P.end = (int32_t)parts.back().second; P.n_ref = (int32_t)src->ref_names.size();
      P.n_seg = (uint32_t)segs.size(); P.n_parts = (uint32_t)parts.size();
      uint32_t err = 0; std::vector<std::string> dev_names;
      for (auto& sg : segs) { const uint32_t n = ingest_walk_segment(raw.data(), plan.raw_total, sg, nullptr, &err);
        std::vector<unsigned long long> offs(n); ingest_walk_segment(raw.data(), plan.raw_total, sg, offs.data(), &err);
        for (auto o : offs) { MkpRecInfo R; ingest_parse_record(raw.data(), o, P, pv.data(), &R, &err);
          if (R.kind == 1) dev_names.push_back(std::string((const char*)raw.data() + R.core + 32) + ":" + std::to_string(R.pos));
          } }
      if (err) return fail("multi-part error bits", t, err, 0);
      BamBatch batch; src->fetch_parts(t, parts, &batch);
      std::vector<std::string> host_names; for (auto& e : batch.recs) { const mkp_record r = batch.view(e);
        if (Packer::keep(r)) host_names.push_back(batch.qname(e) + ":" + std::to_string(e.pos));
        }
      if (dev_names != host_names) return fail("kept records of a multi-part fetch", t, (long long)dev_names.size(), (long long)host_names.size());
      // one by one: the union, first sighting kept
      std::vector<std::string> uni; std::set<std::string> seen;
      for (auto& pr : parts) { BamBatch b1; src->fetch(t, (uint32_t)pr.first, (uint32_t)pr.second, &b1); for (auto& e : b1.recs) {
          const mkp_record r = b1.view(e); if (!Packer::keep(r)) continue; const std::string k = b1.qname(e) + ":" + std::to_string(e.pos);
          if (seen.insert(k).second) uni.push_back(k); } }
      if (uni != host_names) return fail("multi-part fetch vs the windows one by one", t, (long long)host_names.size(), (long long)uni.size());
      n_parts_cmp += host_names.size();
    }
  }
  printf("ok plan compared=%zu multipart=%zu segments=%zu\n", n_cmp, n_parts_cmp, n_seg);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: ingest_emul in.bam [entry_every] | ingest_emul --plan in.bam\n"); return 2; }
  if (argc > 2 && !strcmp(argv[1], "--plan")) { try { return plan_mode(argv[2]); } catch (const Error& e) { fprintf(stderr, "plan: %s\n", e.what());
      return 2; } }
  const size_t every = argc > 2 ? std::max(1, atoi(argv[2])) : 7;
  BamData bd;
  try { bd = load_bam(argv[1], 4, true); } catch (const Error& e) { fprintf(stderr, "load: %s\n", e.what()); return 2; }
  const uint8_t* raw = bd.raw.data(); const uint64_t raw_len = bd.raw.size();
  // CRC join: pieces of every size class
  { size_t bad = 0, n = 0; for (uint64_t o = 0; o < raw_len; n++) {
      const uint32_t len = (uint32_t)std::min<uint64_t>(raw_len - o,
          (n % 5 == 0) ? 65280 : (n % 5 == 1) ? 255 : (n % 5 == 2) ? 4099 : (n % 5 == 3) ? 65536 : 257);
      if (crc_sliced(raw + o, len) != crc32_of(raw + o, len)) { bad++; } o += len; }
    if (bad) { fprintf(stderr, "MISMATCH sliced CRC on %zu pieces\n", bad); return 1; } }
  if (bd.recs.empty()) { printf("ok records=0\n"); return 0; }
  // chain segments: an entry point every `every` records (as the BAI's linear index gives them), the last one open-ended
  std::vector<MkpSeg> segs;
  for (size_t i = 0; i < bd.recs.size(); i += every) { MkpSeg s; s.start = bd.recs[i].off - 4; const size_t j = i + every;
    s.exact = j < bd.recs.size(); s.stop = s.exact ? bd.recs[j].off - 4 : raw_len; s.pad = 0; segs.push_back(s); }
  size_t total_cmp = 0, total_bad = 0, total_tags = 0;
  for (size_t t = 0; t < bd.ref_names.size(); t++) {
    // the whole contig and two sub-regions
    const int64_t L = bd.ref_lens[t];
    const int64_t regions[3][2] = {{0, L}, {L / 3, 2 * L / 3 + 1}, {L / 2, L / 2 + 50}};
    for (int rg = 0; rg < 3; rg++) {
      MkpIngestParams P; memset(&P, 0, sizeof(P)); P.raw_len = raw_len; P.tid = (int32_t)t; P.beg = (int32_t)regions[rg][0];
        P.end = (int32_t)regions[rg][1]; P.n_ref = (int32_t)bd.ref_names.size();
      P.n_seg = (uint32_t)segs.size();
      MkpIngestTotals tot; memset(&tot, 0, sizeof(tot));
      std::vector<uint32_t> seg_cnt(segs.size() + 1, 0);
      for (size_t i = 0; i < segs.size(); i++) seg_cnt[i] = ingest_walk_segment(raw, raw_len, segs[i], nullptr, &tot.err);
      { uint64_t run = 0; for (size_t i = 0; i < segs.size(); i++) { const uint32_t v = seg_cnt[i]; seg_cnt[i] = (uint32_t)run; run += v;
        } seg_cnt[segs.size()] = (uint32_t)run; tot.n_all = (uint32_t)run; }
      if (tot.n_all != bd.recs.size()) return fail("record count", 0, tot.n_all, (long long)bd.recs.size());
      P.rec_cap = tot.n_all;
      std::vector<unsigned long long> rec_off(tot.n_all);
      for (size_t i = 0; i < segs.size(); i++) ingest_walk_segment(raw, raw_len, segs[i], rec_off.data() + seg_cnt[i], &tot.err);
      for (size_t i = 0; i < rec_off.size(); i++) if (rec_off[i] + 4 != bd.recs[i].off) return fail("record offset", i, (long long)rec_off[i] + 4,
          (long long)bd.recs[i].off);
      std::vector<MkpRecInfo> info(tot.n_all); std::vector<uint32_t> sz(6 * (size_t)tot.n_all); std::vector<std::pair<int32_t, int32_t>> extra;
      for (uint32_t i = 0; i < tot.n_all; i++) {
        ingest_parse_record(raw, rec_off[i], P, nullptr, &info[i], &tot.err);
        const MkpRecInfo& R = info[i]; const bool k = R.kind == 1, pk = R.kind == 1 || R.kind == 3;
        sz[i] = k; sz[(size_t)tot.n_all + i] = pk ? ingest_cigar_words(R.n_cigar) : 0;
          sz[2 * (size_t)tot.n_all + i] = pk ? ingest_chunk_pairs(R.n_cigar) : 0; sz[3 * (size_t)tot.n_all + i] = pk ? ingest_seq_bytes(R.l_seq) : 0;
        sz[4 * (size_t)tot.n_all + i] = pk ? R.ml_n : 0; sz[5 * (size_t)tot.n_all + i] = R.kind == 3;
        if (R.kind == 2) { const long long e = (long long)R.pos + (R.reflen > 0 ? R.reflen : 1);
          extra.push_back({R.pos, (int32_t)std::min<long long>(e, 0x7fffffffll)}); }
      }
      uint64_t totals[6];
      for (int q = 0; q < 6; q++) { uint64_t run = 0; for (uint32_t i = 0; i < tot.n_all; i++) { uint32_t& a = sz[(size_t)q * tot.n_all + i];
          const uint32_t v = a; a = (uint32_t)run; run += v; } totals[q] = run; }
      tot.n_kept = (uint32_t)totals[0]; tot.n_sample_only = (uint32_t)totals[5];
      const uint32_t n_pk = tot.n_kept + tot.n_sample_only;
      std::vector<MkpReadHdr> hdr(n_pk); std::vector<uint32_t> cigar(totals[1] + 1), chunk(2 * totals[2] + 2), ranks(totals[4] + 1);
        std::vector<uint8_t> seq(totals[3] + 4), ml(totals[4] + 1);
      std::vector<MkpTagRef> tagref((size_t)n_pk * MKP_MAX_TAGS + 1); std::vector<MkpRecDigest> dig(n_pk + 1);
      for (uint32_t i = 0; i < tot.n_all; i++) if (info[i].kind == 1 || info[i].kind == 3)
        ingest_pack_record(raw, info[i], i, info[i].kind == 1 ? sz[i] : tot.n_kept + sz[5 * (size_t)tot.n_all + i], sz[(size_t)tot.n_all + i],
            sz[2 * (size_t)tot.n_all + i], sz[3 * (size_t)tot.n_all + i], sz[4 * (size_t)tot.n_all + i],
                           hdr.data(), chunk.data(), tagref.data(), ranks.data(), dig.data(), &tot);
      // the bulk half, as the 64 lanes of its wave (any order: no lane reads what another wrote)
      for (uint32_t i = 0; i < tot.n_all; i++) if (info[i].kind == 1 || info[i].kind == 3)
        for (uint32_t lane = 64; lane-- > 0;) ingest_copy_record(raw, info[i], sz[(size_t)tot.n_all + i], sz[3 * (size_t)tot.n_all + i],
            sz[4 * (size_t)tot.n_all + i], cigar.data(), seq.data(), ml.data(), lane, 64u);
      // ---- the host path over the same region: the fetch's region test, Packer::keep, Packer::add
      std::vector<mkp_record> recs, so_recs; std::vector<std::pair<int32_t, int32_t>> hextra;
      for (auto& e : bd.recs) {
        if (e.tid != (int32_t)t || (int64_t)e.pos >= P.end || (int64_t)e.end <= P.beg) continue;
        const mkp_record r = bd.view(e);
        if (Packer::keep(r)) recs.push_back(r);
        else if (!(r.flag & (4 | 256 | 1024 | 2048)) && r.l_qseq > 0) so_recs.push_back(r);   // the sampler's candidates that the pileup drops
        else if ((r.flag & 2048) && !(r.flag & (4 | 256 | 512 | 1024))
            && r.n_cigar) hextra.push_back({e.pos, (int32_t)std::min<int64_t>((int64_t)e.pos + std::max<int64_t>(e.reflen, 1), INT32_MAX)});
      }
      Packer pk; ShardHost S; S.tid = (int32_t)t; uint32_t host_err = 0;
      const size_t n_keep_host = recs.size();
      // (one host shard: the kept records, then the sampler-only ones — the order of the device headers)
      try { for (auto& r : recs) pk.add(r, S); for (auto& r : so_recs) pk.add(r, S); }
      catch (const Error& e) {
        const std::string m = e.what();
        host_err = m.find("non-ASCII") != std::string::npos ? MKP_IE_NONASCII : m.find("more than 4 mod codes") != std::string::npos ? MKP_IE_CODES
            : m.find("more than 8 MM tags") != std::string::npos ? MKP_IE_TAGS
                 : m.find("CIGAR query length") != std::string::npos ? MKP_IE_QLEN : m.find("2^26") != std::string::npos ? MKP_IE_SPAN : 0x80000000u;
      }
      if (host_err) { if (!(tot.err & host_err)) return fail("error bits (host threw)", 0, tot.err, host_err); continue; }
      if (tot.err) return fail("error bits (host did not throw)", 0, tot.err, 0);
      if (n_keep_host != tot.n_kept) return fail("kept records", 0, tot.n_kept, (long long)n_keep_host);
      if (so_recs.size() != tot.n_sample_only) return fail("sampler-only records", 0, tot.n_sample_only, (long long)so_recs.size());
      std::sort(extra.begin(), extra.end()); std::sort(hextra.begin(), hextra.end());
      if (extra != hextra) return fail("supplementary spans", 0, (long long)extra.size(), (long long)hextra.size());
      for (uint32_t j = 1; j < n_pk; j++) if (j != tot.n_kept
          && dig[j].win_idx <= dig[j - 1].win_idx) return fail("window order of the packed records", j, (long long)dig[j].win_idx,
          (long long)dig[j - 1].win_idx);
      uint64_t calls = 0, ml_used = 0;
      for (size_t j = 0; j < S.hdr.size(); j++) {
        const MkpReadHdr &d = hdr[j], &h = S.hdr[j]; total_cmp++;
#define CMP(f) if ((long long)d.f != (long long)h.f) { total_bad++; return fail(#f, j, (long long)d.f, (long long)h.f); }
        CMP(ref_start) CMP(ref_end) CMP(l_seq) CMP(n_cigar) CMP(n_tags) CMP(flags) CMP(event_cap)
        for (uint32_t k = 0; k < h.n_cigar; k++) if (cigar[d.cigar_off + k] != S.cigar[h.cigar_off + k]) return fail("cigar word", j,
            cigar[d.cigar_off + k], S.cigar[h.cigar_off + k]);
        for (uint32_t k = 0; k < 2 * ingest_chunk_pairs(h.n_cigar); k++) if (chunk[2 * d.chunk_off + k] != S.chunk_pfx[2 * (size_t)h.chunk_off + k]) return fail("chunk prefix",
            j, chunk[2 * d.chunk_off + k], S.chunk_pfx[2 * (size_t)h.chunk_off + k]);
        if (memcmp(&seq[d.seq_off], &S.seq[h.seq_off], ingest_seq_bytes(h.l_seq)) != 0) return fail("seq bytes", j, 0, 0);
        if (dig[j].name_hash != S.name_hash[j]) return fail("name hash", j, (long long)dig[j].name_hash, (long long)S.name_hash[j]);
        if (h.n_tags) {
          const LayoutHost& Lh = pk.layouts[h.layout];
          if (dig[j].key_hash != fnv(pk.layout_keys[h.layout])) return fail("layout key hash", j, (long long)dig[j].key_hash,
              (long long)fnv(pk.layout_keys[h.layout]));
          for (uint32_t tg = 0; tg < h.n_tags; tg++) {
            const MkpTagRef &a = tagref[d.tag_off + tg], &b = S.tagref[h.tag_off + tg]; total_tags++;
            if (a.n != b.n) return fail("tag calls", j, a.n, b.n);
            if (a.n && memcmp(&ranks[a.rank_off], &S.ranks[b.rank_off], 4 * (size_t)a.n) != 0) return fail("ranks", j, tg, 0);
            const size_t mb = (size_t)a.n * Lh.tags[tg].codes.size();
            if (mb && memcmp(&ml[a.ml_off], &S.ml[b.ml_off], mb) != 0) return fail("ml bytes", j, tg, 0);
            const bool same = tg > 0 && S.tagref[h.tag_off + tg - 1].n == b.n
                && (b.n == 0 || memcmp(&S.ranks[S.tagref[h.tag_off + tg - 1].rank_off], &S.ranks[b.rank_off], 4 * (size_t)b.n) == 0);
            if ((a.pad != 0) != same) return fail("same-list flag", j, a.pad, same);
            calls += b.n; ml_used += mb;
          }
          bool sum_bad = false;   // the planner's probability-sum test (make_resident, mkp_api.cpp)
          if (h.n_tags == 2 && tagref[d.tag_off + 1].pad) {
            const MkpTagRef &t0 = S.tagref[h.tag_off], &t1 = S.tagref[h.tag_off + 1];
              const uint32_t nc0 = (uint32_t)Lh.tags[0].codes.size(), nc1 = (uint32_t)Lh.tags[1].codes.size();
            for (uint32_t q = 0; q < t0.n && !sum_bad; q++) { uint32_t num = 0;
              for (uint32_t i = 0; i < nc0; i++) num += 2u * S.ml[t0.ml_off + q * nc0 + i] + 1u;
              for (uint32_t i = 0; i < nc1; i++) num += 2u * S.ml[t1.ml_off + q * nc1 + i] + 1u;
              sum_bad = num >= 518u; }
          }
          if ((d.pad & 1u) != (sum_bad ? 1u : 0u)) return fail("probability-sum flag", j, d.pad, sum_bad);
        } else if (d.pad) return fail("probability-sum flag on a read without tags", j, d.pad, 0);
      }
      if (calls != tot.n_calls || calls != S.n_calls) return fail("total calls", 0, (long long)tot.n_calls, (long long)S.n_calls);
      if (ml_used != tot.n_ml_used || ml_used != S.ml.size()) return fail("total ML bytes", 0, (long long)tot.n_ml_used, (long long)S.ml.size());
    }
  }
  printf("ok records=%zu compared=%zu tags=%zu segments=%zu\n", bd.recs.size(), total_cmp, total_tags, segs.size());
  return total_bad ? 1 : 0;
}
