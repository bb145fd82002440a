// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_core.hpp header).
// `modkit_oracle pileup <in.bam> <out.bed> [flags]` — CPU restatement of
// `modkit pileup` (src/pileup/subcommand.rs:382-816): option resolution, interval
// feeder, motif / BED focus positions, default threshold sampler, bedMethyl writer.
// Flags carry the reference's names so tests read like tests/test_pileup.rs.
#include <atomic>
#include <chrono>
#include <fstream>
#include <sstream>
#include <thread>

#include "oracle_pileup.hpp"
#include "oracle_extract.hpp"

using namespace mko;

// ------------------------------------------------------------------- FASTA
struct Fasta {
  std::map<std::string, std::string> seqs;
  static Fasta load(const std::string& path) {
    Fasta f; std::ifstream in(path);
    if (!in) throw MkErr("cannot open fasta " + path);
    std::string line, name;
    while (std::getline(in, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (line.empty()) continue;
      if (line[0] == '>') { name = line.substr(1); size_t sp = name.find_first_of(" \t"); if (sp != std::string::npos) name = name.substr(0, sp);
        f.seqs[name]; }
      else f.seqs[name] += line;
    }
    return f;
  }
  // bio IndexedReader fetch + read: error when stop > length
  std::string fetch(const std::string& contig, uint64_t start, uint64_t stop) const {
    auto it = seqs.find(contig);
    if (it == seqs.end()) throw MkErr("Unknown sequence name " + contig);
    if (stop > it->second.size() || start > stop) throw MkErr("FASTA read interval was out of bounds");
    return it->second.substr(start, stop - start);
  }
};

// ------------------------------------------------------------------- motifs
struct Motif {  // RegexMotif (find_motifs/motif_bed.rs:143-252)
  std::string raw;
  std::vector<std::string> fwd_classes, rev_classes;  // allowed letters per position
  MotifInfo info;
  std::string label() const { return raw + "," + std::to_string(info.forward_offset); }
};
static std::string iupac_class(char c) {  // iupac_to_regex 21-46
  switch (c) {
    case 'A': return "A"; case 'C': return "C"; case 'G': return "G"; case 'T': return "T"; case 'U': return "U";
    case 'M': return "AC"; case 'R': return "AG"; case 'W': return "AT"; case 'S': return "CG"; case 'Y': return "CT";
    case 'K': return "GT"; case 'V': return "ACG"; case 'H': return "ACT"; case 'D': return "AGT"; case 'B': return "CGT";
    case 'X': case 'N': return "ACGT";
    default: throw MkErr(std::string("Invalid IUPAC code: ") + c);
  }
}
static Motif parse_motif(const std::string& raw, size_t offset) {  // parse_string 197-223
  Motif m; m.raw = raw;
  if (raw.size() == 1 && !(raw == "A" || raw == "C" || raw == "G" || raw == "T")) throw MkErr("degenerate single base motif");
  for (char c : raw) m.fwd_classes.push_back(iupac_class(c));
  // motif_rev_comp on the regex text: reverse, complement letters (48-64)
  for (auto it = m.fwd_classes.rbegin(); it != m.fwd_classes.rend(); ++it) {
    std::string cls(it->rbegin(), it->rend());
    for (auto& ch : cls) ch = (ch == 'A') ? 'T' : (ch == 'C') ? 'G' : (ch == 'G') ? 'C' : (ch == 'T') ? 'A' : (ch == 'U') ? 'A' : ch;
    m.rev_classes.push_back(cls);
  }
  if (raw.size() < offset + 1) throw MkErr("motif not long enough for offset");
  m.info.forward_offset = offset; m.info.reverse_offset = raw.size() - (offset + 1); m.info.length = raw.size();
  // palindrome == regex strings equal (220): compare as written strings
  auto as_regex = [](const std::vector<std::string>& cl) { std::string s; for (auto& c : cl) { if (c.size() == 1) s += c; else s += "[" + c + "]";
    } return s; };
  m.info.is_palindrome = as_regex(m.fwd_classes) == as_regex(m.rev_classes);
  return m;
}
static std::vector<size_t> find_overlapping(const std::string& seq, const std::vector<std::string>& cls) {
  std::vector<size_t> out; size_t L = cls.size();
  if (seq.size() < L) return out;
  for (size_t i = 0; i + L <= seq.size(); i++) {
    bool ok = true;
    for (size_t j = 0; j < L && ok; j++) ok = cls[j].find(seq[i + j]) != std::string::npos;
    if (ok) out.push_back(i);
  }
  return out;
}
// find_motif_hits (288-337): (position, is_negative)
static std::vector<std::pair<size_t, bool>> find_motif_hits(const std::string& seq, const Motif& m) {
  std::vector<std::pair<size_t, bool>> hits;
  if (m.info.is_palindrome) {
    for (size_t s : find_overlapping(seq, m.fwd_classes)) {
      if (m.info.forward_offset <= m.info.reverse_offset) { hits.push_back({s + m.info.forward_offset, false});
        hits.push_back({s + m.info.reverse_offset, true}); }
      else { hits.push_back({s + m.info.reverse_offset, true}); hits.push_back({s + m.info.forward_offset, false}); }
    }
  } else if (m.info.length == 1) {
    char fw = m.raw[0], rv = comp_char(fw);
    for (size_t i = 0; i < seq.size(); i++) { if (seq[i] == fw) hits.push_back({i, false}); else if (seq[i] == rv) hits.push_back({i, true}); }
  } else {
    for (size_t s : find_overlapping(seq, m.fwd_classes)) hits.push_back({s + m.info.forward_offset, false});
    for (size_t s : find_overlapping(seq, m.rev_classes)) hits.push_back({s + m.info.reverse_offset, true});
    std::stable_sort(hits.begin(), hits.end(), [](auto& a, auto& b) { return a.first < b.first; });
  }
  return hits;
}

// -------------------------------------------------------- StrandedPositionFilter
struct PositionFilter {  // position_filter.rs:20-24
  std::map<uint32_t, std::vector<Iv>> pos, neg;
  bool contains(int32_t tid, uint64_t p, bool neg_strand) const {
    const auto& m = neg_strand ? neg : pos;
    auto it = m.find((uint32_t)tid);
    return it != m.end() && lapper_any(it->second, p, p + 1);
  }
  bool overlaps_not_stranded(uint32_t tid, uint64_t s, uint64_t e) const {
    auto a = pos.find(tid); if (a != pos.end() && lapper_any(a->second, s, e)) return true;
    auto b = neg.find(tid); return b != neg.end() && lapper_any(b->second, s, e);
  }
  bool contains_chrom(int64_t tid) const { return tid >= 0 && (pos.count((uint32_t)tid) || neg.count((uint32_t)tid)); }
  static PositionFilter from_bed(const std::string& path, const std::map<std::string, uint32_t>& chrom_to_tid) {  // 230-347
    PositionFilter pf; std::ifstream in(path);
    if (!in) throw MkErr("cannot open BED " + path);
    std::string line; std::set<std::string> warned;
    while (std::getline(in, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (line.empty()) continue;
      std::istringstream ss(line); std::vector<std::string> parts; std::string w;
      while (ss >> w) parts.push_back(w);
      if (parts.size() < 3) continue;
      if (warned.count(parts[0])) continue;
      char* e1; char* e2;
      uint64_t start = strtoull(parts[1].c_str(), &e1, 10), stop = strtoull(parts[2].c_str(), &e2, 10);
      if (*e1 || *e2 || parts[1].empty() || parts[2].empty()) continue;
      bool ps, ns;
      if (parts.size() == 3) { ps = ns = true; }
      else if (parts.size() >= 6) {
        if (parts[5] == "+") { ps = true; ns = false; } else if (parts[5] == "-") { ps = false; ns = true; } else if (parts[5] == ".") {
          ps = ns = true; } else continue;
      } else continue;
      auto it = chrom_to_tid.find(parts[0]);
      if (it == chrom_to_tid.end()) { warned.insert(parts[0]); continue; }
      if (ps) pf.pos[it->second].push_back({start, stop});
      if (ns) pf.neg[it->second].push_back({start, stop});
    }
    if (pf.pos.empty() && pf.neg.empty()) throw MkErr("zero valid positions parsed from BED file");
    for (auto& kv : pf.pos) lapper_merge(kv.second);
    for (auto& kv : pf.neg) lapper_merge(kv.second);
    return pf;
  }
};

struct ReferenceRecord { uint32_t tid, start, length; std::string name; uint32_t end() const { return start + length; } };

// optimize_reference_records + group_genome_intervals (position_filter.rs:103-210)
static std::vector<ReferenceRecord> optimize_reference_records(const PositionFilter& pf, const std::vector<ReferenceRecord>& recs,
    uint32_t interval_size) {
  std::map<uint32_t, ReferenceRecord> lut; for (auto& r : recs) lut[r.tid] = r;  // later duplicates overwrite (collect into HashMap)
  std::set<uint32_t> tids; for (auto& kv : pf.pos) tids.insert(kv.first); for (auto& kv : pf.neg) tids.insert(kv.first);
  std::vector<ReferenceRecord> out;
  for (uint32_t tid : tids) {
    auto li = lut.find(tid); if (li == lut.end()) continue;
    std::vector<Iv> ivs;
    auto a = pf.pos.find(tid); if (a != pf.pos.end()) ivs.insert(ivs.end(), a->second.begin(), a->second.end());
    auto b = pf.neg.find(tid); if (b != pf.neg.end()) ivs.insert(ivs.end(), b->second.begin(), b->second.end());
    lapper_merge(ivs);
    if (ivs.empty()) continue;
    std::vector<Iv> agg; Iv cur = ivs[0];
    for (size_t i = 1; i < ivs.size(); i++) {
      if (cur.stop - cur.start > interval_size) { agg.push_back(cur); cur = ivs[i]; continue; }
      cur.stop = ivs[i].stop;
    }
    agg.push_back(cur);
    for (auto& iv : agg) out.push_back({tid, (uint32_t)iv.start, (uint32_t)(iv.stop - iv.start), li->second.name});
  }
  return out;
}

// ------------------------------------------------ MotifLocationsLookup (fasta.rs)
struct MotifLocations { std::map<uint32_t, StrandRule> locs; };  // one per motif, this tid
struct MotifLookup {
  Fasta fasta; bool mask = false; std::vector<Motif> motifs; uint64_t longest = 0;
  std::vector<MotifLocations> motifs_on_seq(std::string seq, uint64_t start, uint32_t tid, const PositionFilter* pf) const {  // 42-90
    std::vector<MotifLocations> out;
    for (const Motif& m : motifs) {
      MotifLocations ml;
      for (auto& h : find_motif_hits(seq, m)) {
        uint64_t p = h.first + start;
        if (pf && !pf->contains((int32_t)tid, p, h.second)) continue;
        auto it = ml.locs.find((uint32_t)p);
        if (it != ml.locs.end()) it->second = rule_absorb(it->second, h.second);
        else ml.locs[(uint32_t)p] = h.second ? RULE_NEG : RULE_POS;
      }
      out.push_back(std::move(ml));
    }
    return out;
  }
  std::string get_seq(const std::string& contig, uint64_t s, uint64_t e) const {
    std::string seq = fasta.fetch(contig, s, e);
    if (!mask) for (auto& c : seq) c = (char)toupper((unsigned char)c);
    return seq;
  }
  // get_motif_positions (190-228) / _combine_strands (92-188); returns new interval end
  std::vector<MotifLocations> get_motif_positions(const std::string& contig, uint32_t tid, uint32_t ref_end_u, uint64_t rstart, uint64_t rend,
                                                  const PositionFilter* pf, bool combine, uint32_t* end_out) const {
    if (!combine) { *end_out = (uint32_t)rend; return motifs_on_seq(get_seq(contig, rstart, rend), rstart, tid, pf); }
    uint64_t ref_end = ref_end_u, buffer = longest * 5, end = rend;
    uint64_t end_w = std::min(rend + buffer, ref_end);
    uint64_t too_close = end_w >= longest ? end_w - longest : 0;
    for (;;) {
      auto locs = motifs_on_seq(get_seq(contig, rstart, end_w), rstart, tid, pf);  // throws if end_w > contig (the reference loops forever here)
      std::vector<Iv> ivs;
      for (size_t i = 0; i < motifs.size(); i++) {
        size_t len = motifs[i].info.length, fo = motifs[i].info.forward_offset;
        uint64_t adj = len >= fo ? len - fo : len;
        for (auto& kv : locs[i].locs) ivs.push_back({kv.first, kv.first + adj});
      }
      lapper_merge(ivs);
      uint64_t search_end = end;
      { uint64_t qs = end >= 1 ? end - 1 : 0; for (auto& iv : ivs) if (iv.start < end && iv.stop > qs) { search_end = iv.stop; break; } }
      if (search_end < too_close || end_w >= ref_end) {
        for (auto& ml : locs) { for (auto it = ml.locs.begin(); it != ml.locs.end();) { if ((uint64_t)it->first <= search_end) ++it;
            else it = ml.locs.erase(it);
          } }
        *end_out = (uint32_t)search_end;
        return locs;
      }
      end = end_w; end_w += buffer; too_close = end_w >= longest ? end_w - longest : 0;
    }
  }
};

// ------------------------------------------------------------ FocusPositions ctors
// interval_chunks.rs:62-202
static FocusPositions focus_new_motif(const std::vector<MotifLocations>& mls, const std::vector<Motif>& motifs, uint32_t start, uint32_t end) {
  FocusPositions f; f.kind = FocusPositions::MOTIF;
  bool all_single = true; for (auto& m : motifs) if (m.info.length != 1) all_single = false;
  auto in_range = [&](uint32_t p) { return p >= start && p < end; };
  if (mls.size() == 1) {
    for (auto& kv : mls[0].locs) {
      if (!in_range(kv.first)) continue;
      auto it = f.positions.find(kv.first);
      if (it != f.positions.end() && !all_single) it->second = rule_combine(it->second, kv.second); else f.positions[kv.first] = kv.second;
      if (kv.second == RULE_POS || kv.second == RULE_BOTH) f.positive_motif_ids[kv.first] = {0};
      if (kv.second == RULE_NEG || kv.second == RULE_BOTH) f.negative_motif_ids[kv.first] = {0};
    }
  } else if (all_single) {
    auto add = [&](const char* top, const char* bottom) {  // add_single_base_motifs 204-248
      int a = -1, t = -1;
      for (size_t i = 0; i < motifs.size(); i++) { if (motifs[i].raw == top) a = (int)i; if (motifs[i].raw == bottom) t = (int)i; }
      if (a < 0) return;
      for (auto& kv : mls[a].locs) {
        if (!in_range(kv.first)) continue;
        if (t >= 0) { f.positions[kv.first] = RULE_BOTH; f.positive_motif_ids[kv.first] = {(size_t)a, (size_t)t};
          f.negative_motif_ids[kv.first] = {(size_t)a, (size_t)t}; }
        else { f.positions[kv.first] = kv.second; if (kv.second == RULE_POS) f.positive_motif_ids[kv.first] = {(size_t)a};
          else if (kv.second == RULE_NEG) f.negative_motif_ids[kv.first] = {(size_t)a};
          }
      }
    };
    add("A", "T"); add("C", "G");
  } else {
    for (size_t id = 0; id < mls.size(); id++) for (auto& kv : mls[id].locs) {
      if (!in_range(kv.first)) continue;
      auto it = f.positions.find(kv.first);
      if (it != f.positions.end()) it->second = rule_combine(it->second, kv.second); else f.positions[kv.first] = kv.second;
      if (kv.second == RULE_POS || kv.second == RULE_BOTH) f.positive_motif_ids[kv.first].push_back(id);
      if (kv.second == RULE_NEG || kv.second == RULE_BOTH) f.negative_motif_ids[kv.first].push_back(id);
    }
  }
  return f;
}
// 250-297
static FocusPositions focus_new_motif_combine(const std::vector<MotifLocations>& mls, const std::vector<Motif>& motifs, uint32_t start,
    uint32_t end) {
  FocusPositions f; f.kind = FocusPositions::MOTIF_COMBINE;
  for (size_t id = 0; id < mls.size(); id++) for (auto& kv : mls[id].locs) {
    if (!(kv.first >= start && kv.first < end)) continue;
    auto it = f.positions.find(kv.first);
    if (it != f.positions.end()) it->second = rule_combine(it->second, kv.second); else f.positions[kv.first] = kv.second;
    if (kv.second == RULE_POS || kv.second == RULE_BOTH) f.positive_motifs[kv.first].push_back({motifs[id].info, id});
    else f.negative_motif_ids[kv.first].push_back(id);
  }
  return f;
}
static FocusPositions focus_new_regions(const PositionFilter& pf, uint32_t tid, uint32_t start, uint32_t end) {  // 299-349
  FocusPositions f; f.kind = FocusPositions::REGIONS;
  auto clip = [&](const std::map<uint32_t, std::vector<Iv>>& m, std::vector<Iv>& out) {
    auto it = m.find(tid); if (it == m.end()) return;
    for (auto& iv : it->second) if (iv.start < end && iv.stop > start) out.push_back({std::max<uint64_t>(iv.start, start),
        std::min<uint64_t>(iv.stop, end)});
    lapper_merge(out);
  };
  clip(pf.pos, f.pos_intervals); clip(pf.neg, f.neg_intervals);
  return f;
}

struct ChromCoordinates { uint32_t tid, start, end; FocusPositions focus; uint32_t len() const { return end >= start ? end - start : 0; } };
typedef std::vector<ChromCoordinates> MultiChromCoordinates;

// ReferenceIntervalsFeeder (interval_chunks.rs:497-652)
struct Feeder {
  std::vector<ReferenceRecord> contigs; size_t next_contig = 0;
  size_t batch_size; uint32_t interval_size; const MotifLookup* motifs; const PositionFilter* pf; bool combine;
  ReferenceRecord cur; uint32_t cur_pos = 0; bool done = false;
  Feeder(std::vector<ReferenceRecord> recs, size_t bs, uint32_t is, bool comb, const MotifLookup* ml, const PositionFilter* p)
      : contigs(std::move(recs)), batch_size(bs), interval_size(is), motifs(ml), pf(p), combine(comb) {
    if (combine && !motifs) throw MkErr("cannot combine strands without a motif");
    if (contigs.empty()) throw MkErr("should be at least 1 contig");
    cur = contigs[0]; next_contig = 1; cur_pos = cur.start;
  }
  void update_current() { if (next_contig < contigs.size()) { cur = contigs[next_contig++]; cur_pos = cur.start; } else done = true; }
  bool next_batch(std::vector<MultiChromCoordinates>* ret) {
    ret->clear(); MultiChromCoordinates batch; uint32_t batch_length = 0;
    for (;;) {
      if (done) break;
      if (ret->size() >= batch_size) break;
      uint32_t start = cur_pos, tid = cur.tid;
      uint32_t end = (uint32_t)std::min<uint64_t>((uint64_t)start + interval_size, cur.end());
      ChromCoordinates cc; cc.tid = tid; cc.start = start;
      if (motifs) {
        uint32_t new_end;
        auto locs = motifs->get_motif_positions(cur.name, tid, cur.end(), start, end, pf, combine, &new_end);
        end = std::min(new_end, cur.end());
        cc.focus = combine ? focus_new_motif_combine(locs, motifs->motifs, start, end) : focus_new_motif(locs, motifs->motifs, start, end);
      } else if (pf) cc.focus = focus_new_regions(*pf, tid, start, end);
      cc.end = end;
      batch_length += cc.len(); batch.push_back(std::move(cc));
      if (batch_length >= interval_size) { ret->push_back(std::move(batch)); batch.clear(); batch_length = 0; }
      if (end >= cur.end()) update_current(); else cur_pos = end;
    }
    if (!batch.empty()) ret->push_back(std::move(batch));
    return !ret->empty();
  }
};

// --------------------------------------------------------------------- options
struct Options {
  std::string in_bam, out_bed, region, sample_region, include_bed, ignore, ref_fasta, edge_filter, preset;
  uint32_t max_depth = 8000, interval_size = 100000, sampling_interval_size = 1000000;
  size_t threads = 4, num_reads = 10042; bool have_chunk = false; size_t chunk_size = 0;
  bool have_frac = false; double sampling_frac = 0; bool no_filtering = false; float filter_percentile = 0.1f;
  bool have_seed = false; uint64_t seed = 0;   // --seed (RecordSampler::new_sample_frac, record_sampler.rs:29-38)
  bool serial_sampler = false;   // sample-probs / summary / extract calls: a BAM without an index is sampled serially (reads_sampler/mod.rs:129-158)
  std::vector<std::string> filter_threshold, mod_thresholds, motif_parts;
  bool include_unmapped = false, force_allow = false, cpg = false, mask = false, combine_mods = false, combine_strands = false;
  bool invert_edge = false, mixed_delim = false, with_header = false;
  size_t workers = 1;  // oracle-only: interval-parallel std::thread workers for the CPU baseline
  bool hemi = false;   // `pileup-hemi` (DuplexModBamPileup, src/pileup/subcommand.rs:827-1514)
  // `sample-probs` (SampleModBaseProbs, src/commands.rs:549-887): the percentiles table only
  bool sample_probs_cmd = false; std::string percentiles = "0.1,0.5,0.9";
};

struct Region { std::string name; uint32_t start, end; };
static Region parse_region(const std::string& raw, const BamFile& bam) {  // util.rs:463-524
  if (raw.find(':') != std::string::npos) {
    size_t c = raw.find(':'); if (raw.find(':', c + 1) != std::string::npos) throw MkErr("invalid region " + raw);
    std::string name = raw.substr(0, c), se = raw.substr(c + 1);
    std::vector<std::string> parts; size_t s = 0;
    for (;;) { size_t d = se.find('-', s); parts.push_back(se.substr(s, d == std::string::npos ? std::string::npos : d - s));
      if (d == std::string::npos) break;
      s = d + 1; }
    if (parts.size() != 2) throw MkErr("invalid region " + raw);
    uint32_t v[2];
    for (int i = 0; i < 2; i++) { std::string cl; for (char ch : parts[i]) if (ch != ',') cl += ch;
      if (cl.empty()) throw MkErr("invalid region " + raw);
      char* e; v[i] = (uint32_t)strtoul(cl.c_str(), &e, 10); if (*e) throw MkErr("invalid region " + raw); }
    if (v[1] <= v[0]) throw MkErr("invalid region " + raw);
    return {name, v[0], v[1]};
  }
  int tid = bam.tid_of(raw); if (tid < 0) throw MkErr("contig-missing");
  return {raw, 0, bam.ref_lens[tid]};
}
static std::vector<ReferenceRecord> get_targets(const BamFile& bam, const Region* region) {  // util.rs:409-446
  std::vector<ReferenceRecord> out;
  for (size_t tid = 0; tid < bam.ref_names.size(); tid++) {
    if (region) {
      if (bam.ref_names[tid] == region->name) out.push_back({(uint32_t)tid, region->start, region->end - region->start, bam.ref_names[tid]});
      }
    else out.push_back({(uint32_t)tid, 0, bam.ref_lens[tid], bam.ref_names[tid]});
  }
  return out;
}

// idxstats as rust-htslib index_stats reports them from the BAI pseudo-bins: per tid
// mapped / placed-unmapped counts; (-1) = records without coordinates.
struct IdxStats {
  std::map<int64_t, uint64_t> tid_mapped; uint64_t mapped = 0, unmapped = 0;
  static IdxStats make(const BamFile& bam, const Region* region, const PositionFilter* pf) {  // sampling_schedule.rs:649-716
    IdxStats st; int region_tid = region ? bam.tid_of(region->name) : -1;
    if (region && region_tid < 0) throw MkErr("did not find target_id for region");
    std::vector<uint64_t> m(bam.ref_names.size(), 0), u(bam.ref_names.size(), 0); uint64_t nocoor = 0;
    for (auto& r : bam.recs) { if (r.tid < 0) { nocoor++; continue; } if (r.flag & 4) u[r.tid]++; else m[r.tid]++; }
    auto keep = [&](int64_t tid) { if (region) return tid == region_tid; if (pf) return pf->contains_chrom(tid); return true; };
    for (size_t tid = 0; tid < m.size(); tid++) if (keep((int64_t)tid)) { st.mapped += m[tid]; st.unmapped += u[tid];
      st.tid_mapped[(int64_t)tid] = m[tid]; }
    if (keep(-1)) st.unmapped += nocoor;
    return st;
  }
};

struct Count { enum K { COUNT, ALL } k = COUNT; size_t n = 0; };
struct SamplingSchedule {  // sampling_schedule.rs:73-76
  std::map<uint32_t, Count> counts; bool has_unmapped = false;
  static SamplingSchedule from_num_reads(const IdxStats& st, size_t num_reads, bool include_unmapped) {  // 171-273
    SamplingSchedule s; uint64_t total_u = include_unmapped ? st.mapped + st.unmapped : st.mapped;
    if (total_u == 0) throw MkErr("zero reads found in bam index");
    float total = (float)total_u; size_t total_to_sample = 0;
    for (auto& kv : st.tid_mapped) {
      if (kv.second == 0) continue;
      float frac = (float)kv.second / total;
      size_t n = std::min<size_t>((size_t)ceilf((float)num_reads * frac), (size_t)kv.second);
      total_to_sample += n; Count c; c.n = n; s.counts[(uint32_t)kv.first] = c;
    }
    if (include_unmapped) { float frac = (float)st.unmapped / total; total_to_sample += (size_t)ceilf((float)num_reads * frac); s.has_unmapped = true;
      }
    size_t floor = 1;
    // pruning iterates an FxHashMap<u32,_>: order here is ascending tid (parity unpinned)
    while ((double)total_to_sample / (double)num_reads > 1.5) {
      for (auto& kv : s.counts) { if (kv.second.n <= floor) { total_to_sample -= kv.second.n; kv.second.n = 0;
        } if (total_to_sample <= num_reads) break; }
      total_to_sample = 0; for (auto& kv : s.counts) total_to_sample += kv.second.n;
      floor++;
    }
    for (auto it = s.counts.begin(); it != s.counts.end();) { if (it->second.n == 0) it = s.counts.erase(it); else ++it; }
    return s;
  }
  static SamplingSchedule from_sample_frac(const IdxStats& st, float f, bool include_unmapped) {  // 321-381
    if (f > 1.0f) throw MkErr("sample fraction must be <= 1");
    SamplingSchedule s; uint64_t total_u = include_unmapped ? st.mapped + st.unmapped : st.mapped;
    if (total_u == 0) throw MkErr("zero reads found in bam index");
    for (auto& kv : st.tid_mapped) {
      if (kv.second == 0) continue;
      Count c; if (f == 1.0f) c.k = Count::ALL; else c.n = (size_t)ceilf((float)kv.second * f);
      s.counts[(uint32_t)kv.first] = c;
    }
    s.has_unmapped = include_unmapped;
    return s;
  }
};

// ReadIdsToBaseModProbs (read_ids_to_base_mod_probs.rs:39-363): read id -> base -> argmax probabilities
struct SampledProbs {
  std::map<std::string, std::map<int, std::vector<float>>> inner;
  std::map<std::string, std::map<int, std::vector<BaseModProbs>>> calls;   // `summary` only: the sampled maps themselves
  void merge(SampledProbs&& o) {  // op_mut 205-213: the first occurrence of a read id wins
    for (auto& kv : o.inner) if (!inner.count(kv.first)) { auto c = o.calls.find(kv.first);
      if (c != o.calls.end()) calls.emplace(kv.first, std::move(c->second));
      inner.emplace(kv.first, std::move(kv.second)); }
  }
  size_t len() const { return inner.size(); }
};

struct SampleCtx { const BamFile* bam; const CollapseMethod* collapse; const EdgeFilter* edge; const PositionFilter* pf; bool only_mapped; bool keep_calls = false; };

// process_records (223-362) over an iterator of records; limit: -1 = passthrough, else first-N
static SampledProbs process_records(const std::vector<const BamRecord*>& recs, long limit, const SampleCtx& cx, StdRng* rng = nullptr,
    double frac = 1.0) {
  SampledProbs out; size_t used = 0;
  for (const BamRecord* rp : recs) {
    const BamRecord& r = *rp;
    // with_mod_base_info (mod_bam.rs:154-199)
    if ((r.flag & (2048 | 256 | 1024)) || r.l_seq == 0) continue;
    ModBaseInfo info;
    try { info = mod_base_info_from_record(r); } catch (const MkErr&) { continue; }
    if (info.is_empty()) continue;
    if ((cx.only_mapped || cx.edge->active) && (r.flag & 4)) continue;
    if (limit >= 0 && used >= (size_t)limit) break;  // RecordSampler::ask -> Done
    if (rng && !rng->gen_bool(frac)) continue;        // check_sample_frac (record_sampler.rs:80-86): one draw per record the iterator yields -> Skip
    std::unordered_map<size_t, uint64_t> pairs;
    if (cx.only_mapped) {
      size_t q = 0; uint64_t rpos = (uint64_t)r.pos; size_t L = (size_t)r.l_seq;
      for (uint32_t c : r.cigar) { int op = c & 15; uint32_t len = c >> 4;
        if (op == 0 || op == 7 || op == 8) { for (uint32_t k = 0; k < len; k++) { size_t qq = q + k;
            if (qq < L) pairs[r.is_reverse() ? L - 1 - qq : qq] = rpos + k;
          } q += len; rpos += len; }
        else if (op == 1 || op == 4) q += len; else if (op == 2 || op == 3) rpos += len; }
    }
    if (out.inner.count(r.qname)) continue;  // seen
    bool added = false;
    for (int s = 0; s < 2; s++) for (auto& kv : (s ? info.neg_strand : info.pos_strand)) {
      int canonical_base = s ? complement(kv.first) : kv.first;
      // filter_positions (966-1070)
      if (cx.edge->active && !cx.edge->read_can_be_trimmed((size_t)r.l_seq)) continue;
      std::vector<float> vals; std::vector<BaseModProbs> kept;
      for (auto& pp : kv.second.pos) {
        bool keep = !cx.edge->active || cx.edge->keep_position(pp.first, (size_t)r.l_seq);
        if (cx.only_mapped && !pairs.count(pp.first)) keep = false;
        if (cx.pf) { auto ap = pairs.find(pp.first); bool ref_neg = (s == 1) != r.is_reverse();
          if (ap == pairs.end() || !cx.pf->contains(r.tid, ap->second, ref_neg)) keep = false;
          }
        if (!keep) continue;
        if (cx.collapse->active) { BaseModProbs c2 = collapse_redistribute(pp.second, cx.collapse->code); vals.push_back(c2.argmax_value());
          if (cx.keep_calls) kept.push_back(c2);
          }
        else { vals.push_back(pp.second.argmax_value()); if (cx.keep_calls) kept.push_back(pp.second); }
      }
      if (vals.empty()) continue;
      auto& dst = out.inner[r.qname][canonical_base]; dst.insert(dst.end(), vals.begin(), vals.end());
      if (cx.keep_calls) { auto& dc = out.calls[r.qname][canonical_base]; dc.insert(dc.end(), kept.begin(), kept.end()); }
      added = true;
    }
    if (added) used++;
  }
  return out;
}
static std::vector<const BamRecord*> fetch(const BamFile& bam, uint32_t tid, uint32_t start, uint32_t end) {
  std::vector<const BamRecord*> v;
  for (auto& r : bam.recs) if (r.tid == (int32_t)tid && (int64_t)r.pos < (int64_t)end && (int64_t)r.end_pos() > (int64_t)start) v.push_back(&r);
  return v;
}

// get_sampled_read_ids_to_base_mod_probs + sample_reads_base_mod_calls_over_regions (reads_sampler/mod.rs:30-257)
// bam::IndexedReader::from_path(..).is_ok() (reads_sampler/mod.rs:47): htslib finds <bam>.bai, <bam>.csi or the extension replaced
static bool bam_has_index(const std::string& path) {
  std::vector<std::string> c = {path + ".bai", path + ".csi"};
  if (path.size() > 4 && path.compare(path.size() - 4, 4, ".bam") == 0) { c.push_back(path.substr(0, path.size() - 4) + ".bai");
    c.push_back(path.substr(0, path.size() - 4) + ".csi"); }
  for (auto& f : c) { FILE* p = fopen(f.c_str(), "rb"); if (p) { fclose(p); return true; } }
  return false;
}
static SampledProbs sample_reads(const BamFile& bam, const Options& o, const Region* region, const CollapseMethod& collapse,
                                 const EdgeFilter& edge, const PositionFilter* pf, bool keep_calls) {
  bool only_mapped = !o.include_unmapped;
  if (o.serial_sampler && !bam_has_index(o.in_bam)) {
    // no index next to the BAM (reads_sampler/mod.rs:129-158): one pass over the file in file order — mapped and unmapped records alike —
    // under RecordSampler::new_from_options: the first --num-reads records that yield values, or a Bernoulli draw per record
    if (region) throw MkErr("cannot use region without indexed BAM");
    SampleCtx cx{&bam, &collapse, &edge, pf, only_mapped, keep_calls};
    std::vector<const BamRecord*> all; all.reserve(bam.recs.size()); for (auto& r : bam.recs) all.push_back(&r);
    const bool draws = o.have_frac && o.sampling_frac < 1.0;
    if (draws && !o.have_seed) throw MkErr("--sampling-frac < 1 on a BAM without an index draws from an entropy-seeded rand::StdRng: give --seed");
    StdRng rng = StdRng::seed_from_u64(o.seed);
    return process_records(all, o.have_frac ? -1 : (long)o.num_reads, cx, draws ? &rng : nullptr, o.sampling_frac);
  }
  IdxStats st = IdxStats::make(bam, region, pf);
  SamplingSchedule sched = o.have_frac ? SamplingSchedule::from_sample_frac(st, (float)o.sampling_frac, !only_mapped)
                                       : SamplingSchedule::from_num_reads(st, o.num_reads, !only_mapped);
  size_t batch_size = (size_t)floorf((float)o.threads * 1.5f);
  std::vector<ReferenceRecord> contigs; for (auto& r : get_targets(bam, region)) if (sched.counts.count(r.tid)) contigs.push_back(r);
  std::map<uint32_t, uint32_t> contig_sizes; for (auto& r : contigs) contig_sizes[r.tid] = r.length;
  SampleCtx cx{&bam, &collapse, &edge, pf, only_mapped, keep_calls};
  SampledProbs agg; std::map<uint32_t, size_t> sampled_per_chr;
  if (!contigs.empty()) {
    Feeder feeder(contigs, batch_size, o.sampling_interval_size, false, nullptr, nullptr);
    std::vector<MultiChromCoordinates> super_batch;
    while (feeder.next_batch(&super_batch)) {
      // accumulate_sample_counts (sampling_schedule.rs:440-615)
      std::vector<ChromCoordinates> all; for (auto& m : super_batch) for (auto& c : m) all.push_back(c);
      std::stable_sort(all.begin(), all.end(), [](auto& a, auto& b) { return a.tid != b.tid ? a.tid < b.tid : a.start < b.start; });
      std::map<uint32_t, uint32_t> len_per_chrom; for (auto& c : all) len_per_chrom[c.tid] += c.len();
      std::map<uint32_t, Count> per_chrom;
      for (auto& kv : len_per_chrom) {
        auto cs = contig_sizes.find(kv.first); if (cs == contig_sizes.end()) continue;
        size_t so_far = sampled_per_chr.count(kv.first) ? sampled_per_chr[kv.first] : 0;
        float f = (float)kv.second / (float)cs->second;
        auto sc = sched.counts.find(kv.first); if (sc == sched.counts.end()) continue;
        if (sc->second.k == Count::ALL) per_chrom[kv.first] = sc->second;
        else if (sc->second.n > so_far) { Count c; c.n = (size_t)ceilf(f * (float)(sc->second.n - so_far)); per_chrom[kv.first] = c; }
      }
      struct G { ChromCoordinates cc; Count c; };
      std::vector<G> grouped; bool have_slack = false; ChromCoordinates slack; size_t slack_n = 0;
      auto merge_cc = [](const ChromCoordinates& a, const ChromCoordinates& b) { ChromCoordinates m = a; m.start = std::min(a.start, b.start);
        m.end = std::max(a.end, b.end); return m; };
      for (auto& cc : all) {
        auto pc = per_chrom.find(cc.tid); if (pc == per_chrom.end()) continue;
        float f = (float)cc.len() / (float)len_per_chrom[cc.tid];
        if (pc->second.k == Count::ALL) { grouped.push_back({cc, pc->second}); continue; }
        size_t x = (size_t)ceilf((float)pc->second.n * f);
        if (x < 50) {
          if (have_slack) {
            if (slack.tid == cc.tid) { ChromCoordinates m = merge_cc(slack, cc); size_t tot = x + slack_n; if (tot < 50) { slack = m; slack_n = tot;
              } else { Count c; c.n = tot; grouped.push_back({m, c}); have_slack = false; } }
            else { Count c; c.n = slack_n; grouped.push_back({slack, c}); slack = cc; slack_n = x; }
          } else { have_slack = true; slack = cc; slack_n = x; }
        } else {
          Count cx2; cx2.n = x;
          if (have_slack) {
            have_slack = false;
            if (slack.tid == cc.tid) { Count c; c.n = slack_n + x; grouped.push_back({merge_cc(slack, cc), c}); }
            else { Count c; c.n = slack_n; grouped.push_back({slack, c}); grouped.push_back({cc, cx2}); }
          } else grouped.push_back({cc, cx2});
        }
      }
      if (have_slack) { Count c; c.n = slack_n; grouped.push_back({slack, c}); }
      // run_batch (reads_sampler/mod.rs:259-338): every (interval, count) is independent
      SampledProbs batch_res; std::map<uint32_t, size_t> batch_counts;
      for (auto& g : grouped) {
        if (!sched.counts.count(g.cc.tid)) continue;
        if (pf && !pf->overlaps_not_stranded(g.cc.tid, g.cc.start, g.cc.end)) continue;
        SampledProbs r = process_records(fetch(bam, g.cc.tid, g.cc.start, g.cc.end), g.c.k == Count::ALL ? -1 : (long)g.c.n, cx);
        batch_counts[g.cc.tid] += r.len();
        batch_res.merge(std::move(r));
      }
      agg.merge(std::move(batch_res));
      for (auto& kv : batch_counts) sampled_per_chr[kv.first] += kv.second;
    }
  }
  if ((sched.has_unmapped || agg.len() < 100) && !only_mapped) {  // 89-125
    std::vector<const BamRecord*> un; for (auto& r : bam.recs) if (r.tid < 0) un.push_back(&r);
    long limit;
    if (!o.have_frac) limit = (long)(o.num_reads > agg.len() ? o.num_reads - agg.len() : 0);
    else if (o.sampling_frac >= 1.0) limit = -1;
    else limit = -1;
    // new_from_options (record_sampler.rs:51-61): --num-reads -> first N; --sampling-frac -> a Bernoulli draw per record from StdRng,
    // seeded by --seed; without --seed the reference seeds from entropy and no two runs agree: refused here as by the product
    const bool draws = o.have_frac && o.sampling_frac < 1.0;
    if (draws
        && !un.empty()
        && !o.have_seed) throw MkErr("unmapped-read sampling with --sampling-frac < 1 draws from an entropy-seeded rand::StdRng: give --seed");
    StdRng rng = StdRng::seed_from_u64(o.seed);
    agg.merge(process_records(un, limit, cx, draws ? &rng : nullptr, o.sampling_frac));
  }
  return agg;
}
static std::map<int, std::vector<float>> flatten_probs(const SampledProbs& agg) {  // mle_probs_per_base 67-101
  std::map<int, std::vector<float>> per_base;
  for (auto& kv : agg.inner) for (auto& bv : kv.second) { auto& d = per_base[bv.first]; d.insert(d.end(), bv.second.begin(), bv.second.end()); }
  return per_base;
}
static std::map<int, std::vector<float>> sample_probs(const BamFile& bam, const Options& o, const Region* region, const CollapseMethod& collapse,
                                                      const EdgeFilter& edge, const PositionFilter* pf) {
  return flatten_probs(sample_reads(bam, o, region, collapse, edge, pf, false));
}

// parse_thresholds / parse_per_base_thresholds (command_utils.rs:47-206)
static void parse_thresholds(const std::vector<std::string>& raws, ThresholdCaller* c) {
  bool have_default = false;
  for (auto& raw : raws) {
    size_t col = raw.find(':');
    if (col != std::string::npos) {
      if (raw.find(':', col + 1) != std::string::npos || col == 0) throw MkErr("illegal per-base threshold " + raw);
      int b = base_from_char(raw[0]); if (b < 0) throw MkErr("failed to parse base");
      if (c->per_base.count(b)) throw MkErr("repeated threshold for base");
      c->per_base[b] = strtof(raw.c_str() + col + 1, nullptr);
    } else { if (have_default) throw MkErr("default threshold encountered more than once"); have_default = true;
      c->default_threshold = strtof(raw.c_str(), nullptr); }
  }
}

// BedMethylWriter::write_feature_counts (writers.rs:87-156)
static void write_rows(FILE* f, const std::string& chrom, const std::map<uint32_t, std::vector<Row>>& rows, bool mixed,
    const std::vector<std::string>& labels, uint64_t* n_rows) {
  char sp = mixed ? ' ' : '\t';
  for (auto& kv : rows) for (const Row& r : kv.second) {
    std::string name = code_str(r.code);
    if (labels.size() >= 2 && r.motif_idx >= 0 && (size_t)r.motif_idx < labels.size()) name += "," + labels[r.motif_idx];
    float pct = r.frac * 100.0f;
    fprintf(f, "%s\t%u\t%u\t%s\t%u\t%c\t%u\t%u\t255,0,0\t%u%c%.2f%c%u%c%u%c%u%c%u%c%u%c%u%c%u\n", chrom.c_str(), r.pos, r.pos + 1, name.c_str(),
        r.cov, r.strand, r.pos,
            r.pos + 1, r.cov, sp, (double)pct, sp, r.n_mod, sp, r.n_can, sp, r.n_other, sp, r.n_delete, sp, r.n_fail, sp, r.n_diff, sp, r.n_nocall);
    (*n_rows)++;
  }
}

// PileupWriter<DuplexModBasePileup> for BedMethylWriter (writers.rs:185-258)
static void write_duplex_rows(FILE* f, const std::string& chrom, const std::map<uint32_t, std::vector<DuplexRow>>& rows, bool mixed,
    uint64_t* n_rows) {
  char sp = mixed ? ' ' : '\t';
  auto el = [](ModCode c) { return c == 0 ? std::string("-") : code_str(c); };
  for (auto& kv : rows) for (const DuplexRow& r : kv.second) {
    uint32_t cov = r.count + r.n_other;
    float pct = ((float)r.count / (float)cov) * 100.0f;
    std::string name = el(r.pat[0]) + "," + el(r.pat[1]) + "," + std::string(1, r.base);
    fprintf(f, "%s\t%u\t%u\t%s\t%u\t.\t%u\t%u\t255,0,0\t%u%c%.2f%c%u%c%u%c%u%c%u%c%u%c%u%c%u\n", chrom.c_str(), r.pos, r.pos + 1, name.c_str(), cov,
        r.pos, r.pos + 1,
            cov, sp, (double)pct, sp, r.count, sp, r.n_can, sp, r.n_other, sp, r.n_delete, sp, r.n_fail, sp, r.n_diff, sp, r.n_nocall);
    (*n_rows)++;
  }
}

static int run_pileup(const Options& o) {
  auto t0 = std::chrono::steady_clock::now();
  BamFile bam = read_bam(o.in_bam);
  auto t_load = std::chrono::steady_clock::now();
  Region region, sregion; bool have_region = !o.region.empty(), have_sregion = !o.sample_region.empty();
  if (have_region) region = parse_region(o.region, bam);
  if (have_sregion) sregion = parse_region(o.sample_region, bam);
  PileupOptions po; po.max_depth = o.max_depth; po.force_allow = o.force_allow;
  if (!o.edge_filter.empty()) {  // parse_edge_filter_input (command_utils.rs:243-277)
    po.edge_filter.active = true; po.edge_filter.inverted = o.invert_edge;
    size_t c = o.edge_filter.find(',');
    if (c != std::string::npos) { po.edge_filter.start = strtoul(o.edge_filter.c_str(), nullptr, 10);
      po.edge_filter.end = strtoul(o.edge_filter.c_str() + c + 1, nullptr, 10); }
    else po.edge_filter.start = po.edge_filter.end = strtoul(o.edge_filter.c_str(), nullptr, 10);
  }
  std::map<ModCode, float> per_mod;
  for (auto& raw : o.mod_thresholds) { size_t c = raw.find(':'); if (c == std::string::npos) throw MkErr("illegal per-mod threshold"); ModCode mc;
    if (!parse_mod_code(raw.substr(0, c), &mc)) throw MkErr("failed to parse mod code");
    per_mod[mc] = strtof(raw.c_str() + c + 1, nullptr); }
  std::vector<ReferenceRecord> reference_records = get_targets(bam, have_region ? &region : nullptr);
  PositionFilter pf_store; const PositionFilter* pf = nullptr;
  if (!o.include_bed.empty()) { std::map<std::string, uint32_t> c2t; for (auto& r : reference_records) c2t[r.name] = r.tid;
    pf_store = PositionFilter::from_bed(o.include_bed, c2t); pf = &pf_store; }
  if (IdxStats::make(bam, have_region ? &region : nullptr, pf).mapped == 0) throw MkErr("did not find any mapped reads");
  size_t chunk_size = o.have_chunk ? o.chunk_size : (size_t)floorf((float)o.threads * 1.5f);
  if (o.filter_percentile > 1.0f) throw MkErr("filter percentile must be <= 1.0");
  if (o.combine_strands && !(o.cpg || !o.motif_parts.empty())) throw MkErr("need to specify either --motif or --cpg to combine strands");
  if (o.hemi) {  // subcommand.rs:1247-1276: one palindromic motif, --cpg xor --motif
    if (!o.cpg && o.motif_parts.empty()) throw MkErr("either --cpg or a --motif must be provided for pileup-hemi");
    if (o.cpg && !o.motif_parts.empty()) throw MkErr("--cpg cannot be used with --motif");
    if (o.motif_parts.size() > 2) throw MkErr("motif arg should be length 2, eg. CG 0");
    if (o.ref_fasta.empty()) throw MkErr("--ref is required");
  }
  CollapseMethod thr_collapse; bool combine_strands = o.combine_strands || o.hemi;  // the feeder runs with combine_strands = true (1383-1390)
  if (o.preset == "traditional") { po.numeric = NUM_COLLAPSE; po.collapse.active = true; po.collapse.code = code_char('h'); combine_strands = true;
    thr_collapse = po.collapse; }
  else if (!o.preset.empty()) throw MkErr("unknown preset");
  else if (o.combine_mods) po.numeric = NUM_COMBINE;
  else if (!o.ignore.empty()) { ModCode mc; if (!parse_mod_code(o.ignore, &mc)) throw MkErr("failed to parse mod code"); po.numeric = NUM_COLLAPSE;
    po.collapse.active = true; po.collapse.code = mc; thr_collapse = po.collapse; }
  po.combine_strands = combine_strands;
  std::vector<Motif> motifs; bool have_motifs = false;
  if (!o.motif_parts.empty()) {  // RegexMotif::from_raw_parts (motif_bed.rs:152-195)
    if (!o.preset.empty()) throw MkErr("cannot use presets and motifs together");
    std::vector<std::string> parts = o.motif_parts;
    if (o.cpg) { bool has = false; for (size_t i = 0; i + 1 < parts.size(); i += 2) if (parts[i] == "CG" && parts[i + 1] == "0") has = true;
      if (!has) { parts.push_back("CG");
        parts.push_back("0"); } }
    for (size_t i = 0; i + 1 < parts.size(); i += 2) motifs.push_back(parse_motif(parts[i], strtoul(parts[i + 1].c_str(), nullptr, 10)));
    have_motifs = true;
  } else if (o.preset == "traditional" || o.cpg) { motifs.push_back(parse_motif("CG", 0)); have_motifs = true; }
  std::vector<std::string> labels; for (auto& m : motifs) labels.push_back(m.label());
  MotifLookup lookup; const MotifLookup* lk = nullptr;
  if (have_motifs) {
    if (o.ref_fasta.empty()) throw MkErr("reference fasta is required for using --motif or --cpg options");
    if (combine_strands) for (auto& m : motifs) if (!m.info.is_palindrome) throw MkErr(o.hemi ? "motif must be palindromic for pileup-hemi"
        : "cannot combine strands with a motif that is not a palindrome");
    lookup.fasta = Fasta::load(o.ref_fasta); lookup.mask = o.mask; lookup.motifs = motifs;
      for (auto& m : motifs) lookup.longest = std::max<uint64_t>(lookup.longest, m.info.length);
    lk = &lookup;
  }
  ThresholdCaller caller; caller.per_mod = per_mod;
  if (!o.filter_threshold.empty()) parse_thresholds(o.filter_threshold, &caller);
  else if (o.no_filtering) { caller.per_mod.clear(); }  // new_passthrough (threshold_mod_caller.rs:16-22)
  else {
    const Region* sr = have_sregion ? &sregion : (have_region ? &region : nullptr);
    auto per_base = sample_probs(bam, o, sr, thr_collapse, po.edge_filter, pf);
    for (auto& kv : per_base) { std::sort(kv.second.begin(), kv.second.end());
      caller.per_base[kv.first] = percentile_linear_interp(kv.second, o.filter_percentile); }
    for (auto& kv : caller.per_base) fprintf(stderr, "[oracle] threshold %c %.9g (n=%zu)\n", base_char(kv.first), (double)kv.second,
        per_base[kv.first].size());
  }
  auto t_thr = std::chrono::steady_clock::now();
  if (pf) reference_records = optimize_reference_records(*pf, reference_records, o.interval_size);
  FILE* out = (o.out_bed.empty() || o.out_bed == "-" || o.out_bed == "stdout") ? stdout : fopen(o.out_bed.c_str(), "w");
  if (!out) throw MkErr("failed to make output file");
  if (o.with_header) fputs("chrom\tchromStart\tchromEnd\tname\tscore\tstrand\tthickStart\tthickEnd\tcolor\tvalid_coverage\tpercent_modified\tcount_modified\tcount_canonical\tcount_other_mod\tcount_delete\tcount_fail\tcount_diff\tcount_nocall\n",
      out);
  uint64_t n_rows = 0, n_positions = 0, n_proc = 0, n_skip = 0;
  if (!reference_records.empty()) {
    Feeder feeder(reference_records, chunk_size, o.interval_size, combine_strands, lk, pf);
    std::vector<MultiChromCoordinates> super_batch;
    while (feeder.next_batch(&super_batch)) {
      std::vector<const ChromCoordinates*> work; for (auto& m : super_batch) for (auto& c : m) work.push_back(&c);
      std::vector<IntervalResult> results(work.size()); std::vector<DuplexIntervalResult> dresults(o.hemi ? work.size() : 0);
        std::vector<std::string> errs(work.size());
      auto job = [&](size_t i) {
        try {
          if (o.hemi) dresults[i] = process_region_duplex(bam, work[i]->tid, work[i]->start, work[i]->end, caller, po, work[i]->focus);
          else results[i] = process_region(bam, work[i]->tid, work[i]->start, work[i]->end, caller, po, work[i]->focus);
        } catch (const MkErr& e) { errs[i] = e.what(); }
      };
      if (o.workers <= 1) for (size_t i = 0; i < work.size(); i++) job(i);
      else { std::vector<std::thread> th; std::atomic<size_t> nxt{0}; for (size_t w = 0; w < o.workers; w++) th.emplace_back([&]() { for (;;) {
            size_t i = nxt++; if (i >= work.size()) break; job(i); } }); for (auto& t : th) t.join(); }
      for (size_t i = 0; i < work.size(); i++) {
        if (!errs[i].empty()) { fprintf(stderr, "[oracle] interval error: %s\n", errs[i].c_str());
          if (errs[i].find("max-depth") != std::string::npos) throw MkErr(errs[i]);
          continue; }
        if (o.hemi) { write_duplex_rows(out, bam.ref_names[work[i]->tid], dresults[i].rows, o.mixed_delim, &n_rows); n_proc += dresults[i].processed;
          n_skip += dresults[i].skipped; }
        else { write_rows(out, bam.ref_names[work[i]->tid], results[i].rows, o.mixed_delim, labels, &n_rows); n_proc += results[i].processed;
          n_skip += results[i].skipped; }
        n_positions += work[i]->len();
      }
    }
  }
  if (out != stdout) fclose(out);
  auto t1 = std::chrono::steady_clock::now();
  auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
  fprintf(stderr, "[oracle] rows=%llu positions=%llu processed~%llu skipped~%llu load_s=%.3f threshold_s=%.3f pileup_s=%.3f total_s=%.3f\n",
      (unsigned long long)n_rows,
          (unsigned long long)n_positions, (unsigned long long)n_proc, (unsigned long long)n_skip, sec(t0, t_load), sec(t_load, t_thr),
              sec(t_thr, t1), sec(t0, t1));
  return 0;
}

// `modkit sample-probs` (src/commands.rs:680-887), the percentile table: sample as the schedule says, per canonical base sort the argmax
// probabilities and interpolate (Percentiles::new -> percentile_linear_interp, src/thresholds.rs:17-38).  One line per (base, percentile):
// base, percentile, value (%.9g), number of values.  The reference prints through prettytable / f32 Display; the numbers are what is pinned.
static int run_sample_probs(const Options& o) {
  BamFile bam = read_bam(o.in_bam);
  Region region; const bool have_region = !o.region.empty();
  if (have_region) region = parse_region(o.region, bam);
  EdgeFilter edge;
  if (!o.edge_filter.empty()) { edge.active = true; edge.inverted = o.invert_edge; size_t c = o.edge_filter.find(','); if (c != std::string::npos) {
      edge.start = strtoul(o.edge_filter.c_str(), nullptr, 10); edge.end = strtoul(o.edge_filter.c_str() + c + 1, nullptr, 10);
    } else edge.start = edge.end = strtoul(o.edge_filter.c_str(), nullptr, 10); }
  CollapseMethod collapse;
  if (!o.ignore.empty()) { ModCode mc; if (!parse_mod_code(o.ignore, &mc)) throw MkErr("failed to parse mod code"); collapse.active = true;
    collapse.code = mc; }
  std::vector<ReferenceRecord> reference_records = get_targets(bam, have_region ? &region : nullptr);
  PositionFilter pf_store; const PositionFilter* pf = nullptr;
  if (!o.include_bed.empty()) { std::map<std::string, uint32_t> c2t; for (auto& r : reference_records) c2t[r.name] = r.tid;
    pf_store = PositionFilter::from_bed(o.include_bed, c2t); pf = &pf_store; }
  auto per_base = sample_probs(bam, o, have_region ? &region : nullptr, collapse, edge, pf);
  std::vector<float> qs; { size_t a = 0; while (a <= o.percentiles.size()) { size_t c = o.percentiles.find(',', a);
      if (c == std::string::npos) c = o.percentiles.size();
      if (c > a) qs.push_back(strtof(o.percentiles.substr(a, c - a).c_str(), nullptr));
      a = c + 1; } }
  FILE* out = (o.out_bed.empty() || o.out_bed == "-") ? stdout : fopen(o.out_bed.c_str(), "w");
  if (!out) throw MkErr("failed to make output file");
  for (auto& kv : per_base) {
    std::sort(kv.second.begin(), kv.second.end());
    for (float q : qs) fprintf(out, "%c\t%.9g\t%.9g\t%zu\n", base_char(kv.first), (double)q, (double)percentile_linear_interp(kv.second, q),
        kv.second.size());
  }
  if (out != stdout) fclose(out);
  return 0;
}

// `modkit summary` (ModSummarize::run, src/commands.rs:1035-1189; summarize_modbam / sampled_reads_to_summary, src/summarize.rs:59-262) as
// counts.  Lines: `total_reads_used N`, `reads_with B N`, `threshold B %.9g`, then `row B code pass fail` per base: canonical ("-") then
// the observed codes in code order.  (The reference's writers iterate std HashMaps and print f32 through Display: not restated.)
static int run_summary(const Options& o) {
  BamFile bam = read_bam(o.in_bam);
  Region region; const bool have_region = !o.region.empty();
  if (have_region) region = parse_region(o.region, bam);
  EdgeFilter edge;
  if (!o.edge_filter.empty()) { edge.active = true; edge.inverted = o.invert_edge; size_t c = o.edge_filter.find(','); if (c != std::string::npos) {
      edge.start = strtoul(o.edge_filter.c_str(), nullptr, 10); edge.end = strtoul(o.edge_filter.c_str() + c + 1, nullptr, 10);
    } else edge.start = edge.end = strtoul(o.edge_filter.c_str(), nullptr, 10); }
  CollapseMethod collapse;
  if (!o.ignore.empty()) { ModCode mc; if (!parse_mod_code(o.ignore, &mc)) throw MkErr("failed to parse mod code"); collapse.active = true;
    collapse.code = mc; }
  std::map<ModCode, float> per_mod;
  for (auto& raw : o.mod_thresholds) { size_t c = raw.find(':'); if (c == std::string::npos) throw MkErr("illegal per-mod threshold"); ModCode mc;
    if (!parse_mod_code(raw.substr(0, c), &mc)) throw MkErr("failed to parse mod code");
    per_mod[mc] = strtof(raw.c_str() + c + 1, nullptr); }
  std::vector<ReferenceRecord> reference_records = get_targets(bam, have_region ? &region : nullptr);
  PositionFilter pf_store; const PositionFilter* pf = nullptr;
  if (!o.include_bed.empty()) { std::map<std::string, uint32_t> c2t; for (auto& r : reference_records) c2t[r.name] = r.tid;
    pf_store = PositionFilter::from_bed(o.include_bed, c2t); pf = &pf_store; }
  SampledProbs agg = sample_reads(bam, o, have_region ? &region : nullptr, collapse, edge, pf, true);
  ThresholdCaller caller;
  if (!o.filter_threshold.empty()) { caller.per_mod = per_mod; parse_thresholds(o.filter_threshold, &caller); }
  else if (o.no_filtering) {}
  else { caller.per_mod = per_mod; auto per_base = flatten_probs(agg); for (auto& kv : per_base) { std::sort(kv.second.begin(), kv.second.end());
      caller.per_base[kv.first] = percentile_linear_interp(kv.second, o.filter_percentile); } }
  // code 0 = canonical
  std::map<int, uint64_t> reads_with; std::map<int, std::map<ModCode, uint64_t>> pass, fail; std::map<int, std::set<ModCode>> observed;
  for (auto& rk : agg.calls) for (auto& bk : rk.second) {
    const int base = bk.first; reads_with[base]++;
    for (const BaseModProbs& bmp : bk.second) {
      const BaseModCall am = argmax_call(bmp), th = caller.call(base, bmp);
      bmp.probs.for_each([&](ModCode c, float) { observed[base].insert(c); });
      if (th.kind == BaseModCall::CANONICAL) pass[base][0]++; else if (th.kind == BaseModCall::MODIFIED) pass[base][th.code]++;
      else if (am.kind == BaseModCall::CANONICAL) fail[base][0]++; else fail[base][am.code]++;
    }
  }
  FILE* out = (o.out_bed.empty() || o.out_bed == "-") ? stdout : fopen(o.out_bed.c_str(), "w");
  if (!out) throw MkErr("failed to make output file");
  fprintf(out, "total_reads_used\t%zu\n", agg.inner.size());
  for (auto& kv : reads_with) fprintf(out, "reads_with\t%c\t%llu\n", base_char(kv.first), (unsigned long long)kv.second);
  for (auto& kv : caller.per_base) fprintf(out, "threshold\t%c\t%.9g\n", base_char(kv.first), (double)kv.second);
  for (auto& kv : reads_with) {
    const int b = kv.first;
    std::vector<ModCode> codes(observed[b].begin(), observed[b].end()); std::sort(codes.begin(), codes.end());
    fprintf(out, "row\t%c\t-\t%llu\t%llu\n", base_char(b), (unsigned long long)pass[b][0], (unsigned long long)fail[b][0]);
    for (ModCode c : codes) fprintf(out, "row\t%c\t%s\t%llu\t%llu\n", base_char(b), code_str(c).c_str(), (unsigned long long)pass[b][c],
        (unsigned long long)fail[b][c]);
  }
  if (out != stdout) fclose(out);
  return 0;
}

// `modkit extract calls` (EntryExtractCalls::run, src/extract/subcommand.rs:452-761), the serial file-order path
static int run_extract_calls(const Options& o, const ExtractOptions& xo_in) {
  ExtractOptions xo = xo_in;
  BamFile bam = read_bam(o.in_bam);
  Region region; const bool have_region = !o.region.empty();
  if (have_region) region = parse_region(o.region, bam);
  PositionFilter pf_store; const PositionFilter* pf = nullptr;
  if (!o.include_bed.empty()) { std::map<std::string, uint32_t> c2t;
    for (auto& rr : get_targets(bam, nullptr)) c2t[rr.name] = rr.tid;   // name_to_tid of the whole header (subcommand.rs:514-517), not of --region
    pf_store = PositionFilter::from_bed(o.include_bed, c2t); pf = &pf_store;
    xo.include = [pf](int32_t tid, uint64_t p, bool neg) { return pf->contains(tid, p, neg); }; }
  // --motif / --cpg (load_regions, util.rs:157-277): the include filter becomes the motif hits over every contig of the FASTA that the header
  // names (the whole sequence, upper-cased unless --mask), one position per hit and strand, intersected with the --include-bed positions when
  // both are given; from there on it IS the include filter — rows, the estimate, the schedule ("outputs only mapped sites")
  if (o.cpg || !o.motif_parts.empty()) {
    if (xo.ref_fasta.empty()) throw MkErr("--motif / --cpg need --ref");
    std::vector<std::string> parts = o.motif_parts;   // RegexMotif::from_raw_parts (motif_bed.rs:152-195)
    if (o.cpg) { bool has = false; for (size_t i = 0; i + 1 < parts.size(); i += 2) if (parts[i] == "CG" && parts[i + 1] == "0") has = true;
      if (!has) { parts.push_back("CG"); parts.push_back("0"); } }
    std::vector<Motif> motifs;
    for (size_t i = 0; i + 1 < parts.size(); i += 2) motifs.push_back(parse_motif(parts[i], strtoul(parts[i + 1].c_str(), nullptr, 10)));
    Fasta fa = Fasta::load(xo.ref_fasta);
    PositionFilter mf;
    for (auto& kv : fa.seqs) {
      const int tid = bam.tid_of(kv.first); if (tid < 0) continue;
      std::string seq = kv.second; if (!o.mask) for (char& c : seq) c = (char)toupper((unsigned char)c);
      auto& P = mf.pos[(uint32_t)tid]; auto& N = mf.neg[(uint32_t)tid];   // (a searched contig is in the filter even without a hit)
      for (auto& m : motifs) for (auto& h : find_motif_hits(seq, m)) {
        if (pf && !pf->contains(tid, h.first, h.second)) continue;
        (h.second ? N : P).push_back({(uint64_t)h.first, (uint64_t)h.first + 1});
      }
      lapper_merge(P); lapper_merge(N);
    }
    pf_store = std::move(mf); pf = &pf_store;
    xo.include = [pf](int32_t tid, uint64_t p, bool neg) { return pf->contains(tid, p, neg); };
  }
  // --exclude-bed (load_regions, util.rs:177-187): a row filter only — neither the estimate nor the schedule looks at it
  PositionFilter ex_store;
  if (!xo.exclude_bed.empty()) { std::map<std::string, uint32_t> c2t; for (auto& rr : get_targets(bam, nullptr)) c2t[rr.name] = rr.tid;
    ex_store = PositionFilter::from_bed(xo.exclude_bed, c2t); const PositionFilter* ex = &ex_store;
    xo.exclude = [ex](int32_t tid, uint64_t p, bool neg) { return ex->contains(tid, p, neg); }; }
  // With an index (and without --ignore-index) the reference walks interval chunks of the targets (util.rs:329-470): --region then selects the
  // records the fetches of its intervals return — every record overlapping it, once (prev_end) — where the serial scan looks at every record
  // of the file.  Its rows leave in whatever order the pool finishes the intervals; here: file order.  --num-reads with an index goes through
  // the sampling schedule (`scheduled` below).
  FILE* probe = fopen((o.in_bam + ".bai").c_str(), "rb"); const bool use_index = probe && !xo.ignore_index; if (probe) fclose(probe);
  // --num-reads with an index: the sampling schedule per interval (below); with --include-bed over the BED-optimised reference records
  const bool scheduled = use_index && xo.num_reads >= 0;
  if (!use_index) xo.remove_inferred = false;   // process_records_to_chan (the serial scan) never removes inferred calls (util.rs:501-575)
  const int region_tid = have_region ? bam.tid_of(region.name) : -1;
  EdgeFilter edge;
  if (!o.edge_filter.empty()) { edge.active = true; edge.inverted = o.invert_edge; size_t c = o.edge_filter.find(','); if (c != std::string::npos) {
      edge.start = strtoul(o.edge_filter.c_str(), nullptr, 10); edge.end = strtoul(o.edge_filter.c_str() + c + 1, nullptr, 10);
    } else edge.start = edge.end = strtoul(o.edge_filter.c_str(), nullptr, 10); }
  CollapseMethod collapse;
  if (!o.ignore.empty()) { ModCode mc; if (!parse_mod_code(o.ignore, &mc)) throw MkErr("failed to parse mod code"); collapse.active = true;
    collapse.code = mc; }
  std::map<ModCode, float> per_mod;
  for (auto& raw : o.mod_thresholds) { size_t c = raw.find(':'); if (c == std::string::npos) throw MkErr("illegal per-mod threshold"); ModCode mc;
    if (!parse_mod_code(raw.substr(0, c), &mc)) throw MkErr("failed to parse mod code");
    per_mod[mc] = strtof(raw.c_str() + c + 1, nullptr); }
  ThresholdCaller caller;
  if (o.no_filtering) {}
  else if (!o.filter_threshold.empty()) { caller.per_mod = per_mod; parse_thresholds(o.filter_threshold, &caller); }
  // get_threshold_from_options (command_utils.rs:74-134): the pileup's estimate; positions without a reference position count unless --mapped-only
  else {
    Options so = o; so.include_unmapped = !xo.mapped_only && !pf;   // reference_position_filter.only_mapped_positions()
    caller.per_mod = per_mod;
    auto per_base = sample_probs(bam, so, have_region ? &region : nullptr, collapse, edge, pf);
    for (auto& kv : per_base) { std::sort(kv.second.begin(), kv.second.end());
      caller.per_base[kv.first] = percentile_linear_interp(kv.second, o.filter_percentile); }
  }
  std::map<std::string, std::string> ref_seqs;
  if (!xo.ref_fasta.empty()) { Fasta fa = Fasta::load(xo.ref_fasta);
    for (auto& kv : fa.seqs) if (bam.tid_of(kv.first) >= 0) ref_seqs[kv.first] = kv.second;
    }
  FILE* out = (xo.out_tsv.empty() || xo.out_tsv == "-" || xo.out_tsv == "stdout") ? stdout : fopen(xo.out_tsv.c_str(), "w");
  if (!out) throw MkErr("failed to make output file");
  if (!xo.no_headers) fputs(extract_calls_header(), out);
  uint64_t n_used = 0, n_skipped = 0, n_failed = 0, n_rows = 0;
  long n_sent = 0;
  if (scheduled) {
    // run_extract_reads with an index and --num-reads (src/extract/util.rs:329-470, subcommand.rs:662-683): SamplingSchedule::from_num_reads
    // over the index counts; every interval of the feeder (threads * 1.5 groups per super batch, no BED here) gets a RecordSampler of its
    // own — ceil(chrom count * interval length / length of the whole super batch) records (get_record_sampler, sampling_schedule.rs:417-438) —
    // and takes, of the records its fetch returns that do not start in front of the previous interval's end (`cut`), the first that many
    // whose process_record succeeds (ReadsBaseModProfile::process_records, read_ids_to_base_mod_probs.rs:884-945).  Then, unless a region /
    // --mapped-only excludes them, the records without coordinates: the first (N - used) that reach process_record.  The reference's
    // rows leave in pool order; here in interval order.
    const bool include_unmapped_reads = !have_region && !xo.mapped_only && !pf;   // load_regions (util.rs:136-155), no BED / motif here
    IdxStats st = IdxStats::make(bam, have_region ? &region : nullptr, pf);
    SamplingSchedule sched = SamplingSchedule::from_num_reads(st, (size_t)xo.num_reads, include_unmapped_reads);
    std::vector<ReferenceRecord> sched_records = get_targets(bam, have_region ? &region : nullptr);
    if (pf) sched_records = optimize_reference_records(*pf, sched_records, o.interval_size);   // load_regions (util.rs:287-296)
    Feeder feeder(sched_records, (size_t)floorf((float)o.threads * 1.5f), o.interval_size, false, nullptr, nullptr);
    ExtractOptions xs = xo; xs.ask_unmapped = true;   // (this path never drops a record for being unmapped: its rows go with the position filter)
    std::vector<MultiChromCoordinates> super_batch; bool have_prev = false; uint32_t prev_tid = 0, prev_end = 0; size_t aligned_used = 0;
    auto emit = [&](const std::string& rows) { n_used++; for (char c : rows) if (c == '\n') n_rows++; fputs(rows.c_str(), out); };
    while (feeder.next_batch(&super_batch)) {
      uint64_t total_len = 0; for (auto& m : super_batch) for (auto& c : m) total_len += c.len();
      for (auto& m : super_batch) for (auto& cc : m) {
        const bool cut = have_prev && prev_tid == cc.tid; const uint32_t cut_at = prev_end;
        have_prev = true; prev_tid = cc.tid; prev_end = cc.end;
        auto sc = sched.counts.find(cc.tid); if (sc == sched.counts.end()) continue;   // chrom_has_reads
        long nr = -1;
        if (sc->second.k == Count::COUNT) nr = (long)ceil((double)sc->second.n * ((double)(cc.end - cc.start) / (double)(uint32_t)total_len));
        long used = 0;
        for (const BamRecord* rp : fetch(bam, cc.tid, cc.start, cc.end)) {
          const BamRecord& r = *rp;
          if (cut && (int64_t)r.pos < (int64_t)cut_at) continue;
          std::string rows; bool skipped = false, sent = false;
          const bool ok = extract_calls_of_record(bam, r, xs, collapse, edge, caller, ref_seqs, &rows, &skipped, &sent);
          if (!sent) { if (!ok) n_failed++; else n_skipped++; continue; }   // TrackingModRecordIter never offers it
          if (nr >= 0 && used >= nr) break;                                   // RecordSampler::ask -> Done
          if (!ok) { n_failed++; continue; }
          used++; aligned_used++;
          if (xo.mapped_only && (r.flag & 4)) continue;                       // every row of an unmapped record lacks a reference position
          emit(rows);
        }
      }
    }
    if (include_unmapped_reads) {
      const long n_un = (long)((size_t)xo.num_reads > aligned_used ? (size_t)xo.num_reads - aligned_used : 0);
      ExtractOptions xu = xo; xu.allow_non_primary = false; xu.mapped_only = false;
      for (const BamRecord& r : bam.recs) {
        if (r.tid >= 0) continue;
        std::string rows; bool skipped = false, sent = false;
        const bool ok = extract_calls_of_record(bam, r, xu, collapse, edge, caller, ref_seqs, &rows, &skipped, &sent);
        if (sent) n_sent++;
        // process_records_to_chan (util.rs:519-575) looks at its count AFTER a record went to the writer: N = 0 still lets one through
        const bool done = sent && n_sent >= n_un;
        if (!ok) n_failed++; else if (skipped) n_skipped++; else emit(rows);
        if (done) break;
      }
    }
    if (out != stdout) fclose(out);
    fprintf(stderr, "[oracle] extract calls (scheduled): reads=%llu rows=%llu skipped=%llu failed=%llu\n", (unsigned long long)n_used,
        (unsigned long long)n_rows, (unsigned long long)n_skipped, (unsigned long long)n_failed);
    return 0;
  }
  for (const BamRecord& r : bam.recs) {
    // IndexedReader::fetch(tid, start, end): records overlapping the region (a record without reference span counts as one base)
    if (use_index && have_region) {
      const int64_t e = (int64_t)r.pos + std::max<int64_t>((int64_t)r.ref_len(), 1);
      if (r.tid != region_tid || r.pos >= (int64_t)region.end || e <= (int64_t)region.start) continue;
    }
    std::string rows; bool skipped = false, sent = false;
    const bool ok = extract_calls_of_record(bam, r, xo, collapse, edge, caller, ref_seqs, &rows, &skipped, &sent);
    if (sent) n_sent++;
    const bool done = xo.num_reads >= 0 && sent && n_sent >= xo.num_reads;   // process_records_to_chan: stop once N records went to the writer
    if (!ok) { n_failed++; if (done) break; continue; }
    if (done) { if (skipped) n_skipped++; else { n_used++; for (char c : rows) if (c == '\n') n_rows++; fputs(rows.c_str(), out); } break; }
    if (skipped) { n_skipped++; continue; }
    n_used++; for (char c : rows) if (c == '\n') n_rows++;
    fputs(rows.c_str(), out);
  }
  if (out != stdout) fclose(out);
  fprintf(stderr, "[oracle] extract calls: reads=%llu rows=%llu skipped=%llu failed=%llu\n", (unsigned long long)n_used, (unsigned long long)n_rows,
      (unsigned long long)n_skipped, (unsigned long long)n_failed);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2 || (std::string(argv[1]) != "pileup" && std::string(argv[1]) != "pileup-hemi" && std::string(argv[1]) != "sample-probs"
      && std::string(argv[1]) != "summary" && std::string(argv[1]) != "extract-calls")) {
    fprintf(stderr,
        "usage: modkit_oracle pileup <in.bam> <out.bed> [flags as `modkit pileup`]\n       modkit_oracle pileup-hemi <in.bam> -o <out.bed> [flags as `modkit pileup-hemi`]\n"
                    "       modkit_oracle sample-probs <in.bam> [-o table.tsv] [-p 0.1,0.5,0.9] [sampling flags of `modkit sample-probs`]\n"
                    "       modkit_oracle summary <in.bam> [-o counts.tsv] [flags of `modkit summary`]\n"
                    "       modkit_oracle extract-calls <in.bam> <out.tsv> [--ref fa] [--allow-non-primary] [--mapped-only] [--pass-only] [threshold / sampling flags of `modkit extract calls`]\n");
    return 2;
  }
  Options o; std::vector<std::string> pos;
  const bool extract_cmd = std::string(argv[1]) == "extract-calls"; ExtractOptions xo;
  o.hemi = std::string(argv[1]) == "pileup-hemi";
  const bool summary_cmd = std::string(argv[1]) == "summary";
  o.sample_probs_cmd = std::string(argv[1]) == "sample-probs" || summary_cmd;
  // (`pileup` needs an index in the reference; here it keeps the schedule over scanned counts)
  o.serial_sampler = o.sample_probs_cmd || extract_cmd;
  // --only-mapped is off by default; -i is the sampling interval
  if (o.sample_probs_cmd) { o.include_unmapped = true; o.sampling_interval_size = 1000000; }
  try {
    for (int i = 2; i < argc; i++) {
      std::string a = argv[i];
      auto val = [&]() { if (i + 1 >= argc) throw MkErr("missing value for " + a); return std::string(argv[++i]); };
      if (o.hemi && (a == "--preset" || a == "--combine-strands" || a == "--with-header"
          || a == "--header")) throw MkErr("unknown flag " + a + " for pileup-hemi");
      if (o.hemi && (a == "-o" || a == "--out-bed")) { o.out_bed = val(); continue; }
      if (extract_cmd) {
        if (a == "--ref" || a == "--reference") { xo.ref_fasta = val(); continue; }
        if (a == "--allow-non-primary") { xo.allow_non_primary = true; continue; }
        if (a == "--mapped-only") { xo.mapped_only = true; continue; }
        if (a == "--pass-only" || a == "--pass") { xo.pass_only = true; continue; }
        if (a == "--no-headers") { xo.no_headers = true; continue; }
        if (a == "--kmer-size") { xo.kmer_size = std::stoul(val()); continue; }
        if (a == "--num-reads") { xo.num_reads = std::stol(val()); continue; }
        if (a == "--ignore-index") { xo.ignore_index = true; continue; }
        if (a == "--exclude-bed" || a == "-v" || a == "--exclude-positions") { xo.exclude_bed = val(); continue; }
        // (only the interval path looks at it: cleared below without an index)
        if (a == "--ignore-implicit") { xo.remove_inferred = true; continue; }
        if (a == "--force") continue;
      }
      if (o.sample_probs_cmd) {
        if (a == "-o") { o.out_bed = val(); continue; }
        if (!summary_cmd && (a == "-p" || a == "--percentiles")) { o.percentiles = val(); continue; }
        if (summary_cmd && (a == "--tsv" || a == "--table")) continue;
        if (a == "-i" || a == "--interval-size") { o.sampling_interval_size = (uint32_t)std::stoul(val()); continue; }
        if (a == "--only-mapped") { o.include_unmapped = false; continue; }
        if (a == "--no-sampling") { o.have_frac = true; o.sampling_frac = 1.0; continue; }
      }
      if (a == "--region") o.region = val(); else if (a == "--max-depth") o.max_depth = (uint32_t)std::stoul(val());
      else if (a == "-t" || a == "--threads") o.threads = std::stoul(val());
        else if (a == "-i" || a == "--interval-size") o.interval_size = (uint32_t)std::stoul(val());
      else if (a == "--chunk-size") { o.have_chunk = true; o.chunk_size = std::stoul(val()); }
      else if (a == "-n" || a == "--num-reads") o.num_reads = std::stoul(val()); else if (a == "-f" || a == "--sampling-frac") { o.have_frac = true;
        o.sampling_frac = std::stod(val()); }
      else if (a == "--seed") { o.have_seed = true; o.seed = std::stoull(val()); } else if (a == "--no-filtering") o.no_filtering = true;
        else if (a == "-p" || a == "--filter-percentile") o.filter_percentile = std::stof(val());
      else if (a == "--filter-threshold") o.filter_threshold.push_back(val());
        else if (a == "--mod-thresholds" || a == "--mod-threshold") o.mod_thresholds.push_back(val());
      else if (a == "--sample-region") o.sample_region = val();
        else if (a == "--sampling-interval-size") o.sampling_interval_size = (uint32_t)std::stoul(val());
      else if (a == "--include-bed" || a == "--include-positions") o.include_bed = val();
        else if (a == "--include-unmapped") o.include_unmapped = true;
      else if (a == "--ignore") o.ignore = val(); else if (a == "--force-allow-implicit") o.force_allow = true;
      else if (a == "--motif") { o.motif_parts.push_back(val()); o.motif_parts.push_back(val()); } else if (a == "--cpg") o.cpg = true;
      else if (a == "--ref" || a == "-r") o.ref_fasta = val(); else if (a == "--mask" || a == "-k") o.mask = true;
        else if (a == "--preset") o.preset = val();
      else if (a == "--combine-mods") o.combine_mods = true; else if (a == "--combine-strands") o.combine_strands = true;
      else if (a == "--edge-filter") o.edge_filter = val(); else if (a == "--invert-edge-filter") o.invert_edge = true;
      else if (a == "--only-tabs") {} else if (a == "--mixed-delim") o.mixed_delim = true;
        else if (a == "--with-header" || a == "--header") o.with_header = true;
      else if (a == "--suppress-progress") {} else if (a == "--oracle-workers") o.workers = std::stoul(val());
      else if (a == "--partition-tag" || a == "--bedgraph" || a == "--prefix") throw MkErr(a + " is not restated by the oracle");
      else if (!a.empty() && a[0] == '-' && a != "-") throw MkErr("unknown flag " + a);
      else pos.push_back(a);
    }
    if (extract_cmd) { if (pos.size() != 2) throw MkErr("need <in.bam> <out.tsv>"); o.in_bam = pos[0]; xo.in_bam = pos[0]; xo.out_tsv = pos[1];
      return run_extract_calls(o, xo); }
    if (o.sample_probs_cmd) { if (pos.size() != 1) throw MkErr("need <in.bam>"); o.in_bam = pos[0];
      if (!o.include_bed.empty()) o.include_unmapped = false;
      return summary_cmd ? run_summary(o) : run_sample_probs(o); }
    if (o.hemi) { if (pos.size() != 1) throw MkErr("need <in.bam>"); o.in_bam = pos[0]; }
    else { if (pos.size() != 2) throw MkErr("need <in.bam> <out.bed>"); o.in_bam = pos[0]; o.out_bed = pos[1]; }
    return run_pileup(o);
  } catch (const std::exception& e) { fprintf(stderr, "Error! %s\n", e.what()); return 1; }
}
