// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_core.hpp header).
// Per-interval pileup: ReadCache, htslib pileup columns, tallies, row decode,
// strand combine.  Follows /root/reference/src/read_cache.rs and src/pileup/mod.rs.
#pragma once
#include <queue>
#include "oracle_core.hpp"
#include <tuple>

namespace mko {

enum StrandRule { RULE_POS = 0, RULE_NEG = 1, RULE_BOTH = 2 };  // util.rs:297-307
static inline StrandRule rule_combine(StrandRule a, StrandRule b) { return a == b ? a : RULE_BOTH; }  // util.rs:343-349
static inline StrandRule rule_absorb(StrandRule a, bool neg_strand) {  // util.rs:333-341
  if ((a == RULE_POS && !neg_strand) || (a == RULE_NEG && neg_strand)) return a;
  return RULE_BOTH;
}

struct MotifInfo {  // find_motifs/motif_bed.rs:111-141
  size_t forward_offset = 0, reverse_offset = 0, length = 0;
  bool is_palindrome = false;
  bool negative_strand_position(uint32_t p, uint32_t* out) const {
    if (!is_palindrome) return false;
    int64_t adj = (int64_t)p + ((int64_t)reverse_offset - (int64_t)forward_offset);
    if (adj < 0) return false;
    *out = (uint32_t)adj;
    return true;
  }
};

struct Iv { uint64_t start, stop; };
// rust-lapper 1.1: Lapper::new sorts; merge_overlaps merges when top.stop >= next.start
static inline void lapper_merge(std::vector<Iv>& v) {
  std::sort(v.begin(), v.end(), [](const Iv& a, const Iv& b) { return a.start != b.start ? a.start < b.start : a.stop < b.stop; });
  std::vector<Iv> out;
  for (const Iv& iv : v) {
    if (out.empty()) { out.push_back(iv); continue; }
    Iv& top = out.back();
    if (top.stop < iv.start) out.push_back(iv);
    else if (top.stop < iv.stop) top.stop = iv.stop;
  }
  v.swap(out);
}
static inline bool lapper_any(const std::vector<Iv>& v, uint64_t start, uint64_t stop) {  // find(): iv.start < stop && iv.stop > start
  for (const Iv& iv : v) if (iv.start < stop && iv.stop > start) return true;
  return false;
}

// FocusPositions (interval_chunks.rs:32-59)
struct FocusPositions {
  enum Kind { ALL, MOTIF, MOTIF_COMBINE, REGIONS } kind = ALL;
  std::map<uint32_t, StrandRule> positions;
  std::map<uint32_t, std::vector<size_t>> positive_motif_ids, negative_motif_ids;
  std::map<uint32_t, std::vector<std::pair<MotifInfo, size_t>>> positive_motifs;  // MOTIF_COMBINE
  std::vector<Iv> pos_intervals, neg_intervals;                                    // REGIONS
  bool check_position(uint32_t pos, StrandRule* rule) const {  // 352-374
    switch (kind) {
      case ALL: *rule = RULE_BOTH; return true;
      case MOTIF: case MOTIF_COMBINE: {
        auto it = positions.find(pos);
        if (it == positions.end()) return false;
        *rule = it->second; return true;
      }
      case REGIONS: {
        bool p = lapper_any(pos_intervals, pos, (uint64_t)pos + 1), n = lapper_any(neg_intervals, pos, (uint64_t)pos + 1);
        if (p && n) *rule = RULE_BOTH; else if (p) *rule = RULE_POS; else if (n) *rule = RULE_NEG; else return false;
        return true;
      }
    }
    return false;
  }
  const std::vector<size_t>* get_positive_ids(uint32_t pos, std::vector<size_t>* tmp) const {  // 376-393
    if (kind == MOTIF) { auto it = positive_motif_ids.find(pos); return it == positive_motif_ids.end() ? nullptr : &it->second; }
    if (kind == MOTIF_COMBINE) {
      auto it = positive_motifs.find(pos);
      if (it == positive_motifs.end()) return nullptr;
      tmp->clear(); for (auto& m : it->second) tmp->push_back(m.second);
      return tmp;
    }
    return nullptr;
  }
  const std::vector<size_t>* get_negative_ids(uint32_t pos) const {  // 395-408
    if (kind == MOTIF || kind == MOTIF_COMBINE) { auto it = negative_motif_ids.find(pos);
      return it == negative_motif_ids.end() ? nullptr : &it->second; }
    return nullptr;
  }
};

enum NumericMode { NUM_PASSTHROUGH, NUM_COMBINE, NUM_COLLAPSE };  // pileup/mod.rs:667-671

struct PileupOptions {
  NumericMode numeric = NUM_PASSTHROUGH;
  CollapseMethod collapse;       // for NUM_COLLAPSE
  bool force_allow = false;
  bool combine_strands = false;
  uint32_t max_depth = 8000;
  EdgeFilter edge_filter;
};

struct Row {  // PileupFeatureCounts (pileup/mod.rs:54-68)
  uint32_t pos = 0;
  char strand = '+';
  ModCode code = 0;
  int motif_idx = -1;
  uint32_t cov = 0, n_can = 0, n_mod = 0, n_other = 0, n_delete = 0, n_fail = 0, n_diff = 0, n_nocall = 0;
  float frac = 0.0f;
};

// Tally (pileup/mod.rs:167-224)
struct Tally {
  uint32_t n_delete = 0, n_filtered = 0;
  uint32_t basecall[4] = {0, 0, 0, 0};
  bool has_modcall[4] = {false, false, false, false};
  uint32_t canonical[4] = {0, 0, 0, 0};
  std::map<ModCode, uint32_t> modified[4];
  uint32_t diff_calls_count(int pb) const {  // 198-223
    uint32_t n = 0;
    for (int b = 0; b < 4; b++) if (b != pb) {
      n += basecall[b];
      n += canonical[b];
      for (auto& kv : modified[b]) n += kv.second;
    }
    return n;
  }
};

struct ReadCacheEntry {
  // [mod strand][read base] -> ref pos -> call   (read_cache.rs:28-29)
  bool have_pos = false, have_neg = false;
  std::unordered_map<uint64_t, BaseModCall> calls[2][4];
  bool have_calls[2][4] = {{false, false, false, false}, {false, false, false, false}};
  // [tally strand][threshold base] -> codes      (read_cache.rs:34-35)
  bool have_pos_codes = false, have_neg_codes = false;
  std::set<ModCode> codes[2][4];
};

struct ReadCache {  // read_cache.rs:24-43
  const ThresholdCaller* caller;
  const PileupOptions* opts;
  std::unordered_map<std::string, ReadCacheEntry> reads;
  std::unordered_set<std::string> skip_set;

  // add_record (read_cache.rs:111-211)
  void add_record(const BamRecord& r) {
    ModBaseInfo info = mod_base_info_from_record(r);
    if (info.is_empty()) throw MkErr("no-modbase-info");
    for (int s = 0; s < 2; s++)
      for (auto& kv : (s ? info.neg_strand : info.pos_strand))
        if (kv.second.skip_mode == MODE_DEFAULT_IMPLICIT && !opts->force_allow) throw MkErr("invalid-implicit-mode");
    // aligned pairs, forward-oriented (util.rs:122-145)
    std::unordered_map<size_t, uint64_t> pairs;
    {
      size_t q = 0; uint64_t rp = (uint64_t)r.pos; size_t L = (size_t)r.l_seq;
      for (uint32_t c : r.cigar) {
        int op = c & 15; uint32_t len = c >> 4;
        if (op == 0 || op == 7 || op == 8) {
          for (uint32_t k = 0; k < len; k++) { size_t qq = q + k; if (qq < L) pairs[r.is_reverse() ? L - 1 - qq : qq] = rp + k; }
          q += len; rp += len;
        } else if (op == 1 || op == 4) q += len;
        else if (op == 2 || op == 3) rp += len;
      }
    }
    bool added = false;
    ReadCacheEntry entry;
    for (int s = 0; s < 2; s++) {
      for (auto& kv : (s ? info.neg_strand : info.pos_strand)) {
        int dna_base = kv.first;
        int threshold_base = s ? complement(dna_base) : dna_base;  // 147-150
        SeqPosBaseModProbs probs = kv.second;
        if (opts->edge_filter.active) {  // edge_filter_positions (mod_bam.rs:1075-1102)
          if (!opts->edge_filter.read_can_be_trimmed((size_t)r.l_seq)) continue;
          std::map<size_t, BaseModProbs> kept;
          for (auto& pp : probs.pos) if (opts->edge_filter.keep_position(pp.first, (size_t)r.l_seq)) kept.insert(pp);
          if (kept.empty()) continue;
          probs.pos.swap(kept);
          probs.skip_mode = MODE_EXPLICIT;
        }
        if (opts->numeric == NUM_COLLAPSE && opts->collapse.active)
          for (auto& pp : probs.pos) pp.second = collapse_redistribute(pp.second, opts->collapse.code);
        bool tally_neg = (s == 1) != r.is_reverse();  // 181-188
        std::set<ModCode>& cs = entry.codes[tally_neg ? 1 : 0][threshold_base];
        for (auto& pp : probs.pos) pp.second.probs.for_each([&](ModCode c, float) { cs.insert(c); });
        if (tally_neg) entry.have_neg_codes = true; else entry.have_pos_codes = true;
        // add_modbase_probs_for_record_and_canonical_base (69-108)
        auto& dest = entry.calls[s][dna_base];
        entry.have_calls[s][dna_base] = true;
        for (auto& pp : probs.pos) {
          auto ap = pairs.find(pp.first);
          if (ap != pairs.end()) dest[ap->second] = caller->call(threshold_base, pp.second);
        }
        if (s) entry.have_neg = true; else entry.have_pos = true;
        added = true;
      }
    }
    if (!added) throw MkErr("no-modbase-info");
    // keyed by read NAME: a second record with the same name never reaches here
    reads.emplace(r.qname, std::move(entry));
  }
  // returns entry or nullptr (skip set / failed)
  const ReadCacheEntry* ensure(const BamRecord& r) {
    if (skip_set.count(r.qname)) return nullptr;
    auto it = reads.find(r.qname);
    if (it != reads.end()) return &it->second;
    try { add_record(r); } catch (const MkErr&) { skip_set.insert(r.qname); return nullptr; }
    return &reads.find(r.qname)->second;
  }
};

struct IntervalResult {
  std::map<uint32_t, std::vector<Row>> rows;  // position -> rows (writer sorts by position)
  size_t processed = 0, skipped = 0;
};

// FeatureVector::add_tally_to_counts / decode (pileup/mod.rs:283-446)
static inline void add_tally_to_counts(std::vector<Row>& counts, const Tally& t, char strand,
                                       const std::set<ModCode> observed[4], const PileupOptions& o,
                                       const std::vector<size_t>* motif_idxs, uint32_t pos) {
  for (int pb = 0; pb < 4; pb++) {  // FxHashMap<DnaBase,..> iterates A,C,G,T (hash = d*K, 4 buckets)
    if (!t.has_modcall[pb]) continue;
    uint32_t n_nocall = t.basecall[pb];
    uint32_t n_can = t.canonical[pb];
    uint32_t total_mod = 0;
    for (auto& kv : t.modified[pb]) total_mod += kv.second;
    uint32_t cov = total_mod + n_can;
    auto emit = [&](ModCode code, uint32_t n_mod, uint32_t n_other) {
      Row r;
      r.pos = pos; r.strand = strand; r.code = code; r.cov = cov; r.n_can = n_can; r.n_mod = n_mod; r.n_other = n_other;
      r.n_delete = t.n_delete; r.n_fail = t.n_filtered; r.n_diff = t.diff_calls_count(pb); r.n_nocall = n_nocall;
      r.frac = (float)n_mod / (float)cov;
      if (motif_idxs) for (size_t idx : *motif_idxs) { r.motif_idx = (int)idx; counts.push_back(r); }
      else { r.motif_idx = -1; counts.push_back(r); }
    };
    if (o.numeric == NUM_COMBINE) {
      emit(code_char(base_char(pb)), total_mod, 0);
    } else {
      for (ModCode code : observed[pb]) {  // HashSet order is random in the reference; rows are sorted afterwards
        auto it = t.modified[pb].find(code);
        uint32_t n_mod = it == t.modified[pb].end() ? 0 : it->second;
        emit(code, n_mod, total_mod >= n_mod ? total_mod - n_mod : 0);
      }
    }
  }
}

// combine_strand_features (pileup/mod.rs:469-561)
static inline std::map<uint32_t, std::vector<Row>> combine_strand_features(
    const std::map<uint32_t, std::vector<std::pair<MotifInfo, size_t>>>& positive_motifs,
    const std::map<uint32_t, std::vector<Row>>& pfc) {
  std::map<uint32_t, std::vector<Row>> result;
  for (auto& pm : positive_motifs) {
    uint32_t ppos = pm.first;
    auto pit = pfc.find(ppos);
    for (auto& mi : pm.second) {
      uint32_t npos;
      if (!mi.first.negative_strand_position(ppos, &npos)) continue;
      auto nit = pfc.find(npos);
      if (pit == pfc.end() && nit == pfc.end()) continue;  // no partition keys at all
      std::map<ModCode, std::vector<Row>> grouped;           // BTreeMap: derived Ord
      if (pit != pfc.end()) for (const Row& r : pit->second) if (r.strand == '+' && r.motif_idx == (int)mi.second) grouped[r.code].push_back(r);
      if (nit != pfc.end()) for (const Row& r : nit->second) if (r.strand == '-' && r.motif_idx == (int)mi.second) grouped[r.code].push_back(r);
      std::vector<Row>& out = result[ppos];  // entry().or_insert(Vec::new()) even if nothing is appended
      for (auto& g : grouped) {
        Row acc; acc.pos = ppos; acc.strand = '.'; acc.code = g.first; acc.motif_idx = (int)mi.second;
        for (const Row& r : g.second) {  // combine_counts_ignore_strand (93-137)
          acc.n_mod += r.n_mod; acc.n_can += r.n_can; acc.n_other += r.n_other; acc.cov += r.cov;
          acc.n_delete += r.n_delete; acc.n_fail += r.n_fail; acc.n_diff += r.n_diff; acc.n_nocall += r.n_nocall;
          acc.frac = (float)acc.n_mod / (float)acc.cov;
        }
        out.push_back(acc);
      }
    }
  }
  return result;
}

// One read laid out over its reference span: qpos (>=0), -1 deletion, -2 ref-skip.
struct ReadSpan {
  const BamRecord* rec;
  int32_t beg, end;
  std::vector<int32_t> state;
};

// fetch(tid,start,end) + the htslib pileup engine (default mask UNMAP|SECONDARY|QCFAIL|DUP; reads stay in push = file order):
// calls col(P, spans, active) for every column of [start,end) that has at least one read.
template <class ColFn>
static inline void sweep_columns(const BamFile& bam, uint32_t tid, uint32_t start, uint32_t end, uint32_t max_depth, ColFn&& col) {
  std::vector<ReadSpan> spans;
  // bam_plp_push's maxcnt rule (htslib sam.c, the engine behind rust-htslib 0.46's `pileup()`; `set_max_depth` = bam_plp_set_maxcnt,
  // pileup/mod.rs:755-759) — htslib is not in /root/reference: restated from its published source, parity unpinned.  A record is refused
  // when `iter->tid == b->core.tid && iter->pos == b->core.pos && iter->mp->cnt > iter->maxcnt`.  Between two pushes the iterator has
  // walked every column in front of the last buffered record's start P and sits ON P (`max_pos > pos` ends the walk), and a column's
  // walk frees the nodes with `end <= column`; so the test only ever fires for a record that starts where the last BUFFERED record
  // started, and `cnt` is then 1 (the list's empty tail node) + the buffered records that end at or behind P.  The first record of every
  // start is therefore always taken, whatever the depth; a fetch starts a fresh iterator, so the verdict is per interval.
  std::priority_queue<int64_t, std::vector<int64_t>, std::greater<int64_t>> ends;   // buffered records' end positions
  bool have_last = false; int64_t last_beg = 0;
  for (const BamRecord& r : bam.recs) {
    if (r.tid != (int32_t)tid) continue;
    if ((int64_t)r.pos >= (int64_t)end) continue;
    if ((int64_t)r.end_pos() <= (int64_t)start) continue;
    if (r.flag & (4 | 256 | 512 | 1024)) continue;
    {
      const int64_t b = r.pos, e = std::max<int64_t>(r.end_pos(), b + 1);
      while (!ends.empty() && ends.top() < b) ends.pop();
      if (have_last && last_beg == b && 1 + ends.size() > (size_t)max_depth) continue;   // refused: never seen by any column
      ends.push(e); have_last = true; last_beg = b;
    }
    int32_t rl = r.ref_len();
    if (rl <= 0) continue;
    ReadSpan sp; sp.rec = &r; sp.beg = r.pos; sp.end = r.pos + rl; sp.state.assign((size_t)rl, -2);
    int32_t q = 0, rp = 0;
    for (uint32_t c : r.cigar) {
      int op = c & 15; int32_t len = (int32_t)(c >> 4);
      if (op == 0 || op == 7 || op == 8) { for (int32_t k = 0; k < len; k++) sp.state[rp + k] = q + k; q += len; rp += len; }
      else if (op == 1 || op == 4) q += len;
      else if (op == 2) { for (int32_t k = 0; k < len; k++) sp.state[rp + k] = -1; rp += len; }
      else if (op == 3) { rp += len; }
    }
    spans.push_back(std::move(sp));
  }
  if (spans.empty()) return;
  std::vector<size_t> active;
  size_t next = 0;
  int64_t P = std::max<int64_t>(start, spans[0].beg);
  for (; P < (int64_t)end; P++) {
    while (next < spans.size() && spans[next].beg <= P) { active.push_back(next); next++; }
    size_t w = 0;
    for (size_t i = 0; i < active.size(); i++) if (spans[active[i]].end > P) active[w++] = active[i];
    active.resize(w);
    if (active.empty()) {
      if (next >= spans.size()) break;
      P = spans[next].beg - 1;
      if (P + 1 < (int64_t)start) P = (int64_t)start - 1;
      continue;
    }
    col((uint32_t)P, spans, active);
  }
}

// process_region (pileup/mod.rs:718-1020).  `recs` = records of the BAM in file order.
static inline IntervalResult process_region(const BamFile& bam, uint32_t tid, uint32_t start, uint32_t end,
                                            const ThresholdCaller& caller, const PileupOptions& opts,
                                            const FocusPositions& focus) {
  IntervalResult res;
  ReadCache cache; cache.caller = &caller; cache.opts = &opts;
  std::map<uint32_t, std::vector<Row>> pfc;
  std::vector<size_t> tmp_ids;
  sweep_columns(bam, tid, start, end, opts.max_depth, [&](uint32_t P, const std::vector<ReadSpan>& spans, const std::vector<size_t>& active) {
    StrandRule rule;
    if (!focus.check_position((uint32_t)P, &rule)) return;
    Tally pos_tally, neg_tally;
    std::set<ModCode> pos_obs[4], neg_obs[4];
    for (size_t ai : active) {
      const ReadSpan& sp = spans[ai];
      int32_t st = sp.state[(size_t)(P - sp.beg)];
      if (st == -2) continue;  // is_refskip
      const BamRecord& r = *sp.rec;
      if ((r.flag & (2048 | 256 | 1024)) || r.l_seq == 0) continue;  // util.rs:404-407, mod.rs:789
      // add_mod_codes_for_record (read_cache.rs:299-355) — before the deletion check
      const ReadCacheEntry* e = cache.ensure(r);
      if (e) for (int b = 0; b < 4; b++) {
        pos_obs[b].insert(e->codes[0][b].begin(), e->codes[0][b].end());
        neg_obs[b].insert(e->codes[1][b].begin(), e->codes[1][b].end());
      }
      bool aln_neg = r.is_reverse();
      auto add_feature = [&](int kind /*0 del,1 filtered,2 nocall,3 can,4 mod*/, int base, ModCode code, bool read_strand_neg) {
        bool to_pos = (aln_neg == read_strand_neg);  // FeatureVector::add_feature 238-281
        if (rule == RULE_POS && !to_pos) return;
        if (rule == RULE_NEG && to_pos) return;
        Tally& t = to_pos ? pos_tally : neg_tally;
        switch (kind) {
          case 0: t.n_delete++; break;
          case 1: t.n_filtered++; break;
          case 2: t.basecall[base]++; break;
          case 3: t.has_modcall[base] = true; t.canonical[base]++; break;
          case 4: t.has_modcall[base] = true; t.modified[base][code]++; break;
        }
      };
      if (st == -1) { add_feature(0, 0, 0, false); continue; }
      if (st >= r.l_seq) continue;
      int x = base_from_char(r.seq[(size_t)st]);
      if (x < 0) continue;  // 864-874
      int read_base = aln_neg ? complement(x) : x;
      const BaseModCall* pc = nullptr; const BaseModCall* nc = nullptr;
      if (e) {  // get_mod_call (read_cache.rs:232-297)
        if (e->have_pos && e->have_calls[0][read_base]) { auto it = e->calls[0][read_base].find((uint64_t)P);
          if (it != e->calls[0][read_base].end()) pc = &it->second;
          }
        if (e->have_neg && e->have_calls[1][read_base]) { auto it = e->calls[1][read_base].find((uint64_t)P);
          if (it != e->calls[1][read_base].end()) nc = &it->second;
          }
      }
      auto feat = [&](const BaseModCall& c, int pb, bool rs_neg) {  // Feature::from_base_mod_call 38-51
        if (c.kind == BaseModCall::FILTERED) add_feature(1, pb, 0, rs_neg);
        else if (c.kind == BaseModCall::CANONICAL) add_feature(3, pb, 0, rs_neg);
        else add_feature(4, pb, c.code, rs_neg);
      };
      if (pc) feat(*pc, read_base, false);
      if (nc) feat(*nc, complement(read_base), true);
      if (!pc && !nc) add_feature(2, read_base, 0, false);
    }
    std::vector<Row> counts;
    add_tally_to_counts(counts, pos_tally, '+', pos_obs, opts, focus.get_positive_ids((uint32_t)P, &tmp_ids), (uint32_t)P);
    add_tally_to_counts(counts, neg_tally, '-', neg_obs, opts, focus.get_negative_ids((uint32_t)P), (uint32_t)P);
    std::stable_sort(counts.begin(), counts.end(), [](const Row& a, const Row& b) {
      if (a.strand != b.strand) return a.strand < b.strand;  // '+' (43) < '-' (45)
      return a.code < b.code;
    });
    pfc[(uint32_t)P] = std::move(counts);
  });
  if (opts.combine_strands && focus.kind == FocusPositions::MOTIF_COMBINE) res.rows = combine_strand_features(focus.positive_motifs, pfc);
  else res.rows = std::move(pfc);
  res.processed = cache.reads.size();
  res.skipped = cache.skip_set.size();
  return res;
}

// ---------------------------------------------------------------------------------------------------------------------------
// pileup-hemi: duplex pattern counts (src/pileup/duplex.rs, DuplexReadCache src/read_cache.rs:368-468, DuplexModCall
// src/mod_bam.rs:1674-1830).  A pattern element is 0 for a canonical call ('-') or the ModCode: DuplexModCodeRepr's derived order
// (Canonical < Code(char) < ChEbi(u32)) is then the unsigned order of the elements.
struct DuplexRow {  // DuplexPatternCounts + n_delete of the position (duplex.rs:32-85)
  uint32_t pos = 0;
  char base = 'C';            // primary base = the record's SEQ base at the position (reference orientation)
  ModCode pat[2] = {0, 0};
  uint32_t count = 0, n_other = 0, n_diff = 0, n_can = 0, n_fail = 0, n_nocall = 0, n_delete = 0;
};
struct DuplexIntervalResult {
  std::map<uint32_t, std::vector<DuplexRow>> rows;  // rows of a position in writer order: base, then pattern (writers.rs:196-207)
  size_t processed = 0, skipped = 0;
};

// process_region_duplex (duplex.rs:241-339)
static inline DuplexIntervalResult process_region_duplex(const BamFile& bam, uint32_t tid, uint32_t start, uint32_t end,
                                                         const ThresholdCaller& caller, const PileupOptions& opts,
                                                         const FocusPositions& focus) {
  if (focus.kind != FocusPositions::MOTIF_COMBINE) throw MkErr("duplex requires a motif");
  DuplexIntervalResult res;
  ReadCache cache; cache.caller = &caller; cache.opts = &opts;
  auto lookup = [](const ReadCacheEntry* e, int s, int base, uint64_t pos) -> const BaseModCall* {  // get_mod_call, read_cache.rs:232-297
    if (!e) return nullptr;
    if (!(s ? e->have_neg : e->have_pos) || !e->have_calls[s][base]) return nullptr;
    auto it = e->calls[s][base].find(pos);
    return it == e->calls[s][base].end() ? nullptr : &it->second;
  };
  sweep_columns(bam, tid, start, end, opts.max_depth, [&](uint32_t P, const std::vector<ReadSpan>& spans, const std::vector<size_t>& active) {
    StrandRule rule;
    if (!focus.check_position(P, &rule)) return;        // PileupIter
    auto mit = focus.positive_motifs.find(P);            // positions_to_motifs.get(&pos)? (duplex.rs:289-296)
    if (mit == focus.positive_motifs.end()) return;
    const MotifInfo& motif = mit->second[0].first;
    // DuplexFeatureVector (duplex.rs:90-122): kind 0 = ModCall(pattern), 1 = Filtered, 2 = NoCall; key (kind, base, a, b)
    struct Key { int kind; int base; ModCode a, b; bool operator<(const Key& o) const {
        return std::tie(kind, base, a, b) < std::tie(o.kind, o.base, o.a, o.b); } };
    std::map<Key, uint32_t> counts;
    uint32_t n_delete = 0;
    for (size_t ai : active) {
      const ReadSpan& sp = spans[ai];
      int32_t st = sp.state[(size_t)(P - (uint32_t)sp.beg)];
      if (st == -2) continue;  // is_refskip
      const BamRecord& r = *sp.rec;
      if ((r.flag & (2048 | 256 | 1024)) || r.l_seq == 0) continue;
      if (st == -1) { n_delete++; continue; }
      if (st >= r.l_seq) continue;
      int x = base_from_char(r.seq[(size_t)st]);           // get_forward_read_base: NOT complemented here (duplex.rs:308-313)
      if (x < 0) continue;
      // get_duplex_mod_call (read_cache.rs:422-462)
      if (cache.skip_set.count(r.qname)) continue;          // -> None: no feature at all
      bool rev = r.is_reverse();
      int pos_base = rev ? complement(x) : x, neg_base = rev ? x : complement(x);
      const ReadCacheEntry* e = cache.ensure(r);            // a record that fails here still yields one NoCall below
      const BaseModCall* pc = lookup(e, rev ? 1 : 0, pos_base, (uint64_t)P);
      uint32_t npos;
      Key k; k.kind = 2; k.base = x; k.a = 0; k.b = 0;
      if (motif.negative_strand_position(P, &npos)) {
        const BaseModCall* nc = lookup(e, rev ? 0 : 1, neg_base, (uint64_t)npos);
        if (pc && nc) {  // DuplexModCall::from_base_mod_calls (mod_bam.rs:1718-1752)
          if (pc->kind == BaseModCall::FILTERED || nc->kind == BaseModCall::FILTERED) k.kind = 1;
          else {
            k.kind = 0;
            k.a = pc->kind == BaseModCall::CANONICAL ? 0 : pc->code;
            k.b = nc->kind == BaseModCall::CANONICAL ? 0 : nc->code;
            if (opts.numeric == NUM_COMBINE) {  // into_combined (1797-1829)
              if (k.a) k.a = code_char(base_char(x));
              if (k.b) k.b = code_char(base_char(x));
            }
          }
        }
      }
      counts[k]++;
    }
    // DuplexFeatureVector::decode (duplex.rs:124-205)
    std::vector<DuplexRow>& out = res.rows[P];             // position_feature_counts.insert even when nothing is emitted
    for (int pb = 0; pb < 4; pb++) {                       // writer: bases sorted by char = A,C,G,T
      uint32_t total = 0, n_can = 0, n_fail = 0, n_nocall = 0, n_diff = 0;
      for (auto& kv : counts) {
        const Key& k = kv.first;
        if (k.base == pb) {
          if (k.kind == 0) { total += kv.second; if (k.a == 0 && k.b == 0) n_can += kv.second; }
          else if (k.kind == 1) n_fail += kv.second; else n_nocall += kv.second;
        } else if (k.kind == 0) n_diff += kv.second;      // is_mod_call() || is_canonical()
      }
      for (auto& kv : counts) {                            // std::map order = (a, b) ascending = patterns.iter().sorted()
        const Key& k = kv.first;
        if (k.base != pb || k.kind != 0) continue;
        DuplexRow row; row.pos = P; row.base = base_char(pb); row.pat[0] = k.a; row.pat[1] = k.b;
        row.count = kv.second; row.n_other = total - kv.second; row.n_diff = n_diff; row.n_can = n_can; row.n_fail = n_fail;
        row.n_nocall = n_nocall; row.n_delete = n_delete;
        out.push_back(row);
      }
    }
  });
  res.processed = cache.reads.size();
  res.skipped = cache.skip_set.size();
  return res;
}

}  // namespace mko
