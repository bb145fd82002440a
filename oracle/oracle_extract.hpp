// TEST INFRASTRUCTURE ONLY (see oracle_core.hpp).  CPU restatement of `modkit extract calls`:
//   EntryExtractCalls::run                         src/extract/subcommand.rs:452-761
//   process_records_to_chan                        src/extract/util.rs:519-575          (the serial, file-order path)
//   TrackingModRecordIter                          src/mod_bam.rs:27-122
//   ReadBaseModProfile::process_record             src/read_ids_to_base_mod_probs.rs:591-763
//   ReadBaseModProfile::iter_profiles              :785-800
//   ReferencePositionFilter::filter_read_base_mod_probs   src/extract/util.rs:71-124      (--mapped-only only; no BED / motif filters)
//   PositionModCalls::from_profile                 src/read_ids_to_base_mod_probs.rs:1087-1176
//   PositionModCalls::to_row / header              src/extract/writer.rs:12-132
//   Kmer                                           src/util.rs:736-812
//   within_alignment, get_reference_mod_strand     src/util.rs:716-726, 815-831
// Rows come out in FILE order of the records (the reference's indexed path hands batches to its writer in whatever order its
// Rayon pool finishes them; the serial path, which the reference's golden tests pin, is file order).
// Parity unpinned (no reference fixture): two calls of one read on one forward position (duplex mod strands) are ordered by the
// reference's FxHashMap<(usize, Strand, DnaBase)> iteration — here: positive mod strand first.
#pragma once
#include <algorithm>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "oracle_core.hpp"

namespace mko {

// f32 through Rust's Display: the shortest decimal that parses back to the same f32, positional notation
static inline std::string f32_display(float v) {
  if (v == 0.0f) return std::signbit(v) ? "-0" : "0";
  char buf[64]; int prec = 1;
  for (; prec <= 9; prec++) { snprintf(buf, sizeof(buf), "%.*e", prec - 1, (double)v); if (strtof(buf, nullptr) == v) break; }
  // buf = d.ddddde[+-]XX
  std::string s(buf); const size_t e = s.find('e'); int ex = atoi(s.c_str() + e + 1);
  std::string mant = s.substr(0, e); bool neg = false;
  if (!mant.empty() && mant[0] == '-') { neg = true; mant.erase(0, 1); }
  std::string digits; for (char c : mant) if (c != '.') digits.push_back(c);
  while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
  std::string out;
  if (ex >= 0) {
    if ((int)digits.size() <= ex + 1) { out = digits + std::string((size_t)(ex + 1 - (int)digits.size()), '0'); }
    else out = digits.substr(0, (size_t)ex + 1) + "." + digits.substr((size_t)ex + 1);
  } else out = "0." + std::string((size_t)(-ex - 1), '0') + digits;
  return neg ? "-" + out : out;
}

// Kmer::new (util.rs:750-777) as text; '-' where the sequence has no base
static inline std::string kmer_at(const std::string& seq, size_t position, size_t size) {
  const size_t before = size % 2 == 0 ? size / 2 - 1 : size / 2, after = size / 2;
  std::string k;
  for (size_t off = before; off >= 1; off--) k.push_back(position >= off && position - off < seq.size() ? seq[position - off] : '-');
  k.push_back(position < seq.size() ? seq[position] : '-');
  for (size_t off = 1; off <= after; off++) k.push_back(position + off < seq.size() ? seq[position + off] : '-');
  return k;
}
static inline std::string kmer_revcomp(const std::string& k) {
  std::string r;
  for (size_t i = k.size(); i-- > 0;) r.push_back(k[i] == '-' ? '-' : comp_char(k[i]));
  return r;
}

struct ExtractOptions {
  std::string in_bam, out_tsv, ref_fasta, exclude_bed;
  bool allow_non_primary = false, mapped_only = false, pass_only = false, no_headers = false;
  size_t kmer_size = 5;
  // round 6: --num-reads (the serial path's "first N records", util.rs:519-575), --ignore-index, --include-bed (ReferencePositionFilter::keep,
  // util.rs:44-69: the BED is asked with the REFERENCE strand of the mod; rows without a reference position go), --region (util.rs:126-160)
  long num_reads = -1; bool ignore_index = false;
  bool remove_inferred = false;   // --ignore-implicit where the reference honours it: its interval path only (util.rs:413-419)
  bool ask_unmapped = false;   // the scheduled path (ReadsBaseModProfile::process_records) never asks whether a record is mapped
  std::function<bool(int32_t, uint64_t, bool /*reference mod strand is '-'*/)> include;   // empty: no --include-bed
  std::function<bool(int32_t, uint64_t, bool)> exclude;                                    // empty: no --exclude-bed (keep = include hit && !exclude hit)
};

struct ModProfileRow {   // ModProfile (read_ids_to_base_mod_probs.rs:381-397)
  size_t query_position; long ref_position; float q_mod; ModCode code; bool inferred; int strand /*0 +, 1 -*/; int base;
};

static inline const char* extract_calls_header() {
  return "read_id\tforward_read_position\tref_position\tchrom\tmod_strand\tref_strand\tref_mod_strand\tfw_soft_clipped_start\tfw_soft_clipped_end\tread_length\tcall_prob\tcall_code\t"
         "base_qual\tref_kmer\tquery_kmer\tcanonical_base\tmodified_primary_base\tfail\tinferred\twithin_alignment\tflag\n";
}

// one record -> its rows.  Returns false when the record counts as failed (tag / CIGAR error).
// *sent: the record reached process_record (what --num-reads counts: TrackingModRecordIter yielded it and it was not an unmapped record under
// --mapped-only)
static inline bool extract_calls_of_record(const BamFile& bam, const BamRecord& r, const ExtractOptions& o, const CollapseMethod& collapse,
    const EdgeFilter& edge,
                                           const ThresholdCaller& caller, const std::map<std::string, std::string>& ref_seqs, std::string* out,
                                               bool* skipped, bool* sent = nullptr) {
  *skipped = false; if (sent) *sent = false;
  const bool not_primary = (r.flag & (2048 | 256 | 1024)) != 0;                     // record_is_not_primary (util.rs:405-407)
  if (not_primary && !o.allow_non_primary) { *skipped = true; return true; }
  if (r.l_seq == 0) return false;
  ModBaseInfo info;
  try { info = mod_base_info_from_record(r); } catch (const MkErr&) { return false; }
  if (info.is_empty()) { *skipped = true; return true; }
  const bool unmapped = (r.flag & 4) != 0;
  if (unmapped && o.mapped_only && !o.ask_unmapped) { *skipped = true; return true; }
  if (sent) *sent = true;
  const bool rev = r.is_reverse();
  const size_t L = (size_t)r.l_seq;
  // get_soft_clipped (803-824)
  size_t sc_start = 0, sc_end = 0;
  if (!unmapped) {
    bool broke = false; for (uint32_t c : r.cigar) { if ((c & 15u) == 4u) sc_start += c >> 4; else { broke = true; break; } }
    if (!broke) return false;
    broke = false; for (size_t i = r.cigar.size(); i-- > 0;) { const uint32_t c = r.cigar[i]; if ((c & 15u) == 4u) sc_end += c >> 4; else {
        broke = true; break; } }
    if (!broke) return false;
  }
  const size_t clip_start = rev ? sc_end : sc_start, clip_end = rev ? sc_start : sc_end;
  // forward query position -> reference position (aligned_pairs_full: matches carry one, insertions / soft clips none)
  std::vector<long> fwd_to_ref(L, -1);
  if (!unmapped) {
    size_t q = 0; long rp = r.pos;
    for (uint32_t c : r.cigar) {
      const uint32_t op = c & 15u, len = c >> 4;
      if (op == 0 || op == 7 || op == 8) { for (uint32_t k = 0; k < len && q < L; k++, q++, rp++) fwd_to_ref[rev ? L - 1 - q : q] = rp; }
      else if (op == 1 || op == 4) q += len;
      else if (op == 2 || op == 3) rp += len;
    }
  }
  const std::string fwd = forward_sequence(r);
  std::vector<uint8_t> quals(r.qual); if (rev) std::reverse(quals.begin(), quals.end());
  // mod profiles, then sorted by forward position (descending for reverse alignments); stable
  std::vector<ModProfileRow> prof;
  for (int sg = 0; sg < 2; sg++) for (auto& kv : sg ? info.neg_strand : info.pos_strand) {
    const int base = kv.first;
    SeqPosBaseModProbs sp = kv.second;
    if (edge.active) {   // edge_filter_positions (mod_bam.rs:1075-1102)
      if (!edge.read_can_be_trimmed(L)) continue;
      std::map<size_t, BaseModProbs> kept; for (auto& pp : sp.pos) if (edge.keep_position(pp.first, L)) kept.insert(pp);
      if (kept.empty()) continue;
      sp.pos.swap(kept);
    }
    for (auto& pp : sp.pos) {
      const BaseModProbs bmp = collapse.active ? collapse_redistribute(pp.second, collapse.code) : pp.second;
      bmp.probs.for_each([&](ModCode c, float p) { prof.push_back({pp.first, fwd_to_ref[pp.first < L ? pp.first : 0], p, c, bmp.inferred, sg, base});
        });
    }
  }
  if (o.remove_inferred) {   // ReadBaseModProfile::remove_inferred (read_ids_to_base_mod_probs.rs:765-775)
    std::vector<ModProfileRow> k; for (auto& p : prof) if (!p.inferred) k.push_back(p);
    prof.swap(k); }
  std::stable_sort(prof.begin(), prof.end(), [&](const ModProfileRow& a, const ModProfileRow& b) {
    return rev ? a.query_position > b.query_position : a.query_position < b.query_position; });
  // filter_read_base_mod_probs (util.rs:71-124): a profile with a reference position is asked of the BED (reference strand of the mod); one
  // without goes under --mapped-only and under --include-bed (load_regions: "specifying include-only BED outputs only mapped sites")
  if (o.mapped_only || o.include || o.exclude) {
    // include_unmapped_positions (load_regions): --exclude-bed alone keeps them
    const bool unmapped_positions_go = o.mapped_only || (bool)o.include;
    std::vector<ModProfileRow> k;
    for (auto& p : prof) {
      if (unmapped || p.ref_position < 0) { if (!unmapped_positions_go) k.push_back(p); continue; }
      if (o.include && !o.include(r.tid, (uint64_t)p.ref_position, (p.strand != 0) != rev)) continue;
      if (o.exclude && o.exclude(r.tid, (uint64_t)p.ref_position, (p.strand != 0) != rev)) continue;
      k.push_back(p); }
    prof.swap(k);
  }
  if (prof.empty()) { *skipped = true; return true; }
  // iter_profiles: secondary / supplementary records only report what lies inside the alignment
  auto within = [&](size_t qp) { return L >= clip_end && qp >= clip_start && qp < L - clip_end; };
  const bool primary_or_unmapped = r.flag == 0 || r.flag == 16 || r.flag == 4;
  if (!primary_or_unmapped) { std::vector<ModProfileRow> k; for (auto& p : prof) if (within(p.query_position)) k.push_back(p); prof.swap(k); }
  // PositionModCalls::from_profile: group by (position, mod strand, base); the codes of a base over the whole read
  std::map<int, std::set<ModCode>> codes_of_base;
  for (auto& p : prof) codes_of_base[p.base].insert(p.code);
  struct Group { size_t qp; int strand, base; std::vector<const ModProfileRow*> rows; };
  std::vector<Group> groups; std::map<std::tuple<size_t, int, int>, size_t> gidx;
  for (auto& p : prof) {
    auto key = std::make_tuple(p.query_position, p.strand, p.base);
    auto it = gidx.find(key);
    if (it == gidx.end()) { it = gidx.emplace(key, groups.size()).first; groups.push_back({p.query_position, p.strand, p.base, {}}); }
    groups[it->second].rows.push_back(&p);
  }
  const bool neg_aln = !unmapped && rev;
  std::stable_sort(groups.begin(), groups.end(), [&](const Group& a, const Group& b) {
    if (a.qp != b.qp) return neg_aln ? a.qp > b.qp : a.qp < b.qp;
    return a.strand < b.strand;   // (reference: hash-map order — unpinned)
  });
  const std::string chrom = (!unmapped && r.tid >= 0 && (size_t)r.tid < bam.ref_names.size()) ? bam.ref_names[(size_t)r.tid] : std::string();
  const bool have_chrom = !unmapped && r.tid >= 0;
  for (auto& g : groups) {
    BaseModProbs bmp;
    bool any_inferred = false; for (auto* p : g.rows) any_inferred |= p->inferred;
    std::vector<ModCode> all(codes_of_base[g.base].begin(), codes_of_base[g.base].end()); std::sort(all.begin(), all.end());
    if (any_inferred) { bmp.inferred = true; for (ModCode c : all) { float prev; bmp.probs.insert(c, 0.0f, &prev); } }
    else { for (auto* p : g.rows) { float prev; bmp.probs.insert(p->code, p->q_mod, &prev); } for (ModCode c : all) if (bmp.probs.find(c) < 0) {
        float prev; bmp.probs.insert(c, 0.0f, &prev); } }
    const bool filtered = caller.call(g.base, bmp).kind == BaseModCall::FILTERED;
    if (filtered && o.pass_only) continue;
    const BaseModCall am = argmax_call(bmp);
    const long ref_pos = g.rows[0]->ref_position;
    const char mod_strand = g.strand ? '-' : '+';
    const char ref_strand = unmapped ? '.' : (rev ? '-' : '+');
    const char ref_mod_strand = unmapped ? '.' : ((g.strand != 0) != rev ? '-' : '+');
    std::string qk = kmer_at(fwd, g.qp, o.kmer_size); if (g.strand) qk = kmer_revcomp(qk);
    std::string rk = ".";
    if (ref_pos >= 0) { auto it = ref_seqs.find(chrom.empty() ? "." : chrom);
      if (it != ref_seqs.end()) rk = kmer_at(it->second, (size_t)ref_pos, o.kmer_size);
      }
    const uint8_t bq = g.qp < quals.size() ? quals[g.qp] : 0;
    const bool within_aln = have_chrom && within(g.qp);
    char line[1024];
    snprintf(line, sizeof(line), "%s\t%zu\t%ld\t%s\t%c\t%c\t%c\t%zu\t%zu\t%zu\t%s\t%s\t%u\t%s\t%s\t%c\t%c\t%s\t%s\t%s\t%u\n", r.qname.c_str(), g.qp,
        ref_pos >= 0 ? ref_pos : -1L,
             have_chrom ? chrom.c_str() : ".", mod_strand, ref_strand, ref_mod_strand, clip_start, clip_end, L, f32_display(am.p).c_str(),
             am.kind == BaseModCall::CANONICAL ? "-" : code_str(am.code).c_str(), (unsigned)bq, rk.c_str(), qk.c_str(), base_char(g.base),
                 base_char(g.strand ? complement(g.base) : g.base),
             filtered ? "true" : "false", bmp.inferred ? "true" : "false", within_aln ? "true" : "false", (unsigned)r.flag);
    out->append(line);
  }
  return true;
}

}  // namespace mko
