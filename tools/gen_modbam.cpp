// gen_modbam — seeded synthetic modBAM generator for the BASELINE.json configs.
//   gen_modbam --out PREFIX --contig NAME:LEN [--contig ...] --reads N [--style m|hm|hma|duplex]
//              [--seed S] [--mean-len 4000] [--sigma 0.6] [--min-len 500] [--max-len 50000] [--cpg-depleted] [--threads T]
//              [--partition-tag HP:3]
// Writes PREFIX.bam (coordinate sorted BGZF), PREFIX.bam.bai (bins, linear index, idxstats pseudo-bin), PREFIX.fa,
// PREFIX.fa.fai and prints counts.  Deterministic for (flags, seed) whatever --threads is: every read draws from its own
// generator, seeded by (seed, read index).
//   reference  : i.i.d. uniform ACGT (seed S) or, with --cpg-depleted, a first-order chain in which G follows C with
//                probability 0.04 (CpG ~ 1 % of dinucleotides, human-like)
//   reads      : length = clip(round(LogNormal(ln mean-len, sigma))) (--mean-len is the median), start uniform, strand 50/50;
//                per reference base 2 % substitution, 1.5 % insertion, 1.5 % deletion (geometric length, mean 1.5); soft
//                clips U[0,50] at both ends; flags: 1 % each secondary / duplicate / supplementary(+MN), 0.5 % QC-fail
//   tags       : every CpG-context C of the as-sequenced read is called; site methylation beta ~ Beta(0.3,0.3)
//                (seeded per reference position), call ~ Bernoulli(beta); ML = 255-|N(0,25)| if methylated else
//                |N(0,25)|, 10 % of calls uniform.  style m: "C+m?" ; style hm: reads alternate "C+hm?" (interleaved
//                ML) and "C+h?;C+m?" (blocked ML) with a 3-way split of probability and 2 % forced h==m ties;
//                style hma: "C+h?;C+m?;A+a?" with 6mA called on every A.
//                --partition-tag NAME:K adds an integer aux tag NAME:i with values 1..K (5 % of reads carry none).
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  uint32_t below(uint32_t n) { return (uint32_t)(uni() * n); }
  double normal() { double u = uni(), v = uni(); if (u < 1e-300) u = 1e-300; return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v); }
  double gamma(double a) {  // Marsaglia-Tsang
    if (a < 1.0) { double u = uni(); if (u < 1e-300) u = 1e-300; return gamma(a + 1.0) * std::pow(u, 1.0 / a); }
    double d = a - 1.0 / 3.0, c = 1.0 / std::sqrt(9.0 * d);
    for (;;) { double x = normal(), v = 1.0 + c * x; if (v <= 0) continue; v = v * v * v; double u = uni(); if (u < 1e-300) u = 1e-300;
      if (std::log(u) < 0.5 * x * x + d - d * v + d * std::log(v)) return d * v;
      }
  }
};

static const char ACGT[4] = {'A', 'C', 'G', 'T'};
static inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }
static inline uint8_t nib(char c) { return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 4 : c == 'T' ? 8 : 15; }

struct Out { std::vector<uint8_t> d; void put(const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; d.insert(d.end(), b, b + n);
  } void i32(int32_t v) { put(&v, 4); } void u32(uint32_t v) { put(&v, 4); } void u16(uint16_t v) { put(&v, 2); } void u8(uint8_t v) {
    d.push_back(v); } };

static double site_beta(uint64_t seed, uint32_t tid, uint32_t pos) { Rng r(seed ^ ((uint64_t)tid << 40) ^ pos);
  double x = r.gamma(0.3), y = r.gamma(0.3); return x / (x + y); }

template <class F> static void parallel_for(size_t n, unsigned threads, F f) {
  std::atomic<size_t> next{0};
  auto work = [&]() { for (;;) { size_t i = next++; if (i >= n) break; f(i); } };
  std::vector<std::thread> th; for (unsigned t = 0; t < std::max(1u, threads); t++) th.emplace_back(work); for (auto& t : th) t.join();
}

// The uncompressed BAM stream is a sequence of pieces (header, then one piece per block of reads); BGZF blocks cut it every
// 0xff00 bytes whatever the piece boundaries are.
struct Stream {
  std::vector<const std::vector<uint8_t>*> pieces; std::vector<uint64_t> start;  // start[i] = stream offset of piece i; back() = total
  void finish() { start.assign(pieces.size() + 1, 0); for (size_t i = 0; i < pieces.size(); i++) start[i + 1] = start[i] + pieces[i]->size(); }
  uint64_t size() const { return start.back(); }
  void gather(uint64_t off, size_t len, uint8_t* dst) const {
    size_t i = (size_t)(std::upper_bound(start.begin(), start.end(), off) - start.begin()) - 1;
    while (len) { const size_t o = (size_t)(off - start[i]), n = std::min(len, pieces[i]->size() - o); memcpy(dst, pieces[i]->data() + o, n);
      dst += n; off += n; len -= n; i++; }
  }
};
static const size_t BGZF_BS = 0xff00;

// returns the file offset of every BGZF block (one more entry: the EOF block)
static std::vector<uint64_t> bgzf_write(const char* path, const Stream& st, unsigned threads) {
  const size_t nb = (size_t)((st.size() + BGZF_BS - 1) / BGZF_BS);
  std::vector<std::vector<uint8_t>> blocks(nb);
  parallel_for(nb, threads, [&](size_t i) {
    const uint64_t off = (uint64_t)i * BGZF_BS; const size_t len = (size_t)std::min<uint64_t>(BGZF_BS, st.size() - off);
    std::vector<uint8_t> in(len); st.gather(off, len, in.data());
    std::vector<uint8_t>& o = blocks[i]; o.resize(len + 1024);
    z_stream zs; memset(&zs, 0, sizeof(zs)); deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = in.data(); zs.avail_in = (uInt)len; zs.next_out = o.data() + 18; zs.avail_out = (uInt)(o.size() - 26);
    deflate(&zs, Z_FINISH); size_t clen = zs.total_out; deflateEnd(&zs);
    const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0};
    memcpy(o.data(), hdr, 16); uint16_t bsize = (uint16_t)(clen + 25); memcpy(o.data() + 16, &bsize, 2);
    uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in.data(), (uInt)len), isz = (uint32_t)len;
    memcpy(o.data() + 18 + clen, &crc, 4); memcpy(o.data() + 22 + clen, &isz, 4); o.resize(clen + 26);
  });
  FILE* f = fopen(path, "wb"); if (!f) { perror(path); exit(1); }
  std::vector<uint64_t> coff(nb + 1, 0);
  for (size_t i = 0; i < nb; i++) { coff[i + 1] = coff[i] + blocks[i].size(); fwrite(blocks[i].data(), 1, blocks[i].size(), f); }
  static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  fwrite(eof, 1, 28, f); fclose(f);
  return coff;
}

static inline uint32_t reg2bin(int64_t beg, int64_t end) {  // SAM spec 5.3
  --end;
  if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}

struct RecIdx { int32_t tid; uint32_t pos, end; uint16_t flag; uint64_t off; uint32_t size; };  // off/size: block_size field included

int main(int argc, char** argv) {
  std::string out = "synth", style = "m", ptag; uint32_t ptag_k = 0; std::vector<std::pair<std::string, uint32_t>> contigs;
    uint64_t n_reads = 1000, seed = 1; double mean_len = 4000, sigma = 0.6; uint32_t min_len = 500, max_len = 50000;
  bool depleted = false; unsigned threads = std::max(1u, std::thread::hardware_concurrency());
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i]; auto val = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2);
      } return std::string(argv[++i]); };
    if (a == "--out") out = val(); else if (a == "--contig") { std::string v = val(); size_t c = v.find(':');
      contigs.push_back({v.substr(0, c), (uint32_t)strtoul(v.c_str() + c + 1, nullptr, 10)}); }
    else if (a == "--reads") n_reads = strtoull(val().c_str(), nullptr, 10); else if (a == "--seed") seed = strtoull(val().c_str(), nullptr, 10);
      else if (a == "--style") style = val();
    else if (a == "--mean-len") mean_len = atof(val().c_str()); else if (a == "--sigma") sigma = atof(val().c_str());
      else if (a == "--min-len") min_len = (uint32_t)atoi(val().c_str());
      else if (a == "--max-len") max_len = (uint32_t)atoi(val().c_str());
    else if (a == "--cpg-depleted") depleted = true; else if (a == "--threads") threads = (unsigned)atoi(val().c_str());
    else if (a == "--partition-tag") { std::string v = val(); size_t c = v.find(':'); if (c != 2) { fprintf(stderr, "--partition-tag wants XX:K\n");
        return 2; } ptag = v.substr(0, 2); ptag_k = (uint32_t)atoi(v.c_str() + 3); }
    else { fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
  }
  if (contigs.empty()) contigs.push_back({"synth5m", 5000000});
  // ---- reference (one generator per contig so contigs are built in parallel)
  std::vector<std::string> refs(contigs.size()); uint64_t total_len = 0;
  for (auto& c : contigs) total_len += c.second;
  parallel_for(contigs.size(), threads, [&](size_t t) {
    Rng r(seed * 1000003ull + t); std::string s(contigs[t].second, 'A'); char prev = 'A';
    for (uint32_t i = 0; i < contigs[t].second; i++) {
      char b = ACGT[r.below(4)];
      if (depleted && prev == 'C') { b = r.uni() < 0.04 ? 'G' : "ACT"[r.below(3)]; }
      s[i] = b; prev = b;
    }
    refs[t] = std::move(s);
  });
  { FILE* fa = fopen((out + ".fa").c_str(), "w"), *fai = fopen((out + ".fa.fai").c_str(), "w"); uint64_t off = 0; std::string line;
    for (size_t t = 0; t < contigs.size(); t++) { off += (uint64_t)fprintf(fa, ">%s\n", contigs[t].first.c_str());
      fprintf(fai, "%s\t%u\t%llu\t60\t61\n", contigs[t].first.c_str(), contigs[t].second, (unsigned long long)off);
      std::string buf; buf.reserve((size_t)contigs[t].second + contigs[t].second / 60 + 2);
      for (uint32_t i = 0; i < contigs[t].second; i += 60) { uint32_t n = std::min(60u, contigs[t].second - i); buf.append(&refs[t][i], n);
        buf.push_back('\n'); off += n + 1; }
      fwrite(buf.data(), 1, buf.size(), fa); }
    fclose(fa); fclose(fai); }
  // ---- BAM header
  Out hdr; std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (auto& c : contigs) text += "@SQ\tSN:" + c.first + "\tLN:" + std::to_string(c.second) + "\n";
  hdr.put("BAM\1", 4); hdr.i32((int32_t)text.size()); hdr.put(text.data(), text.size()); hdr.i32((int32_t)contigs.size());
  for (auto& c : contigs) { hdr.i32((int32_t)c.first.size() + 1); hdr.put(c.first.c_str(), c.first.size() + 1); hdr.i32((int32_t)c.second); }
  // ---- read plans (sequential generator: cheap), coordinate sorted per contig
  struct Plan { uint32_t tid, start, len; };
  std::vector<Plan> plan;
  { Rng r(seed + 1);
    for (size_t tid = 0; tid < contigs.size(); tid++) {
      const uint32_t G = contigs[tid].second; const uint64_t n_here = (uint64_t)((double)n_reads * G / (double)total_len + 0.5);
        const size_t at = plan.size();
      for (uint64_t k = 0; k < n_here; k++) { double l = std::exp(std::log(mean_len) + sigma * r.normal());
        uint32_t L = (uint32_t)std::min<double>(std::max<double>(std::floor(l + 0.5), min_len), max_len); if (L + 2 > G) L = G > 2 ? G - 2 : 1;
        plan.push_back({(uint32_t)tid, r.below(G - L), L}); }
      std::stable_sort(plan.begin() + (std::ptrdiff_t)at, plan.end(), [](const Plan& a, const Plan& b) { return a.start < b.start; });
    } }
  // ---- reads, in blocks of 128 built in parallel; every read has its own generator
  const size_t RB = 128, nblocks = (plan.size() + RB - 1) / RB; const uint64_t beta_seed = seed + 2;
  std::vector<Out> blk(nblocks); std::vector<std::vector<RecIdx>> blk_idx(nblocks);
    std::vector<uint64_t> blk_aligned(nblocks, 0), blk_calls(nblocks, 0);
  std::atomic<bool> bad{false};
  parallel_for(nblocks, threads, [&](size_t bi) {
    // qref: reference position of each query base or ~0
    Out& bam = blk[bi]; std::vector<uint32_t> cigar; std::string seq, fwd; std::vector<uint32_t> qref;
    for (size_t rid = bi * RB; rid < std::min(plan.size(), (bi + 1) * RB); rid++) {
      const Plan& pl = plan[rid]; const std::string& ref = refs[pl.tid]; const uint32_t tid = pl.tid;
      Rng r((seed + 7) * 0x2545F4914F6CDD1Dull + rid);
      cigar.clear(); seq.clear(); qref.clear();
      auto push = [&](uint32_t n, uint32_t op) { if (!n) return; if (!cigar.empty() && (cigar.back() & 15) == op) cigar.back() += n << 4;
        else cigar.push_back((n << 4) | op);
        };
      uint32_t sc = r.below(51); push(sc, 4); for (uint32_t i = 0; i < sc; i++) { seq.push_back(ACGT[r.below(4)]); qref.push_back(~0u); }
      uint32_t p = pl.start, end = pl.start + pl.len; bool first = true;
      while (p < end) {
        double x = first ? 1.0 : r.uni(); first = false;
        if (x < 0.015) { uint32_t n = 1; while (r.uni() < 1.0 / 3.0) n++; push(n, 1); for (uint32_t i = 0; i < n; i++) {
            seq.push_back(ACGT[r.below(4)]); qref.push_back(~0u); } }
        else if (x < 0.03) { uint32_t n = 1; while (r.uni() < 1.0 / 3.0) n++; if (p + n >= end) n = 1; if (p + n < end) { push(n, 2); p += n; } else {
            push(1, 0); seq.push_back(ref[p]); qref.push_back(p); p++; } }
        else if (x < 0.05) { char b = ACGT[r.below(4)]; if (b == ref[p]) b = comp(b); push(1, 0); seq.push_back(b); qref.push_back(p); p++; }
        else { push(1, 0); seq.push_back(ref[p]); qref.push_back(p); p++; }
      }
      while (!cigar.empty() && (cigar.back() & 15) == 1) { uint32_t n = cigar.back() >> 4; cigar.pop_back(); seq.resize(seq.size() - n);
        qref.resize(qref.size() - n); }
      uint32_t ec = r.below(51); push(ec, 4); for (uint32_t i = 0; i < ec; i++) { seq.push_back(ACGT[r.below(4)]); qref.push_back(~0u); }
      const bool rev = r.uni() < 0.5; const uint32_t L = (uint32_t)seq.size();
      uint16_t flag = rev ? 16 : 0; double fx = r.uni(); bool need_mn = false;
      if (fx < 0.01) { flag |= 256; need_mn = true; } else if (fx < 0.02) flag |= 1024; else if (fx < 0.03) { flag |= 2048; need_mn = true;
        } else if (fx < 0.035) flag |= 512;
      // as-sequenced read and its calls
      fwd.resize(L); for (uint32_t i = 0; i < L; i++) fwd[i] = rev ? comp(seq[L - 1 - i]) : seq[i];
      auto site = [&](uint32_t f) -> double { uint32_t q = rev ? L - 1 - f : f; uint32_t rp = qref[q];
        return rp == ~0u ? 0.5 : site_beta(beta_seed, tid, rev ? rp - 1 : rp); };
      auto qual = [&](bool meth) -> uint8_t { if (r.uni() < 0.1) return (uint8_t)r.below(256); int v = (int)std::floor(std::fabs(r.normal() * 25.0)); if (v > 255) v = 255; return (uint8_t)(meth
          ? 255 - v : v); };
      std::string mm; std::vector<uint8_t> ml; char num[16];
      std::vector<uint32_t> cpos; std::string deltas; { uint32_t skipped = 0; for (uint32_t f = 0; f < L; f++) if (fwd[f] == 'C') {
          if (f + 1 < L && fwd[f + 1] == 'G') { cpos.push_back(f);
            snprintf(num, sizeof(num), ",%u", skipped); deltas += num; skipped = 0; } else skipped++; } }
      if (style == "duplex") {
        // duplex basecalls: the read's own strand as C+h / C+m at its CpG C's, the opposite strand as G-h / G-m at the G's of the same
        // CpGs; the two strands of a site are drawn independently from the site's methylation level (mostly concordant, some hemi)
        std::vector<uint32_t> gpos; std::string gdeltas; { uint32_t skipped = 0; for (uint32_t f = 0; f < L; f++) if (fwd[f] == 'G') {
            if (f > 0 && fwd[f - 1] == 'C') { gpos.push_back(f);
              snprintf(num, sizeof(num), ",%u", skipped); gdeltas += num; skipped = 0; } else skipped++; } }
        auto gsite = [&](uint32_t f) -> double { uint32_t q = rev ? L - 1 - f : f; uint32_t rp = qref[q];
          return rp == ~0u ? 0.5 : site_beta(beta_seed, tid, rev ? rp : rp - 1); };
        std::vector<uint8_t> hv[2], mv[2];
        for (int sd = 0; sd < 2; sd++) for (uint32_t f : (sd ? gpos : cpos)) {
          double b = sd ? gsite(f) : site(f); bool meth = r.uni() < b; uint8_t qm = qual(meth); uint32_t rest = 255u - qm;
            uint8_t qh = (uint8_t)(r.uni() < 0.15 ? r.below(rest + 1) : r.below(rest / 4 + 1));
          hv[sd].push_back(qh); mv[sd].push_back(qm);
        }
        if (rid & 1) { mm = "C+hm?" + deltas + ";G-hm?" + gdeltas + ";"; for (int sd = 0; sd < 2; sd++) for (size_t i = 0; i < hv[sd].size(); i++) {
            ml.push_back(hv[sd][i]); ml.push_back(mv[sd][i]); } }
        else { mm = "C+h?" + deltas + ";C+m?" + deltas + ";G-h?" + gdeltas + ";G-m?" + gdeltas + ";"; for (int sd = 0; sd < 2; sd++) {
            ml.insert(ml.end(), hv[sd].begin(), hv[sd].end()); ml.insert(ml.end(), mv[sd].begin(), mv[sd].end()); } }
        blk_calls[bi] += gpos.size();
      } else {
      const bool hm = style == "hm" || style == "hma";
      if (!hm) { mm = "C+m?" + deltas + ";"; for (uint32_t f : cpos) ml.push_back(qual(r.uni() < site(f))); }
      else {
        std::vector<uint8_t> hv, mv;
        for (uint32_t f : cpos) {
          double b = site(f); bool meth = r.uni() < b; uint8_t qm = qual(meth); uint32_t rest = 255u - qm;
            uint8_t qh = (uint8_t)(r.uni() < 0.15 ? r.below(rest + 1) : r.below(rest / 4 + 1));
          if (r.uni() < 0.02) { qm = qh = (uint8_t)r.below(128); }  // forced tie
          hv.push_back(qh); mv.push_back(qm);
        }
        const bool combined = style == "hm" && (rid & 1);
        if (combined) { mm = "C+hm?" + deltas + ";"; for (size_t i = 0; i < hv.size(); i++) { ml.push_back(hv[i]); ml.push_back(mv[i]); } }
        else { mm = "C+h?" + deltas + ";C+m?" + deltas + ";"; ml.insert(ml.end(), hv.begin(), hv.end()); ml.insert(ml.end(), mv.begin(), mv.end()); }
        if (style == "hma") { mm += "A+a?"; for (uint32_t f = 0; f < L; f++) if (fwd[f] == 'A') { mm += ",0"; ml.push_back(qual(r.uni() < 0.05));
          } mm += ";"; }
      }
      }
      blk_calls[bi] += cpos.size();
      // record
      char qn[32]; int lq = snprintf(qn, sizeof(qn), "r%09llu", (unsigned long long)rid) + 1;
      size_t at = bam.d.size(); bam.i32(0);
      const uint32_t rend = pl.start + pl.len;
      bam.i32((int32_t)tid); bam.i32((int32_t)pl.start); bam.u8((uint8_t)lq); bam.u8(60); bam.u16((uint16_t)reg2bin(pl.start, rend));
        bam.u16((uint16_t)cigar.size()); bam.u16(flag); bam.i32((int32_t)L); bam.i32(-1); bam.i32(-1); bam.i32(0);
      bam.put(qn, (size_t)lq); bam.put(cigar.data(), cigar.size() * 4);
      for (uint32_t i = 0; i < L; i += 2) bam.u8((uint8_t)((nib(seq[i]) << 4) | (i + 1 < L ? nib(seq[i + 1]) : 0)));
      bam.d.insert(bam.d.end(), L, 0xff);
      bam.put("MMZ", 3); bam.put(mm.c_str(), mm.size() + 1);
      bam.put("MLBC", 4); bam.u32((uint32_t)ml.size()); bam.put(ml.data(), ml.size());
      if (need_mn) { bam.put("MNi", 3); bam.i32((int32_t)L); }
      if (ptag_k && r.uni() >= 0.05) { bam.put(ptag.data(), 2); bam.u8('i'); bam.i32((int32_t)(1 + r.below(ptag_k))); }
      int32_t bs = (int32_t)(bam.d.size() - at - 4); memcpy(&bam.d[at], &bs, 4);
      if (cigar.size() > 65535) { fprintf(stderr, "cigar too long\n"); bad = true; }
      blk_idx[bi].push_back({(int32_t)tid, pl.start, rend, flag, (uint64_t)at, (uint32_t)(bam.d.size() - at)});
      blk_aligned[bi] += pl.len;
    }
  });
  if (bad) return 1;
  Stream st; st.pieces.push_back(&hdr.d); for (auto& b : blk) st.pieces.push_back(&b.d); st.finish();
  const std::vector<uint64_t> coff = bgzf_write((out + ".bam").c_str(), st, threads);
  auto voff = [&](uint64_t u) { return (coff[(size_t)(u / BGZF_BS)] << 16) | (u % BGZF_BS); };
  // ---- BAI (SAM spec 5.2): bins with chunk lists, 16 kb linear index, pseudo-bin 37450 with mapped/unmapped counts
  { FILE* f = fopen((out + ".bam.bai").c_str(), "wb"); if (!f) { perror("bai"); return 1; }
    auto w32 = [&](uint32_t v) { fwrite(&v, 4, 1, f); }; auto w64 = [&](uint64_t v) { fwrite(&v, 8, 1, f); };
    fwrite("BAI\1", 1, 4, f); w32((uint32_t)contigs.size());
    size_t bi = 0, ri = 0;   // cursor over (block, record) in file order
    for (size_t tid = 0; tid < contigs.size(); tid++) {
      std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins; std::vector<uint64_t> lin;
        uint64_t n_mapped = 0, n_unmapped = 0, off_beg = ~0ull, off_end = 0;
      for (; bi < nblocks; bi++, ri = 0) {
        bool stop = false;
        for (; ri < blk_idx[bi].size(); ri++) {
          const RecIdx& x = blk_idx[bi][ri]; if (x.tid != (int32_t)tid) { stop = true; break; }
          const uint64_t u0 = st.start[bi + 1] + x.off, v0 = voff(u0), v1 = voff(u0 + x.size);
          auto& ch = bins[reg2bin(x.pos, x.end)]; if (!ch.empty() && ch.back().second == v0) ch.back().second = v1; else ch.push_back({v0, v1});
          for (uint32_t w = x.pos >> 14; w <= (x.end - 1) >> 14; w++) { if (lin.size() <= w) lin.resize(w + 1, 0); if (!lin[w]) lin[w] = v0; }
          if (x.flag & 4) n_unmapped++; else n_mapped++;
          off_beg = std::min(off_beg, v0); off_end = std::max(off_end, v1);
        }
        if (stop) break;
      }
      for (size_t w = 1; w < lin.size(); w++) if (!lin[w]) lin[w] = lin[w - 1];   // htslib fills empty windows with the previous offset
      w32((uint32_t)bins.size() + (n_mapped + n_unmapped ? 1u : 0u));
      for (auto& kv : bins) { w32(kv.first); w32((uint32_t)kv.second.size()); for (auto& c : kv.second) { w64(c.first); w64(c.second); } }
      if (n_mapped + n_unmapped) { w32(37450); w32(2); w64(off_beg); w64(off_end); w64(n_mapped); w64(n_unmapped); }
      w32((uint32_t)lin.size()); for (uint64_t v : lin) w64(v);
    }
    w64(0);  // n_no_coor
    fclose(f); }
  uint64_t aligned = 0, calls = 0; for (size_t i = 0; i < nblocks; i++) { aligned += blk_aligned[i]; calls += blk_calls[i]; }
  printf("{\"reads\": %llu, \"aligned_bases\": %llu, \"cpg_calls\": %llu, \"bam_bytes_uncompressed\": %llu, \"genome\": %llu}\n",
      (unsigned long long)plan.size(), (unsigned long long)aligned, (unsigned long long)calls,
         (unsigned long long)st.size(), (unsigned long long)total_len);
  return 0;
}
