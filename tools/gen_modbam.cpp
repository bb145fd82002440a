// gen_modbam — seeded synthetic modBAM generator for the BASELINE.json configs.
//   gen_modbam --out PREFIX --contig NAME:LEN [--contig ...] --reads N [--style m|hm|hma]
//              [--seed S] [--mean-len 4000] [--sigma 0.6] [--min-len 500] [--max-len 50000] [--cpg-depleted] [--threads T]
// Writes PREFIX.bam (coordinate sorted BGZF), PREFIX.fa, PREFIX.fa.fai and prints counts.
//   reference  : i.i.d. uniform ACGT (seed S) or, with --cpg-depleted, a first-order chain in which a C is followed
//                by G four times less often (human-like CpG depletion)
//   reads      : length = clip(round(LogNormal(ln mean-len, sigma))), start uniform, strand 50/50; per reference base
//                2 % substitution, 1.5 % insertion, 1.5 % deletion (geometric length, mean 1.5); soft clips U[0,50]
//                at both ends; flags: 1 % each secondary / duplicate / supplementary(+MN), 0.5 % QC-fail
//   tags       : every CpG-context C of the as-sequenced read is called; site methylation beta ~ Beta(0.3,0.3)
//                (seeded per reference position), call ~ Bernoulli(beta); ML = 255-|N(0,25)| if methylated else
//                |N(0,25)|, 10 % of calls uniform.  style m: "C+m?" ; style hm: reads alternate "C+hm?" (interleaved
//                ML) and "C+h?;C+m?" (blocked ML) with a 3-way split of probability and 2 % forced h==m ties;
//                style hma: "C+h?;C+m?;A+a?" with 6mA called on every A.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  uint32_t below(uint32_t n) { return (uint32_t)(uni() * n); }
  double normal() { double u = uni(), v = uni(); if (u < 1e-300) u = 1e-300; return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v); }
  double gamma(double a) {  // Marsaglia-Tsang
    if (a < 1.0) { double u = uni(); if (u < 1e-300) u = 1e-300; return gamma(a + 1.0) * std::pow(u, 1.0 / a); }
    double d = a - 1.0 / 3.0, c = 1.0 / std::sqrt(9.0 * d);
    for (;;) { double x = normal(), v = 1.0 + c * x; if (v <= 0) continue; v = v * v * v; double u = uni(); if (u < 1e-300) u = 1e-300; if (std::log(u) < 0.5 * x * x + d - d * v + d * std::log(v)) return d * v; }
  }
};

static const char ACGT[4] = {'A', 'C', 'G', 'T'};
static inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }
static inline uint8_t nib(char c) { return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 4 : c == 'T' ? 8 : 15; }

struct Out { std::vector<uint8_t> d; void put(const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; d.insert(d.end(), b, b + n); } void i32(int32_t v) { put(&v, 4); } void u32(uint32_t v) { put(&v, 4); } void u16(uint16_t v) { put(&v, 2); } void u8(uint8_t v) { d.push_back(v); } };

static double site_beta(uint64_t seed, uint32_t tid, uint32_t pos) { Rng r(seed ^ ((uint64_t)tid << 40) ^ pos); double x = r.gamma(0.3), y = r.gamma(0.3); return x / (x + y); }

static void bgzf_write(const char* path, const std::vector<uint8_t>& data, unsigned threads) {
  const size_t BS = 0xff00; size_t nb = (data.size() + BS - 1) / BS;
  std::vector<std::vector<uint8_t>> blocks(nb);
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (;;) {
      size_t i = next++; if (i >= nb) break;
      size_t off = i * BS, len = std::min(BS, data.size() - off);
      std::vector<uint8_t>& o = blocks[i]; o.resize(len + 1024);
      z_stream zs; memset(&zs, 0, sizeof(zs)); deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
      zs.next_in = const_cast<Bytef*>(&data[off]); zs.avail_in = (uInt)len; zs.next_out = o.data() + 18; zs.avail_out = (uInt)(o.size() - 26);
      deflate(&zs, Z_FINISH); size_t clen = zs.total_out; deflateEnd(&zs);
      const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0};
      memcpy(o.data(), hdr, 16); uint16_t bsize = (uint16_t)(clen + 25); memcpy(o.data() + 16, &bsize, 2);
      uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), &data[off], (uInt)len), isz = (uint32_t)len;
      memcpy(o.data() + 18 + clen, &crc, 4); memcpy(o.data() + 22 + clen, &isz, 4); o.resize(clen + 26);
    }
  };
  std::vector<std::thread> th; for (unsigned t = 0; t < std::max(1u, threads); t++) th.emplace_back(work); for (auto& t : th) t.join();
  FILE* f = fopen(path, "wb"); if (!f) { perror(path); exit(1); }
  for (auto& b : blocks) fwrite(b.data(), 1, b.size(), f);
  static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  fwrite(eof, 1, 28, f); fclose(f);
}

int main(int argc, char** argv) {
  std::string out = "synth", style = "m"; std::vector<std::pair<std::string, uint32_t>> contigs; uint64_t n_reads = 1000, seed = 1; double mean_len = 4000, sigma = 0.6; uint32_t min_len = 500, max_len = 50000;
  bool depleted = false; unsigned threads = std::max(1u, std::thread::hardware_concurrency());
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i]; auto val = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return std::string(argv[++i]); };
    if (a == "--out") out = val(); else if (a == "--contig") { std::string v = val(); size_t c = v.find(':'); contigs.push_back({v.substr(0, c), (uint32_t)strtoul(v.c_str() + c + 1, nullptr, 10)}); }
    else if (a == "--reads") n_reads = strtoull(val().c_str(), nullptr, 10); else if (a == "--seed") seed = strtoull(val().c_str(), nullptr, 10); else if (a == "--style") style = val();
    else if (a == "--mean-len") mean_len = atof(val().c_str()); else if (a == "--sigma") sigma = atof(val().c_str()); else if (a == "--min-len") min_len = (uint32_t)atoi(val().c_str()); else if (a == "--max-len") max_len = (uint32_t)atoi(val().c_str());
    else if (a == "--cpg-depleted") depleted = true; else if (a == "--threads") threads = (unsigned)atoi(val().c_str()); else { fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
  }
  if (contigs.empty()) contigs.push_back({"synth5m", 5000000});
  // ---- reference
  std::vector<std::string> refs; uint64_t total_len = 0;
  { Rng r(seed);
    for (auto& c : contigs) { std::string s(c.second, 'A'); char prev = 'A'; for (uint32_t i = 0; i < c.second; i++) { char b = ACGT[r.below(4)]; if (depleted && prev == 'C' && b == 'G' && r.uni() < 0.75) b = ACGT[r.below(4) == 2 ? 0 : r.below(4)]; s[i] = b; prev = b; } refs.push_back(std::move(s)); total_len += c.second; }
    FILE* fa = fopen((out + ".fa").c_str(), "w"), *fai = fopen((out + ".fa.fai").c_str(), "w"); uint64_t off = 0;
    for (size_t t = 0; t < contigs.size(); t++) { off += (uint64_t)fprintf(fa, ">%s\n", contigs[t].first.c_str()); fprintf(fai, "%s\t%u\t%llu\t60\t61\n", contigs[t].first.c_str(), contigs[t].second, (unsigned long long)off);
      for (uint32_t i = 0; i < contigs[t].second; i += 60) { uint32_t n = std::min(60u, contigs[t].second - i); fwrite(&refs[t][i], 1, n, fa); fputc('\n', fa); off += n + 1; } }
    fclose(fa); fclose(fai); }
  // ---- BAM header
  Out bam; std::string text = "@HD\tVN:1.6\tSO:coordinate\n"; for (auto& c : contigs) text += "@SQ\tSN:" + c.first + "\tLN:" + std::to_string(c.second) + "\n";
  bam.put("BAM\1", 4); bam.i32((int32_t)text.size()); bam.put(text.data(), text.size()); bam.i32((int32_t)contigs.size());
  for (auto& c : contigs) { bam.i32((int32_t)c.first.size() + 1); bam.put(c.first.c_str(), c.first.size() + 1); bam.i32((int32_t)c.second); }
  // ---- reads
  Rng r(seed + 1); const uint64_t beta_seed = seed + 2; uint64_t n_written = 0, aligned = 0, calls = 0, rid = 0;
  for (size_t tid = 0; tid < contigs.size(); tid++) {
    const std::string& ref = refs[tid]; const uint32_t G = contigs[tid].second;
    uint64_t n_here = (uint64_t)((double)n_reads * G / (double)total_len + 0.5);
    struct Plan { uint32_t start, len; }; std::vector<Plan> plan(n_here);
    for (auto& p : plan) { double l = std::exp(std::log(mean_len) + sigma * r.normal()); uint32_t L = (uint32_t)std::min<double>(std::max<double>(std::floor(l + 0.5), min_len), max_len); if (L + 2 > G) L = G > 2 ? G - 2 : 1; p.len = L; p.start = r.below(G - L); }
    std::sort(plan.begin(), plan.end(), [](const Plan& a, const Plan& b) { return a.start < b.start; });
    std::vector<uint32_t> cigar; std::string seq, fwd; std::vector<uint32_t> qref;  // qref: reference position of each query base or ~0
    for (auto& pl : plan) {
      cigar.clear(); seq.clear(); qref.clear();
      auto push = [&](uint32_t n, uint32_t op) { if (!n) return; if (!cigar.empty() && (cigar.back() & 15) == op) cigar.back() += n << 4; else cigar.push_back((n << 4) | op); };
      uint32_t sc = r.below(51); push(sc, 4); for (uint32_t i = 0; i < sc; i++) { seq.push_back(ACGT[r.below(4)]); qref.push_back(~0u); }
      uint32_t p = pl.start, end = pl.start + pl.len; bool first = true;
      while (p < end) {
        double x = first ? 1.0 : r.uni(); first = false;
        if (x < 0.015) { uint32_t n = 1; while (r.uni() < 1.0 / 3.0) n++; push(n, 1); for (uint32_t i = 0; i < n; i++) { seq.push_back(ACGT[r.below(4)]); qref.push_back(~0u); } }
        else if (x < 0.03) { uint32_t n = 1; while (r.uni() < 1.0 / 3.0) n++; if (p + n >= end) n = 1; if (p + n < end) { push(n, 2); p += n; } else { push(1, 0); seq.push_back(ref[p]); qref.push_back(p); p++; } }
        else if (x < 0.05) { char b = ACGT[r.below(4)]; if (b == ref[p]) b = comp(b); push(1, 0); seq.push_back(b); qref.push_back(p); p++; }
        else { push(1, 0); seq.push_back(ref[p]); qref.push_back(p); p++; }
      }
      while (!cigar.empty() && (cigar.back() & 15) == 1) { uint32_t n = cigar.back() >> 4; cigar.pop_back(); seq.resize(seq.size() - n); qref.resize(qref.size() - n); }
      uint32_t ec = r.below(51); push(ec, 4); for (uint32_t i = 0; i < ec; i++) { seq.push_back(ACGT[r.below(4)]); qref.push_back(~0u); }
      const bool rev = r.uni() < 0.5; const uint32_t L = (uint32_t)seq.size();
      uint16_t flag = rev ? 16 : 0; double fx = r.uni(); bool need_mn = false;
      if (fx < 0.01) { flag |= 256; need_mn = true; } else if (fx < 0.02) flag |= 1024; else if (fx < 0.03) { flag |= 2048; need_mn = true; } else if (fx < 0.035) flag |= 512;
      // as-sequenced read and its calls
      fwd.resize(L); for (uint32_t i = 0; i < L; i++) fwd[i] = rev ? comp(seq[L - 1 - i]) : seq[i];
      auto site = [&](uint32_t f) -> double { uint32_t q = rev ? L - 1 - f : f; uint32_t rp = qref[q]; return rp == ~0u ? 0.5 : site_beta(beta_seed, (uint32_t)tid, rev ? rp - 1 : rp); };
      auto qual = [&](bool meth) -> uint8_t { if (r.uni() < 0.1) return (uint8_t)r.below(256); int v = (int)std::floor(std::fabs(r.normal() * 25.0)); if (v > 255) v = 255; return (uint8_t)(meth ? 255 - v : v); };
      std::string mm; std::vector<uint8_t> ml; char num[16];
      std::vector<uint32_t> cpos; std::string deltas; { uint32_t skipped = 0; for (uint32_t f = 0; f < L; f++) if (fwd[f] == 'C') { if (f + 1 < L && fwd[f + 1] == 'G') { cpos.push_back(f); snprintf(num, sizeof(num), ",%u", skipped); deltas += num; skipped = 0; } else skipped++; } }
      const bool hm = style == "hm" || style == "hma";
      if (!hm) { mm = "C+m?" + deltas + ";"; for (uint32_t f : cpos) ml.push_back(qual(r.uni() < site(f))); }
      else {
        std::vector<uint8_t> hv, mv;
        for (uint32_t f : cpos) {
          double b = site(f); bool meth = r.uni() < b; uint8_t qm = qual(meth); uint32_t rest = 255u - qm; uint8_t qh = (uint8_t)(r.uni() < 0.15 ? r.below(rest + 1) : r.below(rest / 4 + 1));
          if (r.uni() < 0.02) { qm = qh = (uint8_t)r.below(128); }  // forced tie
          hv.push_back(qh); mv.push_back(qm);
        }
        const bool combined = style == "hm" && (rid & 1);
        if (combined) { mm = "C+hm?" + deltas + ";"; for (size_t i = 0; i < hv.size(); i++) { ml.push_back(hv[i]); ml.push_back(mv[i]); } }
        else { mm = "C+h?" + deltas + ";C+m?" + deltas + ";"; ml.insert(ml.end(), hv.begin(), hv.end()); ml.insert(ml.end(), mv.begin(), mv.end()); }
        if (style == "hma") { mm += "A+a?"; for (uint32_t f = 0; f < L; f++) if (fwd[f] == 'A') { mm += ",0"; ml.push_back(qual(r.uni() < 0.05)); } mm += ";"; }
      }
      calls += cpos.size();
      // record
      char qn[32]; int lq = snprintf(qn, sizeof(qn), "r%09llu", (unsigned long long)rid++) + 1;
      size_t at = bam.d.size(); bam.i32(0);
      bam.i32((int32_t)tid); bam.i32((int32_t)pl.start); bam.u8((uint8_t)lq); bam.u8(60); bam.u16(4680); bam.u16((uint16_t)cigar.size()); bam.u16(flag); bam.i32((int32_t)L); bam.i32(-1); bam.i32(-1); bam.i32(0);
      bam.put(qn, (size_t)lq); bam.put(cigar.data(), cigar.size() * 4);
      for (uint32_t i = 0; i < L; i += 2) bam.u8((uint8_t)((nib(seq[i]) << 4) | (i + 1 < L ? nib(seq[i + 1]) : 0)));
      bam.d.insert(bam.d.end(), L, 0xff);
      bam.put("MMZ", 3); bam.put(mm.c_str(), mm.size() + 1);
      bam.put("MLBC", 4); bam.u32((uint32_t)ml.size()); bam.put(ml.data(), ml.size());
      if (need_mn) { bam.put("MNi", 3); bam.i32((int32_t)L); }
      int32_t bs = (int32_t)(bam.d.size() - at - 4); memcpy(&bam.d[at], &bs, 4);
      if (cigar.size() > 65535) { fprintf(stderr, "cigar too long\n"); return 1; }
      n_written++; aligned += pl.len;
    }
  }
  bgzf_write((out + ".bam").c_str(), bam.d, threads);
  printf("{\"reads\": %llu, \"aligned_bases\": %llu, \"cpg_calls\": %llu, \"bam_bytes_uncompressed\": %llu, \"genome\": %llu}\n", (unsigned long long)n_written, (unsigned long long)aligned, (unsigned long long)calls,
         (unsigned long long)bam.d.size(), (unsigned long long)total_len);
  return 0;
}
