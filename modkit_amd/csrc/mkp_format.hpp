// bedMethyl row text for mkp_pileup_main, byte-identical to BedMethylWriter::write_feature_counts
// (src/writers.rs:87-156): 18 columns, `{:.2}` of `fraction_modified * 100f32`.  The reference keeps this in Rust; the
// library formats rows only so that its whole-subcommand entry point can be checked against golden files.  Plain C++,
// no device code: tests/test_format_cpu.py compiles it against snprintf for every (n_mod, n_valid) pair up to 4096.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

namespace mkp {

// decimal digits of v, two at a time from a pair table; counts are mostly below 100 and take one of the two short paths
inline char* put_u32(char* p, uint32_t v) {
  static const char kPairs[201] =
      "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
  if (v < 10u) { *p++ = (char)('0' + v); return p; }
  if (v < 100u) { memcpy(p, kPairs + 2u * v, 2); return p + 2; }
  char tmp[10]; int n = 10;
  while (v >= 100u) { const uint32_t q = v / 100u, r = v - q * 100u; n -= 2; memcpy(tmp + n, kPairs + 2u * r, 2); v = q; }
  if (v >= 10u) { n -= 2; memcpy(tmp + n, kPairs + 2u * v, 2); } else tmp[--n] = (char)('0' + v);
  memcpy(p, tmp + n, (size_t)(10 - n));
  return p + (10 - n);
}

// "{:.2}" of an f32: the exact decimal expansion of the value rounded to 2 places, ties to even (Rust's float
// formatting and glibc's printf agree on this).  pct has a 24-bit significand, so pct * 100 is exact in double and
// nearbyint (round-to-nearest-even) of it is the correctly rounded number of hundredths.
inline char* put_pct2(char* p, float pct) {
  const double x = (double)pct * 100.0;
  const uint64_t h = (uint64_t)std::nearbyint(x);
  p = put_u32(p, (uint32_t)(h / 100u));
  *p++ = '.';
  *p++ = (char)('0' + (h / 10u) % 10u);
  *p++ = (char)('0' + h % 10u);
  return p;
}

// one row; `name` = mod code (or `code,MOTIF,offset`); returns the end of the written text (at most ~200 bytes + chrom + name)
inline char* format_row(char* p, const char* chrom, size_t chrom_n, const char* name, size_t name_n, char sp, uint32_t pos, char strand,
                        uint32_t n_valid, uint32_t n_mod, uint32_t n_can, uint32_t n_other, uint32_t n_del, uint32_t n_fail, uint32_t n_diff,
                            uint32_t n_nocall) {
  memcpy(p, chrom, chrom_n); p += chrom_n; *p++ = '\t';
  // start, end and the coverage appear twice in a row (columns 2-3 = 7-8, 5 = 10): converted once, copied the second time
  char* const se = p; p = put_u32(p, pos); *p++ = '\t'; p = put_u32(p, pos + 1u); *p++ = '\t'; const size_t se_n = (size_t)(p - se);
  memcpy(p, name, name_n); p += name_n; *p++ = '\t';
  char* const cv = p; p = put_u32(p, n_valid); const size_t cv_n = (size_t)(p - cv); *p++ = '\t'; *p++ = strand; *p++ = '\t';
  memcpy(p, se, se_n); p += se_n;
  memcpy(p, "255,0,0\t", 8); p += 8;
  memcpy(p, cv, cv_n); p += cv_n; *p++ = sp;
  const float frac = (float)n_mod / (float)n_valid;   // fraction_modified (pileup/mod.rs:401), f32
  p = put_pct2(p, frac * 100.0f); *p++ = sp;
  p = put_u32(p, n_mod); *p++ = sp; p = put_u32(p, n_can); *p++ = sp; p = put_u32(p, n_other); *p++ = sp;
  p = put_u32(p, n_del); *p++ = sp; p = put_u32(p, n_fail); *p++ = sp; p = put_u32(p, n_diff); *p++ = sp;
  p = put_u32(p, n_nocall); *p++ = '\n';
  return p;
}

}  // namespace mkp
