"""The library's bedMethyl row formatter (modkit_amd/csrc/mkp_format.hpp; used by mkp_pileup_main) against printf:
`{:.2}` of an f32 percentage must be the correctly rounded decimal (ties to even), as Rust prints it
(src/writers.rs:140).  Exhaustive over every (n_mod, n_valid) with n_valid <= 4096, plus f32 edge values."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "mkp_format.hpp"
int main() {
  char a[512], b[512]; unsigned long long n = 0;
  for (unsigned v = 1; v <= 4096; v++) for (unsigned m = 0; m <= v; m++) {
    const float pct = ((float)m / (float)v) * 100.0f;
    char* e = mkp::put_pct2(a, pct); *e = 0;
    snprintf(b, sizeof b, "%.2f", (double)pct);
    if (strcmp(a, b)) { printf("MISMATCH %u/%u: %s vs %s\n", m, v, a, b); return 1; }
    n++;
  }
  // hand-picked f32 values sitting on or next to rounding ties
  const float edge[] = {0.005f, 0.015f, 0.125f, 0.375f, 2.675f, 99.995f, 100.0f, 0.0f, 1e-7f, 33.335f, 66.665f, 50.125f, 12.345f, 0.285f, 1.005f};
  for (float x : edge) { char* e = mkp::put_pct2(a, x); *e = 0; snprintf(b, sizeof b, "%.2f", (double)x); if (strcmp(a, b)) { printf("MISMATCH edge %g: %s vs %s\n", x, a, b); return 1; } }
  // a whole row against the printf formulation
  char* e = mkp::format_row(a, "chr20", 5, "m,CG,0", 6, ' ', 1234567u, '-', 37, 11, 26, 0, 2, 5, 1, 3); *e = 0;
  const float pct = ((float)11 / (float)37) * 100.0f;
  snprintf(b, sizeof b, "%s\t%u\t%u\t%s\t%u\t%c\t%u\t%u\t255,0,0\t%u%c%.2f%c%u%c%u%c%u%c%u%c%u%c%u%c%u\n", "chr20", 1234567u, 1234568u, "m,CG,0", 37u, '-', 1234567u, 1234568u, 37u, ' ',
           (double)pct, ' ', 11u, ' ', 26u, ' ', 0u, ' ', 2u, ' ', 5u, ' ', 1u, ' ', 3u);
  if (strcmp(a, b)) { printf("MISMATCH row:\n%s%s", a, b); return 1; }
  printf("ok %llu\n", n);
  return 0;
}
'''


def test_percent_formatting_matches_printf_exhaustively(tmp_path):
    src = tmp_path / "fmt.cpp"
    src.write_text(SRC)
    exe = tmp_path / "fmt"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "modkit_amd", "csrc"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)], text=True)
    assert out.startswith("ok 8394752"), out
