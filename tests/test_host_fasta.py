"""The library's reference FASTA loader (modkit_amd/csrc/mkp_bam.hpp, Fasta::load: the file read and its lines joined on all host
cores) against the one-thread, line-by-line formulation it replaced (Fasta::load_serial), byte for byte, on files with every oddity the
serial loader tolerates: CRLF line ends, blank lines, text in front of the first header, '>' inside header text, names that come twice,
records without a body, no line feed at the end, lines of any length, records larger than the loader's 1 MiB pieces."""
import os
import random
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include <cstring>
#include "mkp_bam.hpp"
int main(int argc, char** argv) {
  using namespace mkp;
  int bad = 0;
  for (int i = 1; i < argc; i++) {
    auto a = Fasta::load_serial(argv[i]); Fasta b = Fasta::load(argv[i]);
    bool eq = a.size() == b.seqs.size();
    for (auto& kv : a) { const FastaSeq* q = b.get(kv.first); eq = eq && q && q->size() == kv.second.size() && (q->size() == 0 || memcmp(q->data(), kv.second.data(), q->size()) == 0); }
    if (!eq) { printf("MISMATCH %s (%zu vs %zu records)\n", argv[i], a.size(), b.seqs.size()); bad++; }
  }
  printf("%s %d\n", bad ? "bad" : "ok", argc - 1);
  return bad ? 1 : 0;
}
'''


def _write_cases(d):
    rng = random.Random(11)
    paths = []

    def seq(n):
        return "".join(rng.choice("ACGTacgtNn") for _ in range(n))

    def wrap(s, w, eol):
        return eol.join(s[i:i + w] for i in range(0, len(s), w))

    cases = {
        "plain": ">chr1\n" + wrap(seq(1000), 60, "\n") + "\n>chr2 some text\n" + wrap(seq(333), 60, "\n") + "\n",
        "crlf": ">chr1\r\n" + wrap(seq(1000), 70, "\r\n") + "\r\n>chr2\tdesc\r\n" + wrap(seq(10), 70, "\r\n") + "\r\n",
        "no_final_newline": ">a\n" + wrap(seq(500), 50, "\n"),
        "no_final_newline_cr": ">a\r\n" + wrap(seq(500), 50, "\r\n") + "\r",
        "blank_lines": "\n\n>a\n\nACGT\n\n\nTTTT\n>b\n\n>c\nAC\n\n",
        "junk_in_front": "this is not a record\nACGT\n>real\nGGGG\nCCCC\n",
        "gt_in_header": ">a > b >c\nACGT\n>x>y\nTT\n",
        "same_name_twice": ">a\nAAAA\n>b\nCC\n>a\nGGGG\n>a extra\nT\n",
        "empty_bodies": ">a\n>b\n>c\nA\n>d\n",
        "empty_name": ">\nACGT\n> spaced\nTT\n",
        "cr_inside_line": ">a\nAC\rGT\r\r\nTT\r\n",
        "only_junk": "no header here\nat all\n",
        "empty_file": "",
        "one_long_line": ">a\n" + seq(300000) + "\n",
    }
    for name, text in cases.items():
        p = os.path.join(d, name + ".fa")
        with open(p, "w", newline="") as f:
            f.write(text)
        paths.append(p)
    # records larger than the pieces the loader works in, with line widths that do not divide them; LF and CRLF
    for name, eol, width, sizes in (("big_lf", "\n", 61, (3_300_000, 17, 1_100_000)), ("big_crlf", "\r\n", 83, (2_500_000, 1_048_576, 5))):
        p = os.path.join(d, name + ".fa")
        with open(p, "w", newline="") as f:
            for k, n in enumerate(sizes):
                block = seq(4096)
                s = (block * (n // 4096 + 1))[:n]
                f.write(">contig%d desc%s" % (k, eol) + wrap(s, width, eol) + eol)
        paths.append(p)
    return paths


def test_parallel_fasta_loader_equals_the_serial_one(tmp_path):
    src = tmp_path / "fa.cpp"
    src.write_text(SRC)
    exe = tmp_path / "fa"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "modkit_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           "-o", str(exe), str(src), "-lz"])
    paths = _write_cases(str(tmp_path))
    for threads in ("1", "3", "8"):   # the pool's size changes how the pieces are dealt out, never the result
        out = subprocess.check_output([str(exe)] + paths, text=True, env=dict(os.environ, MKP_POOL_THREADS=threads))
        assert out.strip().endswith("ok %d" % len(paths)), out
