/* mkpileup — command-line front end of libmkpileup: `mkpileup pileup in.bam out.bed [modkit pileup flags]`,
 * `mkpileup pileup-hemi in.bam -o out.bed [modkit pileup-hemi flags]`, `mkpileup extract calls in.bam out.tsv [flags]`. */
#include <stdio.h>
#include <string.h>
#include "mkpileup.h"
int main(int argc, char** argv) {
  char err[1024] = {0};
  const int hemi = argc >= 2 && strcmp(argv[1], "pileup-hemi") == 0;
  const int extract = argc >= 3 && strcmp(argv[1], "extract") == 0 && strcmp(argv[2], "calls") == 0;
  if (extract) {
    int rc = mkp_extract_calls_main(argc - 3, (const char* const*)(argv + 3), err, sizeof(err));
    if (rc != MKP_OK) { fprintf(stderr, "Error! %s (status %d)\n", err, rc); return 1; }
    return 0;
  }
  if (argc < 2 || (!hemi && strcmp(argv[1], "pileup") != 0)) {
    fprintf(stderr, "usage: mkpileup pileup <in.bam> <out.bed> [flags of `modkit pileup`] [--device N] [--stats]\n"
                    "       mkpileup pileup-hemi <in.bam> -o <out.bed> [flags of `modkit pileup-hemi`] [--device N] [--stats]\n"
                    "       mkpileup extract calls <in.bam> <out.tsv> [flags of `modkit extract calls`] [--device N] [--stats]\n");
    return 2;
  }
  int rc = hemi ? mkp_pileup_hemi_main(argc - 2, (const char* const*)(argv + 2), err, sizeof(err)) : mkp_pileup_main(argc - 2,
      (const char* const*)(argv + 2), err, sizeof(err));
  if (rc != MKP_OK) { fprintf(stderr, "Error! %s (status %d)\n", err, rc); return 1; }
  return 0;
}
