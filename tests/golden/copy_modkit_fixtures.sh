#!/bin/bash
# Provenance: these are DATA fixtures (modBAM inputs, reference FASTA/BED, and the golden
# bedMethyl outputs) of nanoporetech/modkit v0.4.4's own pileup tests (tests/test_pileup.rs),
# copied verbatim from /root/reference/tests/resources so the parity tests can run on the
# GPU box where /root/reference does not exist.  No reference SOURCE code is copied.
set -e
SRC=${1:-/root/reference/tests/resources}
DST=$(dirname "$0")/modkit_fixtures
mkdir -p "$DST"
CG=CG_5mC_20230207_1700_6A_PAG66026_3c0abf27_oligo_741_adapters_modcalls_0th_sort_10_reads
for f in bc_anchored_10_reads.sorted.bam bc_anchored_10_reads.sorted.bam.bai \
  duplex_modbam.sorted.bam duplex_modbam.sorted.bam.bai \
  HG002_small.ch20._other.sorted.bam HG002_small.ch20._other.sorted.bam.bai \
  duplicated.marked.fixed.bam duplicated.marked.fixed.bam.bai \
  empty-tags.sorted.bam empty-tags.sorted.bam.bai \
  $CG.bam $CG.bam.bai $CG-2.bam $CG-2.bam.bai \
  CGI_ladder_3.6kb_ref.fa CGI_ladder_3.6kb_ref.fa.fai CGI_ladder_3.6kb_ref_include_positions.bed \
  modbam.modpileup_nofilt.methyl.bed pileup_with_header.bed modbam.modpileup_filt025.methyl.bed \
  modbam.modpileup_combined.methyl.bed modbam.modpileup_nofilt_oligo_1512_adapters_10_50.bed \
  duplex_modbam_pileup_nofilt.bed bc_anchored_10_reads_nofilt_cg_motif.bed \
  bc_anchored_10_reads_nofilt_cg_motif_strand_combine.bed bc_anchored_10_reads_edge_filter50.bed \
  bc_anchored_10_reads_edge_filter50-0.bed modbam.modpileup_filt_positions_025.methyl.bed \
  modbam.modpileup_filt_positions_025_traditional.methyl.bed cgcg2_cg0_test1.bed cgcg2_cg0_test2.bed \
  cgcg2_cg0_test1_combine_strands.bed cgcg2_cg0_test2_combine_strands.bed \
  pileup-old-tags-regressiontest.methyl.bed \
  bc_anchored_10_reads.haplotyped.sorted.bam bc_anchored_10_reads.haplotyped.sorted.bam.bai \
  duplex_modcalls_sort.bam duplex_modcalls_sort.bam.bai duplex_hemi_nofilt.bed duplex_hemi.bed \
  single_read.bam include_bed_summary_test.bed \
  2_reads_all_context.bam supplementary_and_secondary_read.bam test_read_calls_estimate_thresh.tsv test_supplementary_calls.tsv; do
  cp "$SRC/$f" "$DST/$f"
done
# pileup-hemi (tests/test_pileup_hemi.rs) needs GRCh38_chr20.fa, which the checkout does not ship: the slice the test region
# touches is rebuilt from the MD tags of the test BAM (exact at every covered position)
python3 "$(dirname "$0")/make_hemi_reference.py" "$SRC/duplex_modcalls_sort.bam" "$DST/chr20_hemi_slice.txt"
echo "copied $(ls "$DST" | wc -l) files"
