#!/bin/bash
# experiments (host only): wall time of the packer on the bench BAM against the worker-pool size and the number of pack pieces
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-pk}; mkdir -p $OUT
tools/gen_modbam --out /tmp/pk --contig chr20:64444167 --reads 193000 --seed 20 --style hm --cpg-depleted --mean-len 8353 --threads 128 > $OUT/gen.json
E=modkit_amd/csrc/mkpileup
$E pileup /tmp/pk.bam /tmp/plan.tsv --plan-only --stats --cpg --ref /tmp/pk.fa 2> /dev/null
for cfg in "64 64" "64 256" "128 128" "128 512" "192 192" "32 32" "64 64"; do
  set -- $cfg
  for rep in 1 2; do
    MKP_POOL_THREADS=$1 MKP_PACK_PIECES=$2 $E pileup /tmp/pk.bam /tmp/plan.tsv --plan-only --stats --cpg --ref /tmp/pk.fa 2>&1 | grep -o "load_ms=[0-9.]* threshold_ms=[0-9.]* focus_ms=[0-9.]* pack_ms=[0-9.]*\|total_ms=[0-9.]* shards=[0-9]*" | tr '\n' ' '; echo " pool=$1 pieces=$2"
  done
done
