#!/bin/bash
# --device-inflate A/B on the C3 bench BAM, end to end
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3k; mkdir -p $OUT
P=/tmp/mkp_c3_L64444167_N193000_x1_seed20
tools/gen_modbam --out $P --reads 193000 --seed 20 --threads 16 --style hm --cpg-depleted --mean-len 8353 --contig chr20:64444167 > /dev/null
for M in host device device; do for i in 1 2; do
  F=""; [ $M = device ] && F="--device-inflate"
  MKP_TRACE_PLAN=1 modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o_$M.bed --cpg --ref $P.fa --stats $F 2> $OUT/trace_${M}_$i.txt; echo "$M: $(grep -o 'load_ms=[0-9.]*\|threshold_ms=[0-9.]*\|pack_ms=[0-9.]*\|total_ms=[0-9.]*\|on the device [0-9]*' $OUT/trace_${M}_$i.txt | tr '\n' ' ')"; done; done
sha256sum /tmp/o_host.bed /tmp/o_device.bed
grep "run\]" $OUT/trace_device_2.txt
