#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/gen_modbam --out /tmp/c2 --contig synth5m:5000000 --reads 100000 --seed 1 --style m --threads 8 >/dev/null
for i in 1 2 3; do S=$(date +%s.%N); modkit_amd/csrc/mkpileup pileup /tmp/c2.bam /tmp/c2.bed --stats 2>&1 | tail -n 3; E=$(date +%s.%N); echo "wall $(echo "$E - $S" | bc) s"; done
sha256sum /tmp/c2.bed; ls -la /tmp/c2.bed | awk '{print $5}'
make -C oracle modkit_oracle >/dev/null 2>&1
S=$(date +%s.%N); oracle/modkit_oracle pileup /tmp/c2.bam /tmp/c2.oracle.bed --oracle-workers 16 2>&1 | tail -n 1; E=$(date +%s.%N); echo "oracle wall $(echo "$E - $S" | bc) s"
sha256sum /tmp/c2.oracle.bed
