#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests -m gpu -q -x -n 8 2>&1 | tail -n 2
tools/gen_modbam --out /tmp/c2 --contig synth5m:5000000 --reads 100000 --seed 1 --style m --threads 8 >/dev/null
for i in 1 2; do modkit_amd/csrc/mkpileup pileup /tmp/c2.bam /tmp/c2.bed --stats 2>&1 | tail -n 1 | cut -c1-330; done
sha256sum /tmp/c2.bed
