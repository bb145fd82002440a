cd $GRAFT_REPO_ROOT
tools/gen_modbam --out /tmp/c2 --contig synth5m:5000000 --reads 100000 --seed 1 --style m --threads 8 >/dev/null
for i in 1 2; do S=$(date +%s.%N); modkit_amd/csrc/mkpileup pileup /tmp/c2.bam /tmp/c2.bed --stats 2>&1 | tail -2 | sed -E 's/processed~[0-9]+ skipped~[0-9]+ //'; E=$(date +%s.%N); echo "wall $(echo "$E - $S" | bc) s"; done
ls -la /tmp/c2.bam /tmp/c2.bed | awk '{print $5, $9}'
