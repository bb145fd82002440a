#!/bin/bash
# end-to-end C3 run of the CLI with the plan/run traces on: where does the host second go?  INF="own zlib" A/Bs the host inflate.
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/e2e_trace; mkdir -p $OUT
P=/tmp/mkp_c3_L64444167_N193000_x1_seed20
tools/gen_modbam --out $P --reads 193000 --seed 20 --threads 16 $GEN --contig chr20:64444167 > /dev/null
for I in ${INF:-own}; do for T in ${POOLS:-0}; do for i in 1 2 3; do
  if [ $T = 0 ]; then unset MKP_POOL_THREADS MKP_PACK_PIECES; else export MKP_POOL_THREADS=$T MKP_PACK_PIECES=$T; fi
  MKP_HOST_INFLATE=$I MKP_TRACE_PLAN=1 modkit_amd/csrc/mkpileup pileup $P.bam /tmp/o.bed --cpg --ref $P.fa --stats 2> $OUT/trace_${I}_${T}_$i.txt; echo "inflate $I pool $T: $(grep -o 'load_ms=[0-9.]*\|total_ms=[0-9.]*' $OUT/trace_${I}_${T}_$i.txt | tr '\n' ' ')"; done; done; done
sha256sum /tmp/o.bed
cp $OUT/trace_own_${BEST:-0}_3.txt $OUT/trace2.txt
grep "run\]\|load_ms" $OUT/trace2.txt | cut -c1-330
