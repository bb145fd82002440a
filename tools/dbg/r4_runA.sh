#!/bin/bash
# the whole GPU suite on the build with mkp_inflate_wave2 as the shard-sized inflate kernel, the RCCL one-rank test on its own, the c5 / c4 scale models
TAG=${1:-r4A}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
NCCL_DEBUG=WARN timeout 300 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k rccl 2>&1 | grep -v "^$" | tail -40 | cut -c1-260 > $OUT/rccl.log; grep -E "passed|failed|WARN|Librccl" $OUT/rccl.log | head
bash tools/gpu_scale_models.sh $TAG 2>&1 | tail -8 | cut -c1-700
