#!/bin/bash
# validation of the ingest-ahead path: ingest/scale/fuzz tests, then the c5 / c4 scale models
TAG=${1:-r4h}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_scale.py tests/test_gpu_loud_failures.py -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest_a.log; cat $OUT/pytest_a.log
timeout 900 python -m pytest tests/ -m gpu -x -q -k "fuzz or bed or region or multi" 2>&1 | tail -5 > $OUT/pytest_b.log; cat $OUT/pytest_b.log
bash tools/gpu_scale_models.sh $TAG 2>&1 | tail -12
