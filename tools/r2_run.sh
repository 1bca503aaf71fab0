#!/bin/bash
tag=${1:-r02m}
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "bn_bwd_fused" 2>&1 | tail -5
for rep in 1 2; do
for v in "AYOLO_FUSE_BN_BWD=1" "AYOLO_FUSE_BN_BWD=0"; do
  env $v timeout 300 python bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print('$v', json.loads(sys.stdin.read())['ms_per_step'])" | tee -a $out/${tag}_ab.txt
done; done
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_tests.txt 2>&1; echo "tests rc=$?"; tail -3 $out/${tag}_tests.txt
