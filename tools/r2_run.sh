#!/bin/bash
tag=${1:-r02k}
out=gpurun_out; mkdir -p $out
for rep in 1 2; do
for v in "AYOLO_SIDE_PRIORITY=1" "AYOLO_SIDE_PRIORITY=0" "AYOLO_SIDE_PRIORITY=-1"; do
  env $v python bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print('$v', json.loads(sys.stdin.read())['ms_per_step'])" | tee -a $out/${tag}_ab.txt
done; done
python -m pytest tests/test_gpu_infer.py -m gpu -x -q -s 2>&1 | grep -n "passed\|failed\|largest relative\|full-size fp16\|Error" | cut -c1-300
