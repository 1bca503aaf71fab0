#!/bin/bash
# one GPU-box call of round 2 (outputs under gpurun_out/)
tag=${1:-r02e}
out=gpurun_out; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/${tag}_tests.txt 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_tests.txt
tail -15 $out/${tag}_tests.txt
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; cat $out/${tag}_bench.json; tail -5 $out/${tag}_bench.err
AYOLO_FORCE_DDP=1 python bench.py --no-extras --steps 20 --warmup 5 2>>$out/${tag}_bench.err | tee gpurun_out/ddp_ov.out | tail -1 | python -c "import sys,json; print('force-ddp overlap', json.loads(sys.stdin.read())['ms_per_step'])" | tee $out/${tag}_ab.txt
AYOLO_FORCE_DDP=1 AYOLO_DDP_OVERLAP=0 python bench.py --no-extras --steps 20 --warmup 5 2>>$out/${tag}_bench.err | tee gpurun_out/ddp_single.out | tail -1 | python -c "import sys,json; print('force-ddp single allreduce', json.loads(sys.stdin.read())['ms_per_step'])" | tee -a $out/${tag}_ab.txt
for v in "AYOLO_WGRAD_FIRST=1" "AYOLO_WGRAD_BPC=1" "AYOLO_WGRAD_BPC=2" "AYOLO_WGRAD_FIRST=1 AYOLO_WGRAD_BPC=2"; do
  env $v python bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print('$v', json.loads(sys.stdin.read())['ms_per_step'])" | tee -a $out/${tag}_ab.txt
done
python bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print('default', json.loads(sys.stdin.read())['ms_per_step'])" | tee -a $out/${tag}_ab.txt
