#!/bin/bash
tag=${1:-r02j}
out=gpurun_out; mkdir -p $out
python -m pytest tests -m gpu -x -q -s > $out/${tag}_tests.txt 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_tests.txt
grep -n "passed\|failed\|largest relative\|full-size fp16\|Error" $out/${tag}_tests.txt | cut -c1-250 | head
python tools/eval_loop.py > /dev/null 2>&1
python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
print(bench.config_extras(torch.device("cuda"))["cfg5"])
PY
python bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print('bench', json.loads(sys.stdin.read())['ms_per_step'])"
