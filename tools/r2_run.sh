#!/bin/bash
# one GPU-box call of round 2: tests, A/B bench lines, elementwise sweep (outputs under gpurun_out/)
tag=${1:-r02a}
out=gpurun_out; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/${tag}_tests.txt 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_tests.txt
tail -5 $out/${tag}_tests.txt
AYOLO_MERGE_SIBLINGS=1 python bench.py --no-extras --steps 20 --warmup 5 > $out/${tag}_bench_merge1.json 2> $out/${tag}_bench.err; cat $out/${tag}_bench_merge1.json
AYOLO_MERGE_SIBLINGS=0 python bench.py --no-extras --steps 20 --warmup 5 > $out/${tag}_bench_merge0.json 2>> $out/${tag}_bench.err; cat $out/${tag}_bench_merge0.json
python tools/ew_sweep.py > $out/${tag}_ew.txt 2>&1; cat $out/${tag}_ew.txt
