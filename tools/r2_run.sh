#!/bin/bash
# one GPU-box call of round 2 (outputs under gpurun_out/)
tag=${1:-r02d}
out=gpurun_out; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/${tag}_tests.txt 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_tests.txt
tail -5 $out/${tag}_tests.txt
python bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print('own-sgd', json.loads(sys.stdin.read())['ms_per_step'])" | tee $out/${tag}_ab.txt
AYOLO_TORCH_SGD=1 python bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print('torch-sgd', json.loads(sys.stdin.read())['ms_per_step'])" | tee -a $out/${tag}_ab.txt
python tools/conv_sweep.py > $out/${tag}_conv_sweep.txt 2>&1; tail -4 $out/${tag}_conv_sweep.txt
python tools/host_profile.py > $out/${tag}_host.txt 2>&1; head -3 $out/${tag}_host.txt; tail -5 $out/${tag}_host.txt
bash tools/profile_round.sh $tag
head -30 $out/${tag}_kernel_stats.csv | cut -c1-150
cat $out/${tag}_pmc_top.txt
