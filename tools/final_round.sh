#!/bin/bash
# Round-end validation on one GPU box: full GPU tests, smoke, full bench line, rocprofv3 kernel stats + PMC HBM traffic + SQ counters of
# the train step, kernel stats of cfg 5, the DDP code path of bench.py under torchrun (one rank, exchange forced).
# usage (GPU box, repo root): bash tools/final_round.sh r03
tag=${1:-r05}
out=gpurun_out; mkdir -p $out
[ -z "$SKIP_TESTS" ] && timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $out/${tag}_gpu_tests_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.txt 2>&1
# inline-asm kernel-argument prefetch: its destination SGPRs must survive until the wait (tools/check_kernarg_touch.py, no GPU needed)
timeout 600 python tools/check_kernarg_touch.py ayolov2_amd/csrc/conv.hip ayolov2_amd/csrc/elementwise.hip 2>&1 | tail -3 > $out/${tag}_kernarg_touch_check.txt
# kernel stats + the two PMC passes first: bench.py's roofline.traffic then cites THIS tree's summary (copy it into profiles/)
bash tools/profile_round.sh $tag > /dev/null 2>&1
cp $out/${tag}_pmc_hbm_traffic.json profiles/ 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/${tag}_bench_n1.json
bash tools/pmc_sq.sh $tag > /dev/null 2>&1
export TMPDIR=/tmp; root=$(pwd)
(cd /tmp && rm -rf /tmp/prof_c5 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o c5 -- python $root/tools/cfg5_time.py > $root/$out/${tag}_cfg5_under_rocprof.txt 2>/dev/null; cp $(find /tmp/prof_c5 -name '*kernel_stats.csv' | head -1) $root/$out/${tag}_eval_cfg5_kernel_stats.csv)
AYOLO_FORCE_DDP=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 > $out/${tag}_bench_torchrun_forced_ddp.json
python tools/config_bench.py 3 2>/dev/null | tail -1 > $out/${tag}_cfg3.json
AYOLO_WGRAD_STREAM=0 python tools/op_table.py yolov5l 32 > $out/${tag}_op_table_isolated_yolov5l.txt 2>&1
python tools/op_table.py > $out/${tag}_op_table_in_situ.txt 2>&1
AYOLO_WGRAD_STREAM=0 python tools/op_table.py > $out/${tag}_op_table_isolated.txt 2>&1
cat $out/${tag}_gpu_tests_tail.txt $out/${tag}_smoke.txt 2>/dev/null; python - <<PY
import json
for f in ("$out/${tag}_bench_n1.json", "$out/${tag}_bench_torchrun_forced_ddp.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
