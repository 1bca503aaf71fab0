out=gpurun_out; mkdir -p $out
bash tools/ab_trees.sh 3 40 "base=python bench.py" "main_high=AYOLO_BENCH_STREAM_PRIO=-1 python bench.py" "main_normal_own=AYOLO_BENCH_STREAM_PRIO=0 python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_main_stream_priority.txt
cat $out/r06_ab_main_stream_priority.txt | tail -4
