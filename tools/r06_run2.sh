out=gpurun_out; mkdir -p $out
bash tools/ab_trees.sh 3 40 \
  "base=python bench.py" \
  "deferC_26=AYOLO_WGRAD_DEFER=26,0,0,0 python bench.py" \
  "deferB_26_8=AYOLO_WGRAD_DEFER=26,8,0,0 python bench.py" \
  "deferD_19_7=AYOLO_WGRAD_DEFER=19,7,0,0 python bench.py" \
  "lds55k=AYOLO_WGRAD_LDS=55296 python bench.py" \
  "lds82k=AYOLO_WGRAD_LDS=83968 python bench.py" \
  "mainstream=AYOLO_WGRAD_STREAM=0 python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_fork_placement_1.txt
cat $out/r06_ab_fork_placement_1.txt | tail -8
