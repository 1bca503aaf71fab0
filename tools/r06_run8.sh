out=gpurun_out; mkdir -p $out
AYOLO_PW=15 timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q 2>&1 | tail -3
AYOLO_WGRAD_STREAM=0 python tools/op_table.py > $out/r06_op_table_isolated_base.txt 2>&1
AYOLO_WGRAD_STREAM=0 AYOLO_PW=15 python tools/op_table.py > $out/r06_op_table_isolated_pw15.txt 2>&1
bash tools/ab_trees.sh 3 40 "base=python bench.py" "pw11=AYOLO_PW=11 python bench.py" "pw15=AYOLO_PW=15 python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_pw_v4.txt
tail -4 $out/r06_ab_pw_v4.txt
