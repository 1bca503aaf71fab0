out=gpurun_out; mkdir -p $out
AYOLO_PW=9 timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q 2>&1 | tail -6 > $out/r06_pw9_tests.txt
cat $out/r06_pw9_tests.txt
AYOLO_WGRAD_STREAM=0 python tools/op_table.py > $out/r06_op_table_isolated_base.txt 2>&1
AYOLO_WGRAD_STREAM=0 AYOLO_PW=9 python tools/op_table.py > $out/r06_op_table_isolated_pw9.txt 2>&1
tail -11 $out/r06_op_table_isolated_base.txt | head -4; tail -11 $out/r06_op_table_isolated_pw9.txt | head -4
bash tools/ab_trees.sh 3 40 "base=python bench.py" "pw1=AYOLO_PW=1 python bench.py" "pw9=AYOLO_PW=9 python bench.py" "pw11=AYOLO_PW=11 python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_pw_v3.txt
tail -5 $out/r06_ab_pw_v3.txt
