out=gpurun_out; mkdir -p $out
AYOLO_PW=7 timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q 2>&1 | tail -4 > $out/r06_pw7_tests.txt
cat $out/r06_pw7_tests.txt
AYOLO_WGRAD_STREAM=0 python tools/op_table.py > $out/r06_op_table_isolated_base.txt 2>&1
AYOLO_WGRAD_STREAM=0 AYOLO_PW=7 python tools/op_table.py > $out/r06_op_table_isolated_pw7.txt 2>&1
tail -11 $out/r06_op_table_isolated_base.txt | head -4; tail -11 $out/r06_op_table_isolated_pw7.txt | head -4
bash tools/ab_trees.sh 3 40 "base=python bench.py" "pw1=AYOLO_PW=1 python bench.py" "pw3=AYOLO_PW=3 python bench.py" "pw7=AYOLO_PW=7 python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_pw_v2.txt
tail -5 $out/r06_ab_pw_v2.txt
