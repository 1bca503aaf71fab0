out=gpurun_out; mkdir -p $out
AYOLO_PW=15 AYOLO_GCONV_NW8=1 timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $out/r06_pw15_nw8_full_tests.txt
cat $out/r06_pw15_nw8_full_tests.txt
bash tools/ab_trees.sh 3 40 "base=python bench.py" "pw15=AYOLO_PW=15 python bench.py" "pw15_nw8=AYOLO_PW=15 AYOLO_GCONV_NW8=1 python bench.py" "pw11_nw8=AYOLO_PW=11 AYOLO_GCONV_NW8=1 python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_pw_nw8.txt
tail -5 $out/r06_ab_pw_nw8.txt
AYOLO_PW=15 AYOLO_GCONV_NW8=1 python tools/op_table.py > $out/r06_op_table_in_situ_pw15_nw8.txt 2>&1
AYOLO_WGRAD_STREAM=0 AYOLO_PW=15 AYOLO_GCONV_NW8=1 python tools/op_table.py > $out/r06_op_table_isolated_pw15_nw8.txt 2>&1
tail -12 $out/r06_op_table_in_situ_pw15_nw8.txt
