out=gpurun_out; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $out/r06_full_tests_defaults_v1.txt
cat $out/r06_full_tests_defaults_v1.txt
bash tools/ab_trees.sh 3 40 "r5_equiv=AYOLO_PW=0 AYOLO_GCONV_NW8=0 AYOLO_BN_APPLY2=0 AYOLO_SPPF_FUSED=0 python bench.py" "now=python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_defaults_v1.txt
tail -3 $out/r06_ab_defaults_v1.txt
python tools/op_table.py > $out/r06_op_table_in_situ_v1.txt 2>&1
AYOLO_WGRAD_STREAM=0 python tools/op_table.py > $out/r06_op_table_isolated_v1.txt 2>&1
tail -12 $out/r06_op_table_in_situ_v1.txt
