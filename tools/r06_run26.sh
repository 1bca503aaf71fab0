out=gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_conv.py -m gpu -q 2>&1 | tail -12 > $out/r06_k512_conv_tests.txt; cat $out/r06_k512_conv_tests.txt
timeout 900 python -m pytest tests/test_gpu_infer.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -4
bash tools/ab_trees.sh 3 40 "before=AYOLO_LIB=$PWD/ab/libayolo_before_k512.so python bench.py" "k512=python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_k512.txt
tail -3 $out/r06_ab_k512.txt
AYOLO_WGRAD_STREAM=0 python tools/op_table.py 2>&1 | grep -E "512" | head -40 > $out/r06_op_table_isolated_k512_rows.txt; cat $out/r06_op_table_isolated_k512_rows.txt
