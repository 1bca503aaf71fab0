out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "plan_equals_module_path or full_size_step_properties" 2>&1 | grep -E "plan vs module|full-size fp32|passed|failed" > $out/r06_test_model_numbers.txt
cat $out/r06_test_model_numbers.txt
timeout 900 python -m pytest tests/test_gpu_infer.py -m gpu -q -s -k "routes_agree or cfg3 or yolov5l" 2>&1 | grep -E "^fp16 step|cos|passed|failed|yolov5l" | cut -c1-300 > $out/r06_test_infer_numbers.txt
cat $out/r06_test_infer_numbers.txt
