out=gpurun_out; mkdir -p $out
for pw in 1 3 5 9 15; do
  echo "=== AYOLO_PW=$pw"
  AYOLO_PW=$pw timeout 600 python -m pytest tests/test_gpu_infer.py -m gpu -q -x -k "test_yolov5s_train_step_fp32_and_fp16_vs_oracle" 2>&1 | grep -E "^E  |passed|failed" | head -12
done > $out/r06_pw_bisect.txt 2>&1
cat $out/r06_pw_bisect.txt
