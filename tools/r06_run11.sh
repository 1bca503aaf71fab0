out=gpurun_out; mkdir -p $out
for cfg in "AYOLO_PW=0" "AYOLO_PW=1" "AYOLO_PW=3" "AYOLO_PW=9" "AYOLO_PW=15" "AYOLO_PW=15 AYOLO_GCONV_NW8=1" "AYOLO_PW=0 AYOLO_GCONV_NW8=1" "AYOLO_PW=11 AYOLO_GCONV_NW8=1"; do
  echo "=== $cfg"
  env $cfg timeout 600 python -m pytest tests/test_gpu_infer.py -m gpu -q -s -k "train_step_fp32_and_fp16_vs_oracle or well_conditioned" 2>&1 | grep -E "^yolov5|^\.yolov5|^Fyolov5|passed|failed" | cut -c1-420
done > $out/r06_pw_fp16_numbers_v2.txt 2>&1
cat $out/r06_pw_fp16_numbers_v2.txt
