set -x
export AYOLO_GCONV_TM=256
timeout 300 python tools/tm256_check.py > gpurun_out/tm256_check.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_infer.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/tm256_tests.txt
timeout 200 python tools/conv_sweep.py yolov5s 64 640 > gpurun_out/tm256_sweep_s.txt 2>&1
timeout 200 python tools/cfg5_time.py > gpurun_out/tm256_cfg5.txt 2>&1
timeout 200 python tools/conv_sweep.py yolov5x 8 1280 > gpurun_out/tm256_sweep_x.txt 2>&1
unset AYOLO_GCONV_TM
timeout 200 python tools/conv_sweep.py yolov5s 64 640 > gpurun_out/tm128_sweep_s.txt 2>&1
timeout 200 python tools/cfg5_time.py > gpurun_out/tm128_cfg5.txt 2>&1
timeout 200 python tools/conv_sweep.py yolov5x 8 1280 > gpurun_out/tm128_sweep_x.txt 2>&1
cat gpurun_out/tm256_check.txt; cat gpurun_out/tm256_tests.txt; cat gpurun_out/tm256_cfg5.txt gpurun_out/tm128_cfg5.txt
