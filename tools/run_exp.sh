# experiment runner (GPU box): conv parity tests, per-layer sweeps, cfg 5 time -> gpurun_out/exp_<tag>_*
tag=${1:-a}
timeout 300 python -m pytest tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/exp_${tag}_tests.txt
timeout 200 python tools/conv_sweep.py yolov5s 64 640 > gpurun_out/exp_${tag}_sweep_s.txt 2>&1
timeout 200 python tools/cfg5_time.py > gpurun_out/exp_${tag}_cfg5.txt 2>&1
timeout 200 python tools/conv_sweep.py yolov5x 8 1280 > gpurun_out/exp_${tag}_sweep_x.txt 2>&1
timeout 300 python bench.py --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/exp_${tag}_bench.json
cat gpurun_out/exp_${tag}_tests.txt gpurun_out/exp_${tag}_cfg5.txt; tail -3 gpurun_out/exp_${tag}_sweep_s.txt; tail -3 gpurun_out/exp_${tag}_sweep_x.txt; python -c "
import json; d=json.load(open('gpurun_out/exp_${tag}_bench.json')); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"
