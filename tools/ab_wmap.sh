for m in 0 1 0 1; do
  AYOLO_WGRAD_TMAP=$m timeout 200 python tools/conv_sweep.py yolov5s 64 640 > gpurun_out/wmap_$m.txt 2>/dev/null; tail -1 gpurun_out/wmap_$m.txt | tr '\n' ' '; echo " <- tmap $m"
done
AYOLO_WGRAD_TMAP=1 timeout 200 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "conv" 2>&1 | tail -2
for m in 0 1 0 1; do
  ms=$(AYOLO_WGRAD_TMAP=$m python bench.py --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "tmap $m  $ms ms/step"
done
