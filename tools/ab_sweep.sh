# same-box A/B of the per-layer conv sweep and of the no-overlap step: base vs current
for lib in ab/libayolo_base.so ayolov2_amd/libayolo_hip.so ab/libayolo_base.so ayolov2_amd/libayolo_hip.so; do
  tag=$(basename $lib .so)
  AYOLO_LIB=$(realpath $lib) timeout 200 python tools/conv_sweep.py yolov5s 64 640 2>/dev/null | tail -3 | tr '\n' ' '; echo " <- $tag"
done
for lib in ab/libayolo_base.so ayolov2_amd/libayolo_hip.so ab/libayolo_base.so ayolov2_amd/libayolo_hip.so; do
  ms=$(AYOLO_WGRAD_STREAM=0 AYOLO_LIB=$(realpath "$lib") python bench.py --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "no-overlap $lib  $ms ms/step"
done
