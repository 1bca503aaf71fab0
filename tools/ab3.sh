# A/B/C on one box: base (HEAD), current, current + kernarg preload; then conv parity + sweeps with the current lib
for i in 1 2; do
  for lib in ab/libayolo_base.so ayolov2_amd/libayolo_hip.so ab/libayolo_kp.so; do
    ms=$(AYOLO_LIB=$(realpath "$lib") python bench.py --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$lib  $ms ms/step"
  done
done
