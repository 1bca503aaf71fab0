out=gpurun_out; mkdir -p $out
python tools/race_screen.py yolov5s 64 640 8 2>&1 | grep -v amdgpu.ids | tail -12 > $out/r06_race_screen.txt; cat $out/r06_race_screen.txt
python tools/race_screen.py yolov5l 16 640 4 2>&1 | grep -v amdgpu.ids | tail -4 >> $out/r06_race_screen.txt; tail -4 $out/r06_race_screen.txt
python tools/robust_check.py 2>&1 | grep -v amdgpu.ids | tail -12 > $out/r06_robust_check.txt; cat $out/r06_robust_check.txt
# whole-step reproducibility of the bench configuration: two processes, same seed -> identical loss sequence
for i in 1 2; do python bench.py --no-extras --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('loss'), d['ms_per_step'])"; done
