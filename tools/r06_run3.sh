out=gpurun_out; mkdir -p $out
AYOLO_GCONV_NW8=1 timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -5 > $out/r06_nw8_tests.txt
cat $out/r06_nw8_tests.txt
AYOLO_GCONV_NW8=1 timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_infer.py -m gpu -x -q 2>&1 | tail -5 >> $out/r06_nw8_tests.txt
tail -5 $out/r06_nw8_tests.txt
python tools/conv_sweep.py yolov5s 64 640 2>&1 | grep -v amdgpu > $out/r06_sweep_base.txt
AYOLO_GCONV_NW8=1 python tools/conv_sweep.py yolov5s 64 640 2>&1 | grep -v amdgpu > $out/r06_sweep_nw8.txt
paste <(cut -c1-48 $out/r06_sweep_base.txt) <(cut -c22-48 $out/r06_sweep_nw8.txt) | head -40
bash tools/ab_trees.sh 3 40 "base=python bench.py" "nw8=AYOLO_GCONV_NW8=1 python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_nw8_v1.txt
tail -3 $out/r06_ab_nw8_v1.txt
