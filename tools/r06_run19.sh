out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "sppf or maxpool" 2>&1 | tail -6 > $out/r06_sppf_v2_tests.txt; cat $out/r06_sppf_v2_tests.txt
python tools/winograd_proxy.py 8 320 320 160 160  8 640 640 80 80  8 160 160 320 320  8 1280 1280 40 40  64 128 128 40 40  64 256 256 20 20 2>&1 | grep -v amdgpu.ids > $out/r06_winograd_proxy.txt; cat $out/r06_winograd_proxy.txt
bash tools/ab_trees.sh 3 40 "sppf_off=AYOLO_SPPF_FUSED=0 python bench.py" "sppf_v2=python bench.py" "wgrad3_off=AYOLO_WGRAD3=0 python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_sppf_v2_wgrad3.txt
tail -4 $out/r06_ab_sppf_v2_wgrad3.txt
python tools/op_table.py 2>&1 | grep -E "SPPF|pool_upsample" | head
