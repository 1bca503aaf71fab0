out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "sppf or maxpool" 2>&1 | tail -6
for n in 4 2 1; do AYOLO_SPPF_NCG=$n python tools/pool_time.py 2>&1 | grep -v amdgpu.ids; done > $out/r06_sppf_geometry.txt
cat $out/r06_sppf_geometry.txt
bash tools/ab_trees.sh 3 40 "sppf_off=AYOLO_SPPF_FUSED=0 python bench.py" "sppf_ncg4=python bench.py" "sppf_ncg2=AYOLO_SPPF_NCG=2 python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_sppf_v3.txt
tail -4 $out/r06_ab_sppf_v3.txt
