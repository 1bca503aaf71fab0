out=gpurun_out; mkdir -p $out
python tools/xf_sweep.py 2>&1 | grep -v amdgpu.ids > $out/r06_xf_forward_sweep.txt
cat $out/r06_xf_forward_sweep.txt
AYOLO_PW=3 python tools/xf_sweep.py 2>&1 | grep -v amdgpu.ids > $out/r06_xf_forward_sweep_pw3.txt
cat $out/r06_xf_forward_sweep_pw3.txt
