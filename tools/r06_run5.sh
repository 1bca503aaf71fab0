out=gpurun_out; mkdir -p $out
AYOLO_PW=1 timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q 2>&1 | tail -15 > $out/r06_pw_tests.txt
cat $out/r06_pw_tests.txt
export AYOLO_LIB=$PWD/ab/libayolo_probe.so
for shp in "64 128 128 1 1 0 40 40" "64 128 128 1 1 0 40 40 dgrad" "64 256 256 1 1 0 40 40" "64 256 256 1 1 0 20 20" "64 128 128 1 1 0 80 80"; do
  echo "=== PW=1 probe: B Cin Cout k s p H W = $shp"
  AYOLO_PW=1 python tools/gconv_probe.py $shp 2>&1 | grep -v amdgpu.ids
done > $out/r06_probe_pw_v1.txt
unset AYOLO_LIB
AYOLO_WGRAD_STREAM=0 python tools/op_table.py > $out/r06_op_table_isolated_base.txt 2>&1
AYOLO_WGRAD_STREAM=0 AYOLO_PW=1 python tools/op_table.py > $out/r06_op_table_isolated_pw.txt 2>&1
tail -11 $out/r06_op_table_isolated_base.txt | head -4; tail -11 $out/r06_op_table_isolated_pw.txt | head -4
bash tools/ab_trees.sh 3 40 "base=python bench.py" "pw=AYOLO_PW=1 python bench.py" "pw40=AYOLO_PW=1 AYOLO_PW_MAXM=102400 python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_pw_v1.txt
tail -4 $out/r06_ab_pw_v1.txt
