out=gpurun_out; mkdir -p $out
export AYOLO_LIB=$PWD/ab/libayolo_probe.so
for nw in 0 1; do
for shp in "64 128 128 1 1 0 40 40" "64 128 128 1 1 0 40 40 dgrad" "64 256 256 1 1 0 40 40" "64 512 512 1 1 0 20 20"; do
  echo "=== NW8=$nw k_gconv probe: B Cin Cout k s p H W = $shp"
  AYOLO_GCONV_NW8=$nw python tools/gconv_probe.py $shp 2>&1 | grep -v amdgpu.ids
done; done > $out/r06_probe_nw8_v1.txt
unset AYOLO_LIB
python tools/op_table.py > $out/r06_op_table_in_situ_base.txt 2>&1
AYOLO_GCONV_NW8=1 python tools/op_table.py > $out/r06_op_table_in_situ_nw8.txt 2>&1
AYOLO_WGRAD_STREAM=0 python tools/op_table.py > $out/r06_op_table_isolated_base.txt 2>&1
AYOLO_WGRAD_STREAM=0 AYOLO_GCONV_NW8=1 python tools/op_table.py > $out/r06_op_table_isolated_nw8.txt 2>&1
tail -11 $out/r06_op_table_isolated_base.txt; tail -11 $out/r06_op_table_isolated_nw8.txt
