out=gpurun_out; mkdir -p $out
python bench.py --no-extras --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/r06_base_bench.json
python -c "import json; d=json.load(open('$out/r06_base_bench.json')); print('base ms', d['ms_per_step'])"
export AYOLO_LIB=$PWD/ab/libayolo_probe.so
for shp in "64 128 128 1 1 0 40 40" "64 128 128 1 1 0 40 40 dgrad" "64 256 256 1 1 0 40 40" "64 512 512 1 1 0 20 20" "64 256 128 1 1 0 40 40" "64 512 256 1 1 0 20 20"; do
  echo "=== k_gconv probe: B Cin Cout k s p H W = $shp"
  python tools/gconv_probe.py $shp 2>&1 | grep -v amdgpu.ids
done > $out/r06_probe_small_maps_full.txt
unset AYOLO_LIB
python tools/op_table.py > $out/r06_base_op_table_in_situ.txt 2>&1
AYOLO_WGRAD_STREAM=0 python tools/op_table.py > $out/r06_base_op_table_isolated.txt 2>&1
tail -12 $out/r06_base_op_table_in_situ.txt
