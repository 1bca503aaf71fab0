out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "sppf or maxpool" 2>&1 | tail -15 > $out/r06_sppf_tests.txt; cat $out/r06_sppf_tests.txt
python tools/grad_dump.py dump /tmp/g_base.pt 2>&1 | tail -1
i=0
for cfg in "AYOLO_PW=1" "AYOLO_PW=1" "AYOLO_PW=15 AYOLO_GCONV_NW8=1" "AYOLO_PW=15 AYOLO_GCONV_NW8=1" "AYOLO_BNR=0" "AYOLO_XF=0" "AYOLO_MERGE_SIBLINGS=0" "AYOLO_XF_FIN=0" "AYOLO_GCONV_LIN=0" "AYOLO_WGRAD_STREAM=0" "AYOLO_PW=1 AYOLO_WGRAD_STREAM=0" "AYOLO_PW=1 AYOLO_BNR=0" "AYOLO_PW=1 AYOLO_XF=0"; do
  i=$((i+1))
  env $cfg python tools/grad_dump.py dump /tmp/g_$i.pt 2>&1 | tail -1
  echo "=== base vs [$i] $cfg"; python tools/grad_dump.py cmp /tmp/g_base.pt /tmp/g_$i.pt | grep -E "whole|dzs 03|dzs 04|dzs 10"
done > $out/r06_grad_cmp_neutral.txt 2>&1
echo "=== [1] vs [2] (PW=1 twice)" >> $out/r06_grad_cmp_neutral.txt; python tools/grad_dump.py cmp /tmp/g_1.pt /tmp/g_2.pt | grep -E "whole|dzs 03" >> $out/r06_grad_cmp_neutral.txt
echo "=== [3] vs [4] (PW=15 NW8 twice)" >> $out/r06_grad_cmp_neutral.txt; python tools/grad_dump.py cmp /tmp/g_3.pt /tmp/g_4.pt | grep -E "whole|dzs 03" >> $out/r06_grad_cmp_neutral.txt
cat $out/r06_grad_cmp_neutral.txt
bash tools/ab_trees.sh 3 40 "base=AYOLO_SPPF_FUSED=0 python bench.py" "sppf=python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_sppf.txt
tail -3 $out/r06_ab_sppf.txt
python tools/op_table.py 2>&1 | grep -E "SPPF|MAXPOOL|pool_upsample" | head
