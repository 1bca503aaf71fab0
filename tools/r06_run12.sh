out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "merged_pair" 2>&1 | tail -5 > $out/r06_apply2_tests.txt
cat $out/r06_apply2_tests.txt
AYOLO_BN_APPLY2=0 python tools/grad_dump.py dump /tmp/g_base.pt 2>&1 | tail -1
AYOLO_BN_APPLY2=0 python tools/grad_dump.py dump /tmp/g_base2.pt 2>&1 | tail -1
python tools/grad_dump.py dump /tmp/g_apply2.pt 2>&1 | tail -1
AYOLO_BN_APPLY2=0 AYOLO_PW=1 python tools/grad_dump.py dump /tmp/g_pw1.pt 2>&1 | tail -1
AYOLO_BN_APPLY2=0 AYOLO_PW=3 python tools/grad_dump.py dump /tmp/g_pw3.pt 2>&1 | tail -1
AYOLO_BN_APPLY2=0 AYOLO_PW=15 python tools/grad_dump.py dump /tmp/g_pw15.pt 2>&1 | tail -1
AYOLO_BN_APPLY2=0 AYOLO_GCONV_NW8=1 python tools/grad_dump.py dump /tmp/g_nw8.pt 2>&1 | tail -1
AYOLO_BN_APPLY2=0 AYOLO_PW=15 AYOLO_GCONV_NW8=1 python tools/grad_dump.py dump /tmp/g_pw15nw8.pt 2>&1 | tail -1
for x in base2 apply2 pw1 pw3 pw15 nw8 pw15nw8; do echo "=== base vs $x"; python tools/grad_dump.py cmp /tmp/g_base.pt /tmp/g_$x.pt; done > $out/r06_grad_cmp.txt 2>&1
grep -E "===|whole|loss" $out/r06_grad_cmp.txt
bash tools/ab_trees.sh 3 40 "base=AYOLO_BN_APPLY2=0 python bench.py" "apply2=python bench.py" 2>&1 | grep -v amdgpu.ids > $out/r06_ab_apply2.txt
tail -3 $out/r06_ab_apply2.txt
