out=gpurun_out; mkdir -p $out
python tools/grad_dump.py dump /tmp/g_base.pt 2>&1 | tail -1
AYOLO_PW=1 python tools/grad_dump.py dump /tmp/g_pw1.pt 2>&1 | tail -1
AYOLO_PW=3 python tools/grad_dump.py dump /tmp/g_pw3.pt 2>&1 | tail -1
AYOLO_PW=15 AYOLO_GCONV_NW8=1 python tools/grad_dump.py dump /tmp/g_pw15nw8.pt 2>&1 | tail -1
AYOLO_GCONV_TP=128 python tools/grad_dump.py dump /tmp/g_tp128.pt 2>&1 | tail -1
AYOLO_GCONV_TP=256 python tools/grad_dump.py dump /tmp/g_tp256.pt 2>&1 | tail -1
for x in pw1 pw3 pw15nw8 tp128 tp256; do echo "=== base vs $x"; python tools/grad_dump.py cmp /tmp/g_base.pt /tmp/g_$x.pt | grep -v "^  model"; done > $out/r06_grad_cmp_layers.txt 2>&1
grep -E "===|whole|loss" $out/r06_grad_cmp_layers.txt
