out=gpurun_out; mkdir -p $out
python tools/grad_dump.py dump /tmp/g_base.pt 2>&1 | tail -1
AYOLO_PW=1 python tools/grad_dump.py dump /tmp/g_pw1.pt 2>&1 | tail -1
AYOLO_PW=3 python tools/grad_dump.py dump /tmp/g_pw3.pt 2>&1 | tail -1
for x in pw1 pw3; do echo "=== base vs $x"; python tools/grad_dump.py cmp /tmp/g_base.pt /tmp/g_$x.pt | grep -E "loss grad|bn 00|skipped"; done > $out/r06_grad_cmp_loss.txt 2>&1
cat $out/r06_grad_cmp_loss.txt
