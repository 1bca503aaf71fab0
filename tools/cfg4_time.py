#!/usr/bin/env python3
"""cfg 4 (Tucker-decomposed YOLOv5s, batch 128, 640x640, fp16 eval) per launch form of the decomposed blocks
(AYOLO_TUCKER_FORM is read at import: one process per form).  usage (GPU box): python tools/cfg4_time.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import bench
    out = bench.config_extras(torch.device("cuda", 0), only="cfg4") if "only" in bench.config_extras.__code__.co_varnames else bench.config_extras(torch.device("cuda", 0))
    print(json.dumps(out["cfg4"]))
else:
    for form in ("auto", "factors", "first", "last", "dense"):
        env = dict(os.environ, AYOLO_TUCKER_FORM=form)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
        try:
            d = json.loads(line)
            print(f"{form:8s} {d['ms_per_batch']:7.2f} ms  (dense model {d['undecomposed_ms_per_batch']:.2f} ms)  forms {d.get('launch_forms')}")
        except Exception:
            print(form, "FAILED", line)
