#!/bin/bash
tag=${1:-r02r}
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -s > $out/${tag}_tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|cosine" $out/${tag}_tests.txt | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; python -c "
import json; d=json.load(open('$out/${tag}_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('peak_memory_gb_train_step')); print({k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in list(v)[:6]}) for k, v in d['extra'].items() if k.startswith('cfg')})"
