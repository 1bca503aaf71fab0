#!/bin/bash
tag=${1:-r02g}
out=gpurun_out; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/${tag}_tests.txt 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_tests.txt
tail -25 $out/${tag}_tests.txt
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; python -c "
import json; d=json.load(open('$out/${tag}_bench.json')); print(d['ms_per_step'], d['value']); print(json.dumps(d['extra'], indent=0)[:3000])"; tail -3 $out/${tag}_bench.err
