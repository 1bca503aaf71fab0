#!/bin/bash
tag=${1:-r02}
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_infer.py tests/test_gpu_trainer.py -m gpu -q -s 2>&1 | grep -E "passed|failed|fp16 vs|full-size" | cut -c1-400
bash tools/profile_round.sh $tag
bash tools/pmc_sq.sh $tag > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/${tag}_kernel_stats_train_only.csv")))
steps=13
fam={}
for r in rows:
    n=r["Name"]; t=float(r["TotalDurationNs"])/steps/1e6
    key=("k_gconv" if "k_gconv" in n else "k_wgrad" if "k_wgrad" in n else "k_bn_bwd_reduce" if "bn_bwd_reduce" in n else "k_bn_bwd_apply" if "bn_bwd_apply" in n else "k_bn_train_act" if "bn_train_act" in n else n[:40])
    fam[key]=fam.get(key,0)+t
for k,v in sorted(fam.items(), key=lambda kv:-kv[1])[:16]: print("%8.3f ms/step  %s"%(v,k))
print("sum %.3f"%sum(fam.values()))
PY
cat $out/${tag}_pmc_top.txt | cut -c1-200
cat $out/${tag}_pmc_sq.txt 2>/dev/null | head -20 | cut -c1-200
