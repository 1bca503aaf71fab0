#!/bin/bash
# BASELINE cfg 3 under the reference's convention is 4 images per GPU (32 over 8 GPUs,
# scripts/data_loader/data_loader_utils.py:67).  One GPU can measure the part of the 8-GPU prediction that does not need the other
# seven: YOLOv5l at batch 4 with the bucketed exchange forced on a single-rank RCCL group -- the backward window, when each
# bucket's all-reduce is complete inside it, what stays exposed.  usage (GPU box): bash tools/cfg3_b4.sh r05
tag=${1:-r05}
mkdir -p gpurun_out
AYOLO_FORCE_DDP=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 \
  bench.py --gpus 1 --model yolov5l --batch 4 --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 > gpurun_out/${tag}_cfg3_b4_forced_ddp.json
timeout 300 python bench.py --model yolov5l --batch 4 --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 > gpurun_out/${tag}_cfg3_b4_no_ddp.json
python - <<PY
import json
a = json.loads(open("gpurun_out/${tag}_cfg3_b4_forced_ddp.json").read())
b = json.loads(open("gpurun_out/${tag}_cfg3_b4_no_ddp.json").read())
print("yolov5l batch 4: %.3f ms/step with the exchange forced (one rank), %.3f without" % (a["ms_per_step"], b["ms_per_step"]))
print(json.dumps(a.get("ddp"), indent=1))
PY
