#!/bin/bash
# VERDICT r5 item 9 / hygiene: what synchronised BatchNorm (train_model_builder.py:135-136; ~100 small in-stream collectives per step) costs
# the step, on the one GPU a box has: the exchange forced on a single-rank RCCL group, YOLOv5s batch 64 and YOLOv5l batch 4.
# usage (GPU box): bash tools/sync_bn_time.sh r06
tag=${1:-r06}
mkdir -p gpurun_out
for cfg in "yolov5s 64" "yolov5l 4"; do
  set -- $cfg
  for sb in "" "--sync-bn"; do
    AYOLO_FORCE_DDP=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 \
      bench.py --gpus 1 --model $1 --batch $2 --steps 20 --warmup 5 --no-extras $sb 2>/dev/null | tail -1 | \
      python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1 batch $2 forced one-rank exchange ${sb:-(no sync_bn)}: %.3f ms/step' % d['ms_per_step'])"
  done
done | tee gpurun_out/${tag}_sync_bn_forced_one_rank.txt
