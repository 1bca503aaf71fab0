#!/bin/bash
# Isolated duration of the grouped weight-gradient launches (tools/op_table.py, every op alone on the chip) under the
# planner's item-length switches.  usage (GPU box): bash tools/wgrad_group_sweep.sh > gpurun_out/wgrad_group_sweep.txt
for cfg in "3 24" "1 24" "2 24" "6 24" "3 12" "3 48" "12 8"; do
  set -- $cfg
  echo "== AYOLO_WGRAD_WAVES=$1 AYOLO_WGRAD_MINQ=$2"
  AYOLO_WGRAD_STREAM=0 AYOLO_WGRAD_WAVES=$1 AYOLO_WGRAD_MINQ=$2 python tools/op_table.py 2>/dev/null | grep "OP_WGRAD_GROUP\|^conv_wgrad"
done
