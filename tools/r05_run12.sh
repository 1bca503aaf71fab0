#!/bin/bash
mkdir -p gpurun_out
timeout 900 bash tools/ab_trees.sh 3 30 "off=python bench.py" "w3_all=AYOLO_WGRAD3=1 python bench.py" "w3_ge80=AYOLO_WGRAD3=1 AYOLO_WGRAD3_MINHW=80 python bench.py" "w3_ge160=AYOLO_WGRAD3=1 AYOLO_WGRAD3_MINHW=160 python bench.py" > gpurun_out/r05_ab_w3_minhw.txt 2>&1
tail -5 gpurun_out/r05_ab_w3_minhw.txt
