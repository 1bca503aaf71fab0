#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_trainer.py -m gpu -x -q 2>&1 | grep -E "assert|Error|error|ACTUAL|DESIRED|Mismatch|passed|failed" | head -20
