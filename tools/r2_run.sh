#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_infer.py tests/test_gpu_conv.py -m gpu -q -s -k "full_size or yolov5l" 2>&1 | grep -E "passed|failed|Error|assert|cosine|yolov5l" | cut -c1-300 | tail -30
