timeout 900 python -m pytest tests/test_gpu_infer.py -m gpu -q -x -k "routes_agree" 2>&1 | grep -E "^E |^tests.*Error|assert|passed|failed" | head -30
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q 2>&1 | tail -3
