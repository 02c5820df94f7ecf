set -e
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention or mha16 or attn" 2>&1 | tail -3
timeout -k 10 200 python tools/attn_bench.py 2>&1 | grep -E "global|hfc|window"
