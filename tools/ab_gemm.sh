set -e
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -3
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"achieved\"], {k:v[\"ms_per_step\"] for k,v in d[\"kernel_classes\"].items()})"
