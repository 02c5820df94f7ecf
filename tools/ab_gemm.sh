set -e
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
for f in 0 1; do
echo "== WM_LN_FUSE=$f"
WM_LN_FUSE=$f timeout -k 10 300 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"achieved\"], {k:v[\"ms_per_step\"] for k,v in d[\"kernel_classes\"].items()})"
done
