set -e
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for args in "--batch 1" "--batch 2" "--batch 4"; do
  echo "== $args"
  timeout -k 10 300 python bench.py $args --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"achieved\"], {k:v[\"ms_per_step\"] for k,v in d[\"kernel_classes\"].items()})"
done
