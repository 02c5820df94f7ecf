#!/bin/bash
# GPU tests + default bench (no CPU baseline) on one box
set -o pipefail
O=gpurun_out/${1:-r3c}; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/gputests.txt
rc=$?
cat $O/gputests.txt
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${2:-} > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("tiles/s", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["achieved"], d["roofline"]["frac"])
print({k:v["ms_per_step"] for k,v in d["kernel_classes"].items()})
print("parity", d.get("parity_vs_reference"))
for o in d.get("other_configs") or []: print(o)
PY
