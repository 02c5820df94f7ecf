#!/bin/bash
# round 4, VERDICT item 8: tile-order group size (WM_GEMM_GROUP_M, dev build) in the model: tiles/s and FETCH_SIZE per GEMM launch.
# group_m x tilesN co-resident tiles share an XCD's L2: 8 (default) = 8(M) x 4(N) per 32 workgroups, 4 = 4 x 8, 2 = 2 x 16.
set -o pipefail
O=gpurun_out/${1:-r4gm}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WM_HIP_LIB=build/ab/libwm_dev.so
for round in 1 2; do
  for g in 8 4 2; do
    WM_GEMM_GROUP_M=$g python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench_g${g}_$round.json 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
    python - $O/bench_g${g}_$round.json $g $round <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"group_m {sys.argv[2]} round {sys.argv[3]}: {d['value']:7.2f} tiles/s  gemm class {d['kernel_classes']['gemm16']['ms_per_step']:7.3f} ms/step", flush=True)
PY
  done
done
for g in 8 4 2; do
  WM_GEMM_GROUP_M=$g rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_g$g -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs > $O/pmc_g$g.log 2>&1
  python3 - $O/pmc_g$g $g <<'PY'
import collections, csv, glob, os, sys
csv.field_size_limit(1 << 30)
tot, n = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        if r["Counter_Name"] != "FETCH_SIZE" or "gemm16v5" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void wm::", "").replace("wm::", "")
        tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in sorted(tot):
    print(f"group_m {sys.argv[2]}: {k:60s} launches {n[k]:4d}  FETCH_SIZE x 2 = {2 * tot[k] / n[k] * 1024 / 1e6:8.1f} MB per launch", flush=True)
PY
done
