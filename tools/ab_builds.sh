#!/bin/bash
# A/B of library builds / switches on ONE box, alternating runs of the default bench (no CPU baseline, no other configs).
# WM_AB_BENCH_ARGS: extra bench.py arguments (e.g. "--precision fp8")
# usage: tools/ab_builds.sh <outdir> <label=ENV=VAL[,ENV=VAL]> ...   ("-" for no env), 2 rounds
set -o pipefail
O=gpurun_out/$1; shift; mkdir -p $O
for round in 1 2; do
  for spec in "$@"; do
    label=${spec%%=*}; envs=${spec#*=}
    [ "$envs" = "-" ] && envs=""
    env $(echo $envs | tr ',' ' ') python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs $WM_AB_BENCH_ARGS > $O/${label}_$round.json 2>> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
    python - $O/${label}_$round.json $label $round <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:14s} round {sys.argv[3]}: {d['value']:7.2f} tiles/s  gemm {d['roofline']['achieved']:7.1f} TF", {k:v["ms_per_step"] for k,v in d["kernel_classes"].items()}, flush=True)
PY
  done
done
