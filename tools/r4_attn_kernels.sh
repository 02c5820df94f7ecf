#!/bin/bash
# per-kernel averages of the attention kernels in the model, for several builds / switches (rocprofv3 --kernel-trace --stats)
# usage: r4_attn_kernels.sh <outdir> <label=ENV=VAL[,ENV=VAL]> ...
set -o pipefail
O=gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  [ "$envs" = "-" ] && envs=""
  for kv in $(echo $envs | tr ',' ' '); do export $kv; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$label -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs > $O/$label.log 2>&1
  for kv in $(echo $envs | tr ',' ' '); do unset ${kv%%=*}; done
  python3 - $O/prof_$label $label <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        n = r["Name"]
        if "attn_" in n or "scale_q" in n:
            print(f"{sys.argv[2]:10s} {n.split('(')[0].replace('void wm::','')[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us", flush=True)
PY
done
