#!/bin/bash
# per-grid averages of the decoder-side kernels (gemm32 / gemm32x3 / mha32 / small LayerNorms) from a kernel trace of the default bench
# usage: r4_small_kernels.sh <outdir> <label=ENV=VAL[,ENV=VAL]> ...
set -o pipefail
O=gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  [ "$envs" = "-" ] && envs=""
  for kv in $(echo $envs | tr ',' ' '); do export $kv; done
  rocprofv3 --kernel-trace --output-format csv -d $O/prof_$label -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs > $O/$label.log 2>&1
  for kv in $(echo $envs | tr ',' ' '); do unset ${kv%%=*}; done
  python3 - $O/prof_$label $label <<'PY'
import csv, glob, os, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(k in n for k in ("gemm32", "mha32", "add_bcast", "split_w32")) or "layernorm_kernel" in n:
        d[(n.split("(")[0].replace("void wm::", "").replace("wm::", "")[:44], r.get("Grid_Size") or r.get("Grid_Size_X"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sys.argv[2]:6s} {k[0]:44s} grid {k[1]:>8s} n={len(v):4d} avg {sum(v)/len(v):8.1f} us  sum {sum(v)/1e3:8.2f} ms", flush=True)
    tot += sum(v)
print(f"{sys.argv[2]:6s} total {tot/1e3:.2f} ms over the traced process")
PY
  rm -rf $O/prof_$label
done
