#!/usr/bin/env python3
"""Dev tool: per-kernel averages of every counter in a rocprofv3 -i <pmc file> output directory.
usage: tools/pmc_kernel.py <dir> <kernel-name substring> [...]"""
import collections, csv, glob, os, sys
csv.field_size_limit(1 << 30)
root, keys = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            for k in keys:
                if k in r["Kernel_Name"]:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in keys:
    print("==", k)
    for c, v in sorted(acc[k].items()):
        print(f"  {c:32s} n={len(v):4d} avg={sum(v) / len(v):16.1f}")
