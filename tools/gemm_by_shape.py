#!/usr/bin/env python3
"""Dev tool: per-shape GEMM durations from a rocprofv3 --kernel-trace run of bench.py: groups the gemm16v5 dispatches by
(kernel instance, grid size) -- qkv 3072, lin1 4096, proj / lin2 1024 workgroups at B = 16 (the latter two told apart by their
order inside a block: proj precedes lin1) -- and prints mean / median microseconds.
usage: tools/gemm_by_shape.py <rocprof output dir>"""
import collections, csv, glob, os, statistics, sys
csv.field_size_limit(1 << 30)
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            if "gemm16v5" in r["Kernel_Name"] or "layernorm" in r["Kernel_Name"] or "ln_stats" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
groups = collections.defaultdict(list)
prev1024 = collections.Counter()                     # per instance: proj and lin2 alternate inside the blocks' residual instance
for t, name, grid, dur in rows:
    # template tail <..., DBG, FOLDP, FOLDC, SPLIT> (round 4; rounds 2-3 had no SPLIT)
    inst = ("SPLIT" if "false, true, false, true>" in name else "FOLDP" if "false, true, false, false>" in name else
            "FOLDC" if "false, false, true, false>" in name else ("LN" if "layernorm" in name or "ln_stats" in name else "plain"))
    key = (inst, grid)
    if grid == 1024 and "gemm16v5" in name and ", 320," in name:
        prev1024[inst] += 1
        key = (inst, grid, "proj (or proj_back / patch embed)" if prev1024[inst] % 2 == 1 else "lin2")
    groups[key].append(dur / 1e3)
for k in sorted(groups, key=str):
    v = groups[k]
    print(f"{str(k):50s} n={len(v):5d} mean {statistics.mean(v):8.1f} us  median {statistics.median(v):8.1f} us")
