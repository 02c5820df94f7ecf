#!/usr/bin/env python3
"""Dev tool: instruction mix of a kernel's loops from a `-save-temps` .s file.
usage: isa_loops.py file.s <mangled-name-substring> : per basic block with a back edge (label .. s_cbranch to itself) the opcode counts."""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().endswith(tuple([":"])) is False and ":" in l.split(";")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
def mix(a, b):
    c = Counter()
    for l in body[a:b]:
        t = l.strip().split()
        if t and not t[0].endswith(":") and not t[0].startswith((".", ";")):
            op = t[0]
            c[op] += 1
    return c
def summary(c):
    grp = Counter()
    for op, n in c.items():
        if op.startswith("v_mfma"): grp["mfma"] += n
        elif op.startswith("v_exp") or op.startswith("v_rcp"): grp["trans"] += n
        elif op.startswith("v_"): grp["valu"] += n
        elif op.startswith("ds_"): grp["lds"] += n
        elif op.startswith(("global_", "buffer_", "scratch_")): grp["vmem"] += n
        elif op.startswith("s_"): grp["salu"] += n
    return dict(grp)
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        c = mix(a, i)
        if sum(c.values()) < 40: continue
        print(f"loop {m.group(1)} lines {a}..{i} ({i-a} lines): {summary(c)}")
        print("   ", ", ".join(f"{k} {v}" for k, v in c.most_common(28)))
