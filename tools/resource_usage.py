#!/usr/bin/env python3
"""Dev tool: registers / scratch / occupancy of the hot kernels from `hipcc -Rpass-analysis=kernel-resource-usage` remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -o /tmp/x.so wildlifemapper_amd/csrc/wm_api.hip 2> /tmp/ra.txt; resource_usage.py /tmp/ra.txt [filter ...]"""
import re, sys
txt = open(sys.argv[1]).read()
keys = sys.argv[2:] or ["gemm16v5", "attn_global8", "attn_window", "attn_global_kernel", "gemm8"]
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0]
    if not any(k in name for k in keys):
        continue
    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    short = name.replace("wm::", "")[:120]
    print("%-120s VGPR %3d AGPR %3d scratch %4d occ %d LDS %d" % (short, g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
