#!/usr/bin/env python3
"""Dev tool: decode the phase stamps the dev build of attn_global8_kernel prints (WM_ATTN_DBG=1, tools/build_dev.sh)."""
import sys
import numpy as np
rows = []
for l in open(sys.argv[1]):
    if l.startswith('g8 wave'):
        rows.append([int(x) for x in l.split(':')[1].split()])
a = np.array(rows).astype(np.int64)[:, :60].reshape(8, 5, 12)
names = ["Vstart", "max", "P", "stage", "-", "bar", "PV", "QK", "bar2"]
for w in (0, 1, 4, 5):
    print("wave", w)
    for t in range(5):
        d = a[w, t]
        seq = [(0, 1), (1, 2), (2, 5), (5, 6), (6, 3), (3, 7), (7, 8)]
        lab = ["max", "P", "bar", "PV", "stage", "QK", "bar2"]
        print("  tile", t + 4, "start", int(d[0] - a[0, 0, 0]), {lab[i]: int(d[y] - d[x]) for i, (x, y) in enumerate(seq)},
              "tile total", int(a[w, t + 1, 0] - d[0]) if t < 4 else "")
