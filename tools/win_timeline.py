#!/usr/bin/env python3
"""Dev tool: decode the stamps the dev build of attn_window_kernel prints (WM_ATTN_DBG=1, tools/build_dev.sh)."""
import sys
import numpy as np
rows = []
for l in open(sys.argv[1]):
    if l.startswith('win wave'):
        rows.append([int(x) for x in l.split(':')[1].split()])
a = np.array(rows, dtype=np.int64).reshape(7, 3, 16)
lab = ["prefetch", "relUV", "k0", "k1", "k2", "k3", "store", "bar1", "commit", "bar2"]
for w in range(7):
    print("wave", w)
    for it in range(3):
        d = a[w, it]
        print("  item", it + 1, "start", int(d[0] - a[0, 0, 0]), {lab[i]: int(d[i + 1] - d[i]) for i in range(10)}, "total", int(d[10] - d[0]))
