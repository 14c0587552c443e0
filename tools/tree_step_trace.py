#!/usr/bin/env python3
"""Per-simulation duration of k_tree_step from a rocprofv3 --kernel-trace result database (rocpd SQLite): the tree
grows with the simulation index, so duration vs index separates the step's fixed cost from its per-level cost.
usage: python tools/tree_step_trace.py <results.db> [out.txt]"""
import sqlite3
import sys

import numpy as np

db = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
c = sqlite3.connect(db)
rows = c.execute("select name, start, duration from kernels order by start").fetchall()
idx, seq = 0, []
for name, start, dur in rows:
    if "k_root_begin" in name:
        idx = 0
    elif "k_tree_step" in name:
        seq.append((idx, dur))
        idx += 1
a = np.array(seq, dtype=np.int64)
print(f"k_tree_step dispatches: {len(a)}; mean {a[:, 1].mean():.0f} ns, min {a[:, 1].min()}, max {a[:, 1].max()}", file=out)
print("sim index bucket: mean / min / max duration (ns)", file=out)
edges = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 384, 512, 640, 768, 10**9]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (a[:, 0] >= lo) & (a[:, 0] < hi)
    if m.any():
        d = a[m, 1]
        print(f"  [{lo:4d}, {min(hi, int(a[:, 0].max()) + 1):4d}): {d.mean():8.0f} {d.min():8d} {d.max():8d}   n={m.sum()}", file=out)
