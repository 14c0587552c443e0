#!/usr/bin/env python3
"""Turn the rocprofv3 result databases of tools/profile_round.sh (gpurun_out/<tag>prof/*/..._results.db, rocpd
SQLite) into the small files that are committed under profiles/:
  <tag>_<name>_kernel_stats.csv   per kernel: calls, total / average / min / max duration (ns), percentage
                                  (the same columns as rocprofv3's own --stats CSV)
  <tag>_<name>_pmc.csv            per (kernel, counter): dispatches, mean / min / max counter value per dispatch,
                                  mean dispatch duration (ns)
usage: python tools/summarize_prof.py r02 [--out DIR] [--src DIR]   (default DIR = profiles/; profile_round.sh summarises on the GPU box
into gpurun_out/<tag>prof/summary/ and deletes the databases, which are too big to travel back)"""
import csv
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = sys.argv[sys.argv.index("--src") + 1] if "--src" in sys.argv else os.path.join(ROOT, "gpurun_out", f"{tag}prof")
dst = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")


for db in sorted(glob.glob(os.path.join(src, "*", "*_results.db"))):
    name = os.path.basename(os.path.dirname(db))
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                     "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r[0]), r[1], r[2], f"{r[3]:.1f}", f"{100.0 * r[2] / tot:.2f}", r[4], r[5]])
    n_pmc = c.execute("select count(*) from counters_collection").fetchone()[0]
    if n_pmc:
        prow = c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                         "from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        with open(os.path.join(dst, f"{tag}_{name}_pmc.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel", "Counter", "Dispatches", "MeanValue", "MinValue", "MaxValue", "MeanDurationNs"])
            for r in prow:
                w.writerow([short(r[0]), r[1], r[2], f"{r[3]:.1f}", f"{r[4]:.1f}", f"{r[5]:.1f}", f"{r[6]:.1f}"])
    print(name, "kernels:", len(rows), "pmc rows:", n_pmc)
