#!/bin/bash
# tree-step evidence on the GPU box: per-simulation durations (kernel trace) and wave counters of k_tree_step
# usage (through gpurun): bash tools/exp_tree.sh <tag>
TAG=${1:-r03}
OUT=gpurun_out/${TAG}tree
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
DB=$(find "$OUT/trace" -name "*_results.db" | head -1)
python3 tools/tree_step_trace.py "$DB" "$OUT/tree_step_by_sim.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d "$OUT/pmc_sq" -o t -- python3 bench.py $ARGS > "$OUT/bench_pmc.json" 2> "$OUT/pmc.err"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_tcc" -o t -- python3 bench.py $ARGS > "$OUT/bench_tcc.json" 2> "$OUT/tcc.err"
python3 - "$OUT" <<'P'
import glob, sqlite3, sys, os
out = sys.argv[1]
for d in ("pmc_sq", "pmc_tcc", "trace"):
    for db in glob.glob(os.path.join(out, d, "*", "*_results.db")) + glob.glob(os.path.join(out, d, "*_results.db")):
        c = sqlite3.connect(db)
        with open(os.path.join(out, d + "_summary.txt"), "w") as f:
            for r in c.execute("select name, count(*), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc limit 6"):
                print(r, file=f)
            try:
                for r in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%tree_step%' group by kernel_name, counter_name"):
                    print(r, file=f)
            except Exception as e:
                print("no counters:", e, file=f)
P
find "$OUT" -name "*_results.db" -delete
cat "$OUT/tree_step_by_sim.txt" "$OUT"/*_summary.txt
