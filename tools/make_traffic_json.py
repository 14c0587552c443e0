#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the summarised counter passes of tools/profile_round.sh:
HBM bytes per launch of the tower from FETCH_SIZE / WRITE_SIZE (separate --pmc passes; gfx950: FETCH_SIZE counts 128-byte
requests at 64 bytes -- MI355X_MICROARCH.md, HBM section -- so hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024), at the
bench's own batch (pmc_bench_*: the default two-pipeline command) and at 4096 positions (pmc_tower_*: tools/bench_net.py).
usage: python tools/make_traffic_json.py r03 [summary-dir]   (default summary dir: profiles/)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")


def mean(name, counter, kernel):
    # pass names carry their suffix at the end: pmc_ttt_fetch_gw4 <- mean("pmc_ttt_fetch_gw4", ...)
    path = os.path.join(src, f"{tag}_{name}_pmc.csv")
    if not os.path.exists(path):
        return None
    for r in csv.DictReader(open(path)):
        if r["Counter"] == counter and kernel in r["Kernel"]:
            return float(r["MeanValue"]), int(r["Dispatches"])
    return None


out = {"source": f"tools/profile_round.sh {tag}: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); "
                 "values in KB per launch, mean over launches",
       "correction": "MI355X_MICROARCH.md HBM section: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts "
                     "128-B requests at 64 B: exact for wide coalesced streaming reads; for the tree step's dependent 16-byte-per-lane "
                     "loads the request size is not known, so its entry carries both readings)"}
BENCH_CMD = "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary (two pipelines of 2048 games)"
# key, pass name, kernel-name substring, shape (what bench.py must be running for the figure to apply), command
for key, name, kern, shape, cmd in (
        ("k_tower_bf16", "pmc_bench", "k_tower_bf16", None, BENCH_CMD),
        ("k_tower_bf16@4096", "pmc_tower", "k_tower_bf16", {"positions_per_launch": 4096}, "python3 tools/bench_net.py 4096 60"),
        ("k_tree_step", "pmc_bench", "k_tree_step", {"games_per_launch": 2048, "sims": 800}, BENCH_CMD),
        ("k_search_fused_ttt", "pmc_ttt", "k_search_fused_ttt", {"games": 65536, "sims": 50, "ttt_lanes": 4},
         "python3 bench.py --workload ttt --ttt-lanes 4 --steps 2 --warmup 1 --no-cpu-baseline"),
        ("k_tower_fp8@8192", "pmc_fp8", "k_tower_fp8", {"positions_per_launch": 8192}, "python3 tools/bench_net.py 8192 60 fp8"),
        ("k_tower_fp8@selfplay", "pmc_cfg5sp", "k_tower_fp8", None,
         "python3 bench.py --precision fp8 --games 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary (two pipelines of 4096 games)"),
        ("k_reversi_step", "pmc_env", "k_reversi_step", {"games": 1 << 26}, "python3 tools/bench_env.py")):
    suffix = "_gw4" if name == "pmc_ttt" else ""
    f, w = mean(f"{name}_fetch{suffix}", "FETCH_SIZE", kern), mean(f"{name}_write{suffix}", "WRITE_SIZE", kern)
    if not f or not w:
        continue
    ent = {"fetch_kb": f[0], "write_kb": w[0], "hbm_bytes_per_launch": int((2 * f[0] + w[0]) * 1024),
           "hbm_bytes_per_launch_uncorrected": int((f[0] + w[0]) * 1024), "launches": f[1], "command": cmd}
    if shape:
        ent.update(shape)
    if key in ("k_tower_bf16", "k_tower_fp8@selfplay"):
        try:  # positions per launch of the very run the counters come from
            d = json.load(open(os.path.join(src, f"{tag}_{name}_fetch.json")))
            ent["positions_per_launch"] = d["roofline"]["positions_per_launch"]
        except Exception:
            pass
    out[key] = ent
# the training step: the sum over its ten kernels (tools/prof_train_pmc.sh: the kernels launched eagerly, counters per kernel)
for key, pre, shape in (("train_step@128x6x1024", "train128", {"channels": 128, "blocks": 6, "batch": 1024}),
                        ("train_step@64x4x1024", "train64", {"channels": 64, "blocks": 4, "batch": 1024})):
    fp, wp = os.path.join(src, f"{tag}_{pre}_fetch_pmc.csv"), os.path.join(src, f"{tag}_{pre}_write_pmc.csv")
    if not (os.path.exists(fp) and os.path.exists(wp)):
        continue
    per = {}
    for path, col in ((fp, "fetch_kb"), (wp, "write_kb")):
        for r in csv.DictReader(open(path)):
            if ("k_train_" in r["Kernel"] or "k_pack_weights" in r["Kernel"]) and r["Counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
                per.setdefault(r["Kernel"].split("(")[0], {})[col] = float(r["MeanValue"])
    if not per or any(len(v) != 2 for v in per.values()):
        continue
    tot = sum(2 * v["fetch_kb"] + v["write_kb"] for v in per.values()) * 1024
    out[key] = dict(shape, hbm_bytes_per_launch=int(tot), kernels={k: int((2 * v["fetch_kb"] + v["write_kb"]) * 1024) for k, v in per.items()},
                    note="one launch = one step = the sum over its ten kernels", command=f"python3 tools/prof_train_kernels.py {shape['channels']} {shape['blocks']} {shape['batch']}")
path = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
json.dump(out, open(path, "w"), indent=1)
print(open(path).read())
