#!/usr/bin/env python3
"""How many of a search's leaf evaluations are repeats?  (VERDICT r4, item 4 -- decided on the CPU, no GPU minutes.)

One position = one forward is the engine's and the reference's convention (players.py:85-86).  A search tree reaches some
positions by more than one move order; every such transposition is evaluated again.  This script runs the ORACLE's
800-simulation search (oracle/bz_oracle.c, the bf16-emulating net of the benchmark: random init, seed 0) from positions at
plies ~10 / 30 / 50 of cfg-3 games and counts, per search, the evaluated nodes whose (own, opp) equals an EARLIER evaluated
node of the same search -- exactly what a per-game evaluation cache inside one search could save.  Test infrastructure:
uses oracle/ only; nothing here is product code.

    python tools/measure_leaf_duplication.py [--games 8] [--sims 800] [--walk-sims 16]

The games are reached by cfg-3 self-play at --walk-sims simulations per move (the positions of a cfg-3 game do not depend
much on the search depth with a random-init net); the measured searches run at --sims."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=8)
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--walk-sims", type=int, default=16)
    ap.add_argument("--plies", default="10,30,50")
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--eval", default="net_bf16", choices=["net_bf16", "hash", "uniform"])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    from betazero_amd.net import PolicyValueNet
    from oracle import oracle as orc
    torch.manual_seed(0)
    mod = PolicyValueNet(128, 6, 64).round_to_bf16_()
    net = orc.Net(128, 6, 64, mod.flat_params())
    ev = {"net_bf16": orc.EVAL_NET_BF16, "hash": orc.EVAL_HASH, "uniform": orc.EVAL_UNIFORM}[args.eval]
    plies = [int(x) for x in args.plies.split(",")]
    jobs, lock, results = [], threading.Lock(), []

    def walk(g):
        r = orc.selfplay_game(orc.GAME_REVERSI, g, args.walk_sims, ev, 8, 1, 0, net=net)
        for want in plies:   # row k of a cfg-3 game is ply k + 2 (two opening plies) plus the passes so far: close enough
            k = min(max(want - 2, 0), len(r["own"]) - 1)
            with lock:
                jobs.append((g, want, int(r["own"][k]), int(r["opp"][k]), int(r["mover"][k])))

    def run(th, fn, items):
        ts = [threading.Thread(target=lambda it=it: fn(it)) for it in items]
        for i in range(0, len(ts), th):
            [t.start() for t in ts[i:i + th]]
            [t.join() for t in ts[i:i + th]]

    t0 = time.time()
    run(args.threads, walk, list(range(args.games)))
    print(f"[{time.time() - t0:.0f}s] {len(jobs)} positions from {args.games} games", file=sys.stderr, flush=True)

    def search(job):
        g, want, own, opp, mover = job
        o, p, term = orc.mcts_search_nodes(orc.GAME_REVERSI, own, opp, mover, args.sims, ev, net=net)
        seen, dup, evals = set(), 0, 0
        for a, b, t in zip(o.tolist(), p.tolist(), term.tolist()):
            if t:
                continue
            evals += 1
            if (a, b) in seen:
                dup += 1
            seen.add((a, b))
        with lock:
            results.append({"game": g, "ply": want, "stones": bin(own | opp).count("1"), "nodes": int(len(o)), "terminal_nodes": int(term.sum()),
                            "evaluations": evals, "repeat_evaluations": dup, "repeat_fraction": dup / max(evals, 1)})
            print(f"[{time.time() - t0:.0f}s] {results[-1]}", file=sys.stderr, flush=True)

    run(args.threads, search, jobs)
    by = {}
    for r in results:
        by.setdefault(r["ply"], []).append(r)
    summary = {"sims": args.sims, "evaluator": args.eval, "games": args.games,
               "by_ply": {str(k): {"searches": len(v), "evaluations": sum(x["evaluations"] for x in v),
                                   "repeat_evaluations": sum(x["repeat_evaluations"] for x in v),
                                   "repeat_fraction": sum(x["repeat_evaluations"] for x in v) / max(1, sum(x["evaluations"] for x in v)),
                                   "terminal_nodes": sum(x["terminal_nodes"] for x in v)} for k, v in sorted(by.items())},
               "all": {"evaluations": sum(x["evaluations"] for x in results), "repeat_evaluations": sum(x["repeat_evaluations"] for x in results)}}
    summary["all"]["repeat_fraction"] = summary["all"]["repeat_evaluations"] / max(1, summary["all"]["evaluations"])
    summary["searches"] = sorted(results, key=lambda r: (r["ply"], r["game"]))
    print(json.dumps(summary, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
