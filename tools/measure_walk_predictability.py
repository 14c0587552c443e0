#!/usr/bin/env python3
"""Would prefetching the most-visited child's edge block shorten the tree step?  (VERDICT r4, item 5 -- decided on the CPU.)

`k_tree_step` pays one dependent memory round trip per level of the walk that goes down into an expanded child.  The proposal:
keep, beside every node, a hint naming its currently most visited child, and fetch that child's edge block together with the
node's own -- when the PUCT winner IS that child, the next level's round trip is already done.  Whether that pays is a property
of the SEARCH, not of the GPU: how deep the walks are and how often the winner is the most visited child.  This script runs the
ORACLE's search (oracle/bz_oracle.c, the benchmark's random-init bf16-emulating net) from positions at plies ~10 / 30 / 50 of
cfg-3 games and counts both, with a fresh (never stale) hint -- an upper bound on what the real hint could do.  Test
infrastructure: uses oracle/ only.

    python tools/measure_walk_predictability.py [--games 8] [--sims 800]"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=8)
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--walk-sims", type=int, default=16)
    ap.add_argument("--plies", default="10,30,50")
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    from betazero_amd.net import PolicyValueNet
    from oracle import oracle as orc
    torch.manual_seed(0)
    mod = PolicyValueNet(128, 6, 64).round_to_bf16_()
    net = orc.Net(128, 6, 64, mod.flat_params())
    ev = orc.EVAL_NET_BF16
    plies = [int(x) for x in args.plies.split(",")]
    jobs, lock, results = [], threading.Lock(), []

    def walk(g):
        r = orc.selfplay_game(orc.GAME_REVERSI, g, args.walk_sims, ev, 8, 1, 0, net=net)
        for want in plies:
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

    def search(job):
        g, want, own, opp, mover = job
        st = orc.mcts_search_walkstats(orc.GAME_REVERSI, own, opp, mover, args.sims, ev, net=net)
        with lock:
            results.append(dict(st, game=g, ply=want))
            print(f"[{time.time() - t0:.0f}s] {results[-1]}", file=sys.stderr, flush=True)

    run(args.threads, search, jobs)

    def agg(rows):
        s = {k: sum(r[k] for r in rows) for k in ("sims", "levels", "fav_hits", "levels_below_root", "fav_hits_below_root", "round_trips_with_prefetch")}
        return {"searches": len(rows), **s,
                "dependent_round_trips_per_simulation": s["levels"] / s["sims"],
                "with_two_level_prefetch": s["round_trips_with_prefetch"] / s["sims"],
                "round_trips_saved_fraction": 1.0 - s["round_trips_with_prefetch"] / max(1, s["levels"]),
                "winner_is_most_visited_child": s["fav_hits"] / max(1, s["levels"]),
                "same_below_the_root": s["fav_hits_below_root"] / max(1, s["levels_below_root"])}
    by = {}
    for r in results:
        by.setdefault(r["ply"], []).append(r)
    out = {"sims": args.sims, "games": args.games, "evaluator": "net_bf16 (benchmark net, random init seed 0)",
           "what": "levels = walk steps into an expanded, non-terminal child (each is one dependent edge-block load in k_tree_step, after the "
                   "root's block, which comes with the first round trip); a hit = the PUCT winner is the node's most visited expanded child "
                   "(fresh hint: upper bound for a stored one)",
           "by_ply": {str(k): agg(v) for k, v in sorted(by.items())}, "all": agg(results), "per_search": results}
    print(json.dumps({k: v for k, v in out.items() if k != "per_search"}, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
