#!/usr/bin/env python3
"""bench.py -- self-play games/s on MI355X (BASELINE.json metric).

Default workload = BASELINE config 3: Reversi 8x8, 4096 concurrent games per GPU,
800 MCTS simulations per move, random-init 6x128 conv policy/value net in bf16
(MFMA), tau=1 for moves < 8 + 12 fixed two-ply openings.  `--workload ttt` runs
BASELINE config 2 (65,536 TTT games, 50 sims, uniform priors, tree kernels only).

A "step" (reversi) = one move for every concurrent game: root expansion + 800 x
(select -> net -> expand/backup) + move choice / example row / env step.  The
pool runs in steady state: slot g starts pre-advanced by (g % 58) pseudo-random
plies (untimed setup) and a finished slot immediately starts its next game, so
every step completes ~B/58 games and value = games completed in the timed
region / time.  No work is skipped: every move of every game does all 800
simulations and every non-terminal leaf goes through the full net.
A "step" (ttt) = one complete iteration: all 65,536 games played to the end.

N > 1: one rank per GPU (torch.distributed, backend nccl = RCCL), games sharded
by global id (weak scaling: 4096 per GPU), no data-path collective except ONE
all-gather of the (s, pi, z) buffers at the end of the timed region.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TOWER_FLOP_PER_POS = 12 * 2 * 64 * 9 * 128 * 128      # 226.49e6: the 12 conv3x3 layers of k_tower_bf16
NET_FLOP_PER_POS = 226.86e6                           # stem + tower + heads
MFMA_PEAK_TFLOPS = 2500.0                             # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
PLIES_PER_GAME = 58                                   # searched moves per cfg-3 game (60 - 2 opening plies)


def tree_bytes(c):
    """SURVEY.md 8(d) algorithmic bytes from the kernels' exact work counters"""
    return (32 * c["n_path_nodes"] + 12 * c["n_child_scored"] + 16 * c["n_edges_backed"] + 32 * c["n_expanded"] +
            13 * c["n_child_written"] + 264 * c["n_net_leaves"] + 42 * c["n_env_steps"])


def cpu_baseline_reversi(sims, seconds_hint=20):
    """oracle (CPU restatement, kind "port") on the host cores: `cores` threads x 2 searched
    moves of cfg-3 games each (bf16-emulating net), extrapolated at 58 searched moves/game"""
    import numpy as np
    import torch
    from betazero_amd.net import PolicyValueNet
    from oracle import oracle as orc
    torch.manual_seed(0)
    mod = PolicyValueNet(128, 6, 64).round_to_bf16_()
    net = orc.Net(128, 6, 64, mod.flat_params())
    cores = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16))
    moves = 2
    done = [0] * cores

    def work(i):
        r = orc.selfplay_game(orc.GAME_REVERSI, i, sims, orc.EVAL_NET_BF16, 8, 1, 0, net=net, max_moves=moves)
        done[i] = len(r["own"])
    t0 = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.time() - t0
    return {"value": sum(done) / dt / PLIES_PER_GAME, "unit": "games/s", "cores": cores, "kind": "port",
            "sample": f"{cores} threads x {moves} searched moves ({sims} sims, {sims + 1} net evals per move) of "
                      f"cfg-3 games in {dt:.1f} s; extrapolated at {PLIES_PER_GAME} searched moves per game"}


def cpu_baseline_ttt(sims):
    from oracle import oracle as orc
    cores = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16))
    per = 4000
    t0 = time.time()

    def work(i):
        for g in range(per):
            orc.selfplay_game(orc.GAME_TTT, g, sims, orc.EVAL_UNIFORM)
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.time() - t0
    return {"value": cores * per / dt, "unit": "games/s", "cores": cores, "kind": "port",
            "sample": f"{cores} threads x {per} complete TTT games ({sims} sims/move, uniform priors) in {dt:.1f} s"}


def bench_net_only(args, rank, world, dev):
    import numpy as np
    import torch
    from betazero_amd import _lib
    from betazero_amd.net import DeviceNet, PolicyValueNet
    from betazero_amd.quant import fake_quantize_fp8_
    B = args.games or 8192
    K = args.steps if args.steps is not None else 1000
    W = args.warmup if args.warmup is not None else 50
    fp8 = args.precision != "bf16" or "--precision" not in sys.argv  # cfg 5 is the fp8 run unless bf16 is asked for
    torch.manual_seed(0)
    mod = PolicyValueNet(128, 6, 64).round_to_bf16_()
    if fp8:
        fake_quantize_fp8_(mod)
    net = DeviceNet.from_module(mod, B, dev)
    d = np.load(os.path.join(ROOT, "tests", "golden", "reversi_random_games.npz"))["rows"]
    d = d[d[:, 1] == 8]
    idx = np.arange(B) % len(d)  # positions sampled from fixture F1, tiled to the batch
    own = torch.as_tensor(d[idx, 4].copy().view(np.int64)).to(dev)
    opp = torch.as_tensor(d[idx, 5].copy().view(np.int64)).to(dev)
    L = _lib.lib()
    for _ in range(W):
        net.forward(own, opp, fp8=fp8)
    L.bz_profile_reset(); L.bz_profile_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        net.forward(own, opp, fp8=fp8)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L.bz_profile_enable(0)
    launches, timed, ms = _lib.profile_read()["tower"]
    avg_ms = ms / max(timed, 1)
    peak = 2 * MFMA_PEAK_TFLOPS if fp8 else MFMA_PEAK_TFLOPS
    ach = B * NET_FLOP_PER_POS / (avg_ms * 1e-3) / 1e12
    if rank == 0:
        print(json.dumps({"metric": "net_leaf_evals_per_s", "value": B * K * world / dt, "unit": "evals/s", "n_gpus": world,
                          "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "fp8" if fp8 else "bf16", "data": "synthetic",
                          "config": {"workload": f"reversi8x8_net_forward_batch{B}_{'fp8' if fp8 else 'bf16'}",
                                     "positions": "fixture F1 positions tiled to the batch"},
                          "roofline": {"bound": "mfma", "kernel": "f8::k_tower_fp8" if fp8 else "k_tower_bf16",
                                       "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                       "traffic": None, "launches": launches, "avg_launch_ms": avg_ms,
                                       "flop_per_launch": B * NET_FLOP_PER_POS}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="reversi", choices=["reversi", "ttt", "net"],
                    help="reversi = BASELINE cfg 3 (default, the metric), ttt = cfg 2, net = cfg 5 (net forward only)")
    ap.add_argument("--games", type=int, default=None, help="concurrent games per GPU (default: BASELINE config)")
    ap.add_argument("--sims", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp8"],
                    help="net precision; bf16 is the BASELINE config, fp8 (e4m3, cfg 5 kernel) is a supplementary line")
    ap.add_argument("--mode", default="steady", choices=["steady", "iteration"],
                    help="reversi: steady = one move per step on a staggered pool (default); iteration = a step is a\n"
                         "complete self-play iteration from the start position to the last finished game (cross-check)")
    ap.add_argument("--streams", type=int, default=2, help="independent half-batch pipelines per GPU (reversi)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("BZ_DIST_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 path on one GPU
    local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    n_gpus = world
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)

    from betazero_amd import _lib
    from betazero_amd.distributed import all_gather_example_tensors
    from betazero_amd.engine import SelfPlayEngine
    from betazero_amd.net import DeviceNet, PolicyValueNet
    _lib.require_gpu()
    L = _lib.lib()

    if args.workload == "net":  # BASELINE cfg 5: leaf-eval batch 8192, fp8 e4m3 net, MFMA-utilisation run
        return bench_net_only(args, rank, world, dev)
    reversi = args.workload == "reversi"
    B = args.games or (4096 if reversi else 65536)
    sims = args.sims or (800 if reversi else 50)
    K = args.steps if args.steps is not None else (8 if reversi else 20)
    W = args.warmup if args.warmup is not None else (1 if reversi else 2)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if reversi:
        torch.manual_seed(0)
        mod = PolicyValueNet(128, 6, 64).round_to_bf16_()
        if args.precision == "fp8":
            from betazero_amd.quant import fake_quantize_fp8_
            fake_quantize_fp8_(mod)
        net = DeviceNet.from_module(mod, B, dev)
        rounds = 2 + (W + K) // 40
        # NS independent pipelines of B/NS games, each on its own HIP stream: the tree step of one
        # overlaps the net kernel of the other and their net launches fill each other's tail wave.
        NS = max(1, args.streams)
        assert B % NS == 0
        Bs = B // NS
        streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
        engs = [SelfPlayEngine("reversi", Bs, sims, "net_" + args.precision, net, temp_moves=8, openings=1, seed=0, rounds=rounds,
                               game_id_base=rank * B + i * Bs, game_id_stride=world * B, device=dev,
                               stagger=PLIES_PER_GAME if args.mode == "steady" else 0) for i in range(NS)]
        eng = engs[0]
        for e in engs:
            e.reset_games()
        torch.cuda.synchronize()

        def step():
            if args.mode == "steady":
                for e, st in zip(engs, streams):
                    with torch.cuda.stream(st):
                        e.search()
                        e.play(True)
                return
            for e, st in zip(engs, streams):  # one whole iteration: every game from its opening to the end
                with torch.cuda.stream(st):
                    e.reset_games()
            active = True
            while active:
                for e, st in zip(engs, streams):
                    with torch.cuda.stream(st):
                        e.search()
                        e.play(False)
                torch.cuda.synchronize()
                active = any(e.status()[0] > 0 for e in engs)
    else:
        eng = SelfPlayEngine("ttt", B, sims, "uniform", game_id_base=rank * B, game_id_stride=world * B, device=dev)
        engs = [eng]

        def step():
            eng.reset_games()
            for _ in range(9):  # a TTT game has at most 9 moves; finished slots idle
                eng.search()
                eng.play(False)

    for _ in range(W):
        step()
    torch.cuda.synchronize()
    fin0 = sum(e.status()[1] for e in engs)
    for e in engs:
        e.reset_counters()
    L.bz_profile_reset()
    L.bz_profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    if world > 1:  # the one exchange step: pool this iteration's (s, pi, z)
        for e in engs:
            t = e.example_tensors()
            if backend != "nccl":
                t = {k: v.cpu() for k, v in t.items()}
            pooled = all_gather_example_tensors(t)
            del pooled
    barrier()
    dt = time.perf_counter() - t0
    L.bz_profile_enable(0)
    fin1 = sum(e.status()[1] for e in engs)
    if not reversi or args.mode == "iteration":
        fin1, fin0 = K * B, 0
    games = float(fin1 - fin0)
    if world > 1:
        t = torch.tensor([dt, games], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, games = float(tmax[0]), float(tsum[1])
    cnt = {}
    for e in engs:
        for k, v in e.counters().items():
            cnt[k] = cnt.get(k, 0) + v
    prof = _lib.profile_read()

    if rank == 0:
        out = {"metric": "selfplay_games_per_s", "value": games / dt, "unit": "games/s", "n_gpus": n_gpus, "steps": K,
               "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "data": "synthetic"}
        if reversi:
            out["dtype"] = args.precision
            out["config"] = {"workload": f"reversi8x8_{B}games_{sims}sims_convnet6x128_{args.precision}",
                             "games_per_gpu": B, "sims_per_move": sims, "c_puct": 1.5, "temp_moves": 8,
                             "openings": 12, "evaluator": "policy/value conv tower 6x128 (226.86 MFLOP per leaf), random init seed 0",
                             "step": "one move for all concurrent games (steady-state pool, staggered starts)"
                             if args.mode == "steady" else "one complete self-play iteration (all games, start to end)",
                             "pipelines": f"{NS} x {Bs} games on separate HIP streams",
                             "parallelism": f"games sharded over {n_gpus} GPU(s), one all-gather of examples"}
            launches, timed, ms = prof["tower"]
            avg_ms = ms / max(timed, 1)
            # leaves are packed before the net runs: a launch evaluates only the non-terminal leaves
            pos_per_launch = cnt["n_net_leaves"] / max(launches, 1)
            flop_per_launch = pos_per_launch * NET_FLOP_PER_POS  # stem + tower + heads are ONE kernel
            ach = flop_per_launch / (avg_ms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS if args.precision == "bf16" else 2 * MFMA_PEAK_TFLOPS  # dense fp8 = 5 PF
            union_ms, sum_ms = _lib.profile_union_ms("tower")
            conc = sum_ms / union_ms if union_ms else 1.0
            chip = cnt["n_net_leaves"] * NET_FLOP_PER_POS * (timed / max(launches, 1)) / (union_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": ("k_tower_bf16" if args.precision == "bf16" else "f8::k_tower_fp8") +
                               " (stem + 12 conv3x3 + heads, fused)",
                               "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": None, "launches": launches,
                               "avg_launch_ms": avg_ms, "positions_per_launch": pos_per_launch,
                               "flop_per_launch": flop_per_launch,
                               # launches of the NS pipelines overlap on the chip: per-launch duration is
                               # shared time.  chip-level = flops of all launches / union of their intervals
                               "concurrent_launches": conc, "achieved_chip": chip,
                               "frac_chip": chip / peak,
                               "note": "achieved/frac are per launch as specified (flops of one launch / its mean "
                                       "duration); with --streams 2 two launches share the chip, so each launch's "
                                       "duration is shared time: achieved_chip = flops of all launches / union of "
                                       "their intervals is the chip-level rate (--streams 1 makes the two coincide)"}
            try:
                assert args.precision == "bf16"
                tr = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["k_tower_bf16"]
                out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes at 4096 positions per launch (profiles/r01_pmc_traffic.json)"
            except Exception:
                pass
            tb = tree_bytes(cnt)
            tree_ms = prof["select"][2] * prof["select"][0] / max(prof["select"][1], 1) + \
                prof["expand_backup"][2] * prof["expand_backup"][0] / max(prof["expand_backup"][1], 1)
            out["roofline_tree"] = {"bound": "hbm", "kernels": "k_tree_step (expand + backup + select, 16 lanes per game)",
                                    "achieved": tb / (tree_ms * 1e-3) / 1e9 if tree_ms else None,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": tb / (tree_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if tree_ms else None,
                                    "algorithmic_bytes": tb, "kernel_ms": tree_ms}
            out["sims_per_s"] = cnt["n_sims"] * world / dt
            out["net_evals_per_s"] = cnt["n_net_leaves"] * world / dt
            out["net_tflops_e2e"] = cnt["n_net_leaves"] * world * NET_FLOP_PER_POS / dt / 1e12
            out["kernel_ms_total"] = {k: round(v[2] * v[0] / max(v[1], 1), 3) for k, v in prof.items() if v[0]}
        else:
            out["dtype"] = "u64+f32"
            out["config"] = {"workload": "ttt3x3_65536games_50sims_uniform_tree_only" if (B, sims) == (65536, 50)
                             else f"ttt3x3_{B}games_{sims}sims_uniform_tree_only", "games_per_gpu": B,
                             "sims_per_move": sims, "step": "one complete self-play iteration of all games"}
            launches, timed, ms = prof["search_fused"]
            avg_ms = ms / max(timed, 1)
            tb = tree_bytes(cnt)
            ach = tb / max(launches, 1) / (avg_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "k_search_fused<TicTacToe>", "achieved": ach,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                               "launches": launches, "avg_launch_ms": avg_ms,
                               "algorithmic_bytes_per_launch": tb / max(launches, 1)}
            out["sims_per_s"] = cnt["n_sims"] * world / dt
        out["counters"] = cnt
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_reversi(sims) if reversi else cpu_baseline_ttt(sims)  # bf16-emulating oracle
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
