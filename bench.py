#!/usr/bin/env python3
"""bench.py -- self-play games/s on MI355X (BASELINE.json metric).

Default workload = BASELINE config 3: Reversi 8x8, 4096 concurrent games per GPU,
800 MCTS simulations per move, random-init 6x128 conv policy/value net in bf16
(MFMA), tau=1 for moves < 8 + 12 fixed two-ply openings.  `--workload ttt` runs
BASELINE config 2 (65,536 TTT games, 50 sims, uniform priors, tree kernels only),
`--workload net` config 5 (net forward only, batch 8192, fp8), `--workload env` the
batched board-env step kernel alone.  The default run also measures configs 2 and 5
and the env step briefly after the headline measurement and attaches them as
`secondary` (driver-timed evidence for their rooflines; not part of `value`).

A "step" (reversi) = one move for every concurrent game: root expansion + 800 x
(select -> net -> expand/backup) + move choice / example row / env step.  The
pool runs in steady state: slot g starts pre-advanced by (g % 58) pseudo-random
plies (untimed setup) and a finished slot immediately starts its next game, so
every step completes ~B/58 games and value = games completed in the timed
region / time.  No work is skipped: every move of every game does all 800
simulations and every distinct non-terminal position of a search goes through the
full net.  (A position that a search reaches a second time by another move order --
6-12 % of a search's leaves -- or that the slot's PREVIOUS search already evaluated --
the played move's old subtree, a quarter of the previous search -- shares that
evaluation: the engine's evaluation cache, BZ_ENGINE_EVAL_CACHE(_CARRY).  The tree is
still built from scratch for every move.  Results are bit for bit those
without it, the tree is the same tree; `roofline` counts only the rows the net
really computed, the line says how many evaluations were shared, and
`secondary.cfg3_no_eval_cache` / `--eval-cache off` run the same workload with every
leaf through the net, `secondary.cfg3_eval_cache_in_search_only` / `--eval-cache search`
with repeats inside one search shared only.)
A "step" (ttt) = one complete iteration: all 65,536 games played to the end.

N > 1: one rank per GPU (torch.distributed, backend nccl = RCCL), games sharded
by global id (weak scaling: 4096 per GPU), no data-path collective except ONE
all-gather of the example blocks at the end of the timed region.
`python bench.py --gpus N` starts the N ranks itself (torch.distributed.run
children, before this process touches a GPU); under an external torchrun
(WORLD_SIZE set) it is a rank and `--gpus` must equal the world size.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TOWER_FLOP_PER_POS = 12 * 2 * 64 * 9 * 128 * 128      # 226.49e6: the 12 conv3x3 layers of k_tower_bf16
NET_FLOP_PER_POS = 226.86e6                           # stem + tower + heads
MFMA_PEAK_TFLOPS = 2500.0                             # dense bf16, MI355X_MICROARCH.md (fp8: 2x)
HBM_PEAK_GBS = 8000.0
PLIES_PER_GAME = 58                                   # searched moves per cfg-3 game (60 - 2 opening plies)
WHOLE_GAME_SIMS = 32                                  # cpu_baseline.whole_game: complete games on the C port at this many sims/move
PMC_TRAFFIC = ("profiles/r05_pmc_traffic.json", "profiles/r04_pmc_traffic.json", "profiles/r03_pmc_traffic.json", "profiles/r02_pmc_traffic.json")


def tree_bytes(c):
    """SURVEY.md 8(d) algorithmic bytes from the kernels' exact work counters"""
    return (32 * c["n_path_nodes"] + 12 * c["n_child_scored"] + 16 * c["n_edges_backed"] + 32 * c["n_expanded"] +
            13 * c["n_child_written"] + 264 * c["n_net_leaves"] + 42 * c["n_env_steps"])



def pmc_traffic(key, **shape):
    """(hbm bytes per launch, source note) of a kernel from the committed counter passes (tools/profile_round.sh ->
    tools/make_traffic_json.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 fetch correction
    applied) -- PMC counters cannot run inside a timed process, so the line carries the figure measured on the same
    command, and only when the entry's recorded shape equals this run's (`shape`); (None, None) otherwise."""
    for path in PMC_TRAFFIC:
        try:
            ent = json.load(open(os.path.join(ROOT, path)))[key]
        except Exception:
            continue
        if all(ent.get(k) == v for k, v in shape.items()):
            return ent["hbm_bytes_per_launch"], f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `{ent.get('command', '?')}` ({path})"
    return None, None


_T0 = time.time()


def note(msg):
    """progress line on stderr (stdout carries only the JSON line)"""
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota (a
    container on a 256-thread host may be given 16: one thread per USABLE core, not per visible one)"""
    n = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0))))
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except Exception:
            pass
    return n


# ------------------------------------------------------------------ N ranks from one command
def launch_ranks(n, argv):
    """parent of `python bench.py --gpus N`: start N fresh rank processes and relay their status.
    Nothing here touches a GPU (not even `import torch`), and nothing is exec'ed from a process
    that has: the ranks are children."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env, stdout=_JSON_FD)  # (the parent's own descriptor 1 is stderr by now: claim_stdout)


def rehearsal(args, rank, world):
    """CPU-only rehearsal of the N>1 plumbing (no GPU in this process, BZ_BENCH_REHEARSAL=1): process group,
    barrier + max-over-ranks timing and the single all-gather of packed example blocks (the engine's exact layout)
    into buffers allocated before the timed region.  Measures nothing about the hot path and says so."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from betazero_amd import distributed as bd
    from betazero_amd.engine import Examples, build_packed_block, packed_block_header, unpack_packed_block
    B, na, T = args.games or 256, 65, 64
    g = np.random.default_rng(rank)
    n_games = B // 6 + rank  # finished games differ by rank: the header's count travels with the block
    lens = g.integers(40, T - 4, n_games)
    n = int(lens.sum())
    ex = Examples(own=g.integers(0, 2**62, n, dtype=np.int64).view(np.uint64), opp=np.zeros(n, np.uint64),
                  pi=g.random((n, na), dtype=np.float32), z=np.zeros(n, np.int8), mover=np.ones(n, np.int8),
                  act=np.zeros(n, np.uint8), game=np.concatenate([np.full(k, rank * B + i, np.int64) for i, k in enumerate(lens)]),
                  ply=np.concatenate([np.arange(k, dtype=np.int32) for k in lens]), size=8)
    cap = (B // 6 + world) * T
    buf = bd.GatherBuffers(na, cap, world, "cpu")
    buf.send.copy_(build_packed_block(ex, cap, "reversi"))
    calls, allocs = [], []
    real, real_empty, real_cat = dist.all_gather_into_tensor, torch.empty, torch.cat
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    K = args.steps if args.steps is not None else 2
    dist.barrier()
    t0 = time.perf_counter()
    torch.empty = lambda *a, **k: (allocs.append("empty"), real_empty(*a, **k))[1]
    torch.cat = lambda *a, **k: (allocs.append("cat"), real_cat(*a, **k))[1]
    for _ in range(K):
        time.sleep(0.01)  # stands in for a step
    gathered = bd.all_gather_packed(buf.send, buf.out)
    torch.empty, torch.cat = real_empty, real_cat
    dist.barrier()
    dt = time.perf_counter() - t0
    dist.all_gather_into_tensor = real
    heads = [packed_block_header(gathered[r]) for r in range(world)]
    rows = sum(len(unpack_packed_block(gathered[r])) for r in range(world))
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ranks = [None] * world
    dist.all_gather_object(ranks, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device_index": None,
                                   "device_name": "cpu (rehearsal)", "device_uuid": "", "games_finished": float(n_games), "seconds": dt,
                                   "pid": os.getpid()})
    if rank == 0:
        emit_json({"metric": "selfplay_games_per_s", "value": None, "unit": "games/s", "n_gpus": dist.get_world_size(),
                          "steps": K, "warmup": args.warmup or 0, "ms_per_step": float(t[0]) / K * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                          "data": "synthetic", "rehearsal": True,
                          "config": {"workload": "CPU rehearsal of the multi-rank plumbing only (no GPU in this "
                                                 "process): launcher, process group, one all-gather, timing"},
                          "ranks": {"backend": "gloo", "world_size": world, "distinct_devices": 0, "per_rank": ranks},
                          "collectives": len(calls), "allocations_in_timed_region": allocs, "block_bytes_per_rank": int(buf.nbytes),
                          "gathered_bytes": int(gathered.numel()), "pooled_rows": int(rows),
                          "pooled_games": int(sum(h["n_games"] for h in heads))})


def usable_cores():
    """host_cores() verified by a 1-vs-n thread probe of the oracle (a CPU quota this process cannot
    read would otherwise oversubscribe the baseline run many times over): threads beyond what the
    probe shows to run in parallel are not used"""
    import numpy as np
    from oracle import oracle as orc
    n = host_cores()
    if n == 1:
        return 1
    rng = np.random.default_rng(0)
    net = orc.Net(64, 2, 64, rng.standard_normal(orc.lib().orc_net_param_count(64, 2, 64)).astype(np.float32) * 0.05)
    own = np.array([0x0000000810000000] * 16, np.uint64)
    opp = np.array([0x0000001008000000] * 16, np.uint64)

    def spin(k):
        for _ in range(k):
            net.forward(own, opp)  # a few tens of ms inside the C library (GIL released)
    spin(1)
    t0 = time.time(); spin(4); t1 = time.time() - t0
    tn = None
    for _ in range(2):  # best of two: a noisy neighbour must not halve the estimate
        th = [threading.Thread(target=spin, args=(4,)) for _ in range(n)]
        t0 = time.time(); [t.start() for t in th]; [t.join() for t in th]
        tn = min(tn or 1e9, time.time() - t0)
    return max(1, min(n, int(n * t1 / tn + 0.5)))


# ------------------------------------------------------------------ CPU baselines (rank 0, N = 1)
def _py_twin():
    """oracle/py_twin.py as a module (loaded by path: `oracle` stays the package it is on sys.path)"""
    import importlib.util
    if "bz_py_twin" not in sys.modules:
        spec = importlib.util.spec_from_file_location("bz_py_twin", os.path.join(ROOT, "oracle", "py_twin.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["bz_py_twin"] = mod
        spec.loader.exec_module(mod)
    return sys.modules["bz_py_twin"]



def cpu_baseline_reversi(sims, budget_s=12.0):
    """(i) "port": the C oracle on ALL host cores, one game per thread, 2 searched moves each of cfg-3
    games with the bf16-emulating net, extrapolated at 58 searched moves per game.
    (ii) "python_loop": the build-authored Python MCTS twin (oracle/py_twin.py) over betazero_amd's
    API-compatible ReversiBoard with a batch-1 torch CPU forward of the same net per leaf (the calling
    convention of the reference's AIPlayer, players.py:84-98), ONE core, time-bounded shares of three
    800-simulation searches from an early, a middle and a late position of a fixture game (the turn loop
    being priced: reversi_terminal.py:16-38), extrapolated from their mean.  The reference has no MCTS loop
    to time."""
    import numpy as np
    import torch
    from betazero_amd.net import PolicyValueNet, bits_to_planes
    from oracle import oracle as orc
    torch.manual_seed(0)
    mod = PolicyValueNet(128, 6, 64).round_to_bf16_()
    net = orc.Net(128, 6, 64, mod.flat_params())
    cores = usable_cores()
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
    out = {"value": sum(done) / dt / PLIES_PER_GAME, "unit": "games/s", "cores": cores, "kind": "port",
           "sample": f"{cores} threads x {moves} searched moves ({sims} sims, {sims + 1} net evals per move) of "
                     f"cfg-3 games in {dt:.1f} s; extrapolated at {PLIES_PER_GAME} searched moves per game"}
    # (i-b) the extrapolation checked against WHOLE games: every thread plays one complete cfg-3 game (its own game id:
    # openings, tau = 1 sampling and therefore move counts and branching differ) at a reduced, stated number of
    # simulations, and the same game's first two searched moves again on their own -- the ratio says what "2 moves x 29"
    # over- or under-states against a game's real profile (late searches reach terminal leaves, which cost no net call)
    ws = WHOLE_GAME_SIMS
    whole = [None] * cores

    def work_whole(i):
        t0 = time.time()
        r2 = orc.selfplay_game(orc.GAME_REVERSI, i, ws, orc.EVAL_NET_BF16, 8, 1, 0, net=net, max_moves=moves)
        t2 = time.time() - t0
        t0 = time.time()
        r = orc.selfplay_game(orc.GAME_REVERSI, i, ws, orc.EVAL_NET_BF16, 8, 1, 0, net=net)
        whole[i] = (len(r["own"]), time.time() - t0, len(r2["own"]), t2, r["counters"]["n_net_leaves"], r["passes"])
    t0 = time.time()
    th = [threading.Thread(target=work_whole, args=(i,)) for i in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    dtw = time.time() - t0
    mv = [w[0] for w in whole]
    extrap = [w[3] / max(w[2], 1) * w[0] for w in whole]          # per game: its 2-move sample scaled to its own move count
    extrap58 = [w[3] / max(w[2], 1) * PLIES_PER_GAME for w in whole]  # ... and at the fixed 58 the headline extrapolation uses
    ratio = sum(w[1] for w in whole) / sum(extrap)
    ratio58 = sum(w[1] for w in whole) / sum(extrap58)
    out["whole_game"] = {"value": cores / (sum(w[1] for w in whole) / cores), "unit": "games/s", "cores": cores, "kind": "port",
                         "sims": ws, "games": cores, "searched_moves_per_game": {"mean": sum(mv) / len(mv), "min": min(mv), "max": max(mv)},
                         "net_evals_per_game_mean": sum(w[4] for w in whole) / cores, "passes_total": sum(w[5] for w in whole),
                         "whole_game_s_over_two_move_sample_scaled_to_its_moves": ratio,
                         "whole_game_s_over_two_move_sample_x_58": ratio58,
                         "sample": f"{cores} threads x 1 COMPLETE cfg-3 game each at {ws} sims/move (+ the same game's first {moves} "
                                   f"searched moves again on their own) in {dtw:.1f} s"}
    out["value_corrected_by_whole_game_profile"] = out["value"] / ratio58
    out["sample"] += (f"; whole games at {ws} sims take {ratio58:.2f} x what their first {moves} moves x {PLIES_PER_GAME} / {moves} predict "
                      f"(cpu_baseline.whole_game), so the extrapolated figure is a LOWER bound on the port's speed by about that factor")
    # the same port on ONE thread (1 thread and all cores)
    t0 = time.time()
    r1 = orc.selfplay_game(orc.GAME_REVERSI, 0, sims, orc.EVAL_NET_BF16, 8, 1, 0, net=net, max_moves=1)
    dt1 = time.time() - t0
    out["one_thread"] = {"value": len(r1["own"]) / dt1 / PLIES_PER_GAME, "unit": "games/s", "cores": 1, "kind": "port",
                         "sample": f"1 thread x 1 searched move ({sims} sims) of a cfg-3 game in {dt1:.1f} s; extrapolated at "
                                   f"{PLIES_PER_GAME} searched moves per game"}
    # (ii) the Python loop, one core
    try:
        py_twin = _py_twin()
        import betazero_amd as bz
        nthr = torch.get_num_threads()
        torch.set_num_threads(1)

        class NetTwin(py_twin.Twin):
            def evaluate(self, b, p):
                own, opp = self.bits(b, p)
                with torch.no_grad():
                    lg, v = mod(bits_to_planes(np.array([own], np.uint64), np.array([opp], np.uint64)))
                return [np.float32(x) for x in lg[0].tolist()], np.float32(float(v[0]))
        tw = NetTwin("reversi", "net", boards=(bz.ReversiBoard, bz.TicTacToeBoard))
        # SURVEY 8(d)(ii) asks for whole games; a whole 800-simulation game is ~20 minutes of this loop, so the budget goes to
        # searches from three positions of ONE fixture-F1 game at plies ~10 / 30 / 50 (early, middle and late game: branching
        # and the share of terminal leaves differ -- the start position alone, branching 4, is the cheapest search of a game),
        # a third of the budget each; the game's cost is extrapolated from the MEAN time per simulation of the three
        rows = np.load(os.path.join(ROOT, "tests", "golden", "reversi_random_games.npz"))["rows"]
        game = rows[(rows[:, 0] == rows[rows[:, 1] == 8][0, 0])]
        samples = []
        for want in (10, 30, 50):
            cand = game[(game[:, 2] >= want) & (game[:, 6] != 0) & (game[:, 9] == 0)]   # a ply with a legal move, game not over
            if not len(cand):
                continue
            r = cand[0]
            board, mover = bz.ReversiBoard.from_bits(int(r[4]), int(r[5]), 8), int(r[3]) - 1
            t0 = time.time()
            root = tw.new_node(board, mover)
            tw.expand(root)
            n = 0
            while n < sims and time.time() - t0 < budget_s / 3:
                tw.simulate(root)
                n += 1
            dtp = time.time() - t0
            samples.append({"ply": int(r[2]), "legal_moves": bin(int(r[6])).count("1"), "simulations": n, "seconds": round(dtp, 2),
                            "sims_per_s": n / dtp})
        torch.set_num_threads(nthr)
        per_sim = sum(x["seconds"] / max(x["simulations"], 1) for x in samples) / len(samples)
        per_move = per_sim * sims
        out["python_loop"] = {"value": 1.0 / (per_move * PLIES_PER_GAME), "unit": "games/s", "cores": 1,
                              "kind": "build-authored Python MCTS over betazero_amd.ReversiBoard + batch-1 torch CPU net",
                              "positions": samples,
                              "sample": f"searches from {len(samples)} positions of one fixture-F1 game (plies "
                                        f"{', '.join(str(x['ply']) for x in samples)}), {budget_s / 3:.0f} s each: "
                                        f"{', '.join(format(x['sims_per_s'], '.1f') for x in samples)} simulations/s; extrapolated from their "
                                        f"mean time per simulation to {sims} sims x {PLIES_PER_GAME} searched moves"}
    except Exception as e:  # the baseline must never take the bench line down
        out["python_loop"] = {"value": None, "error": repr(e)}
    return out


def cfg1_python_loop(n_games=100, sims=25, temp_moves=2):
    """BASELINE cfg 1 (plumbing, CPU): the build-authored Python MCTS (oracle/py_twin.py -- the reference has no MCTS,
    SURVEY 0 F2) over betazero_amd's API-compatible TicTacToeBoard, 25 sims/move, uniform priors, seed 0, 100 games,
    ONE core; every game must be legal and finish.  The first `temp_moves` moves of a game are sampled ~ N (tau = 1,
    counter RNG keyed by the game id) so that the 100 games are 100 different games, not one game 100 times.  The trajectory loop has the semantics of the reference's
    TicTacToeHeadless.play (src/tic_tac_toe/tic_tac_toe.py:13-34): position recorded before every move, winner from
    is_game_over."""
    py_twin = _py_twin()
    import betazero_amd as bz
    tw = py_twin.Twin("ttt", "uniform", boards=(bz.ReversiBoard, bz.TicTacToeBoard))
    t0 = time.time()
    plies, res, legal, distinct = 0, {1: 0, -1: 0, 0: 0}, True, set()
    for g in range(n_games):
        ex, w, _ = tw.selfplay(g, sims, temp_moves, 0, 0)
        plies += len(ex)
        res[w] += 1
        distinct.add(tuple(e[4] for e in ex))
        # replay through the board API: every recorded action legal, the game really over, the winner as recorded
        b, p = bz.TicTacToeBoard(), 1
        for (_own, _opp, _pi, mover, a) in ex:
            legal &= mover == p and bool(b.is_valid_move(a // 3, a % 3))
            b = b.make_move(a // 3, a % 3, p)
            p = -p
        over, win = b.is_game_over()
        legal &= bool(over) and (win or 0) == w
    dt = time.time() - t0
    return {"metric": "selfplay_games_per_s", "value": n_games / dt, "unit": "games/s", "cores": 1,
            "config": {"workload": f"ttt3x3_{n_games}games_{sims}sims_python_loop_cpu",
                       "loop": "build-authored Python MCTS (oracle/py_twin.py) over betazero_amd.TicTacToeBoard; "
                               "the reference has no MCTS loop to time"},
            "games": n_games, "temp_moves": temp_moves, "distinct_move_sequences": len(distinct),
            "plies_per_game": plies / n_games, "results_x_o_draw": [res[1], res[-1], res[0]],
            "all_games_legal_and_finished": bool(legal), "seconds": dt}


def cpu_baseline_ttt(sims):
    from oracle import oracle as orc
    cores = usable_cores()
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


# ------------------------------------------------------------------ the three workloads
class Ctx:
    pass


def run_net(ctx, B, K, W, fp8):
    """BASELINE cfg 5: net forward only on a batch of fixture-F1 positions"""
    import numpy as np
    import torch
    from betazero_amd import _lib
    from betazero_amd.net import DeviceNet, PolicyValueNet
    from betazero_amd.quant import fake_quantize_fp8_
    torch.manual_seed(0)
    mod = PolicyValueNet(128, 6, 64).round_to_bf16_()
    if fp8:
        fake_quantize_fp8_(mod)
    net = DeviceNet.from_module(mod, B, ctx.dev)
    d = np.load(os.path.join(ROOT, "tests", "golden", "reversi_random_games.npz"))["rows"]
    d = d[d[:, 1] == 8]
    idx = np.arange(B) % len(d)  # positions sampled from fixture F1, tiled to the batch
    own = torch.as_tensor(d[idx, 4].copy().view(np.int64)).to(ctx.dev)
    opp = torch.as_tensor(d[idx, 5].copy().view(np.int64)).to(ctx.dev)
    L = _lib.lib()
    for _ in range(W):
        net.forward(own, opp, fp8=fp8)
    L.bz_profile_reset()
    L.bz_profile_reserve(_lib.PROF_SLOTS.index("tower"), K + 8)
    L.bz_profile_enable(1)
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        net.forward(own, opp, fp8=fp8)
    ctx.barrier()
    dt = time.perf_counter() - t0
    L.bz_profile_enable(0)
    launches, timed, ms = _lib.profile_read()["tower"]
    assert launches == timed == K, (launches, timed, K)
    avg_ms = ms / timed
    peak = 2 * MFMA_PEAK_TFLOPS if fp8 else MFMA_PEAK_TFLOPS
    ach = B * NET_FLOP_PER_POS / (avg_ms * 1e-3) / 1e12
    prec = "fp8" if fp8 else "bf16"
    tr, tr_src = pmc_traffic("k_tower_fp8@8192" if fp8 else "k_tower_bf16@4096", positions_per_launch=B)
    return {"metric": "net_leaf_evals_per_s", "value": B * K * ctx.world / dt, "unit": "evals/s", "n_gpus": ctx.world,
            "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": prec, "data": "synthetic",
            "config": {"workload": f"reversi8x8_net_forward_batch{B}_{prec}",
                       "positions": "fixture F1 positions tiled to the batch"},
            "roofline": {"bound": "mfma", "kernel": "f8::k_tower_fp8" if fp8 else "k_tower_bf16", "achieved": ach,
                         "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": tr, "traffic_source": tr_src, "launches": launches,
                         "avg_launch_ms": avg_ms, "flop_per_launch": B * NET_FLOP_PER_POS}}


def run_train(ctx, batch, K, W):
    """SURVEY 8(f) row 4 inside the driver-timed process: K optimisation steps of the benchmark net (6 blocks x 128
    channels) on a synthetic data set, the whole step -- stem, tower, heads, losses, every gradient, Adam -- as 10
    hand-written kernels replayed as one HIP graph (betazero_amd.train.GraphedTrainStep).  The figure of merit is the
    convolution work (forward + backward-data + backward-weights of the 12 conv3x3 layers and the stem) per second
    against the dense bf16 MFMA peak."""
    import numpy as np
    import torch
    from betazero_amd.engine import DeviceExamples, Examples
    from betazero_amd.net import PolicyValueNet
    from betazero_amd.train import GraphedTrainStep
    C, NB = 128, 6
    rng = np.random.default_rng(0)
    n = 8 * batch
    x = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
    y = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
    pi = rng.random((n, 65)).astype(np.float32)
    pi /= pi.sum(1, keepdims=True)
    ex = DeviceExamples.from_host(Examples(x & ~y, y & ~x, pi, rng.integers(-1, 2, n).astype(np.int8), np.ones(n, np.int8),
                                           np.zeros(n, np.uint8), np.arange(n), np.zeros(n, np.int32), 8), ctx.dev)
    torch.manual_seed(0)
    step = GraphedTrainStep(PolicyValueNet(C, NB, 64, fused_tower=True).to(ctx.dev), lr=1e-3, batch=batch, device=ctx.dev)
    assert step.step_plan is not None and step.fused_adam   # the kernels, not autograd
    idx = [torch.randint(0, n, (batch,), device=ctx.dev) for _ in range(8)]
    first = None
    for i in range(W):
        out = step(ex, idx[i % 8])
        first = out if first is None else first
    ctx.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(K):
        out = step(ex, idx[i % 8])
    e1.record()
    ctx.barrier()
    dt = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1) / K
    step.check()   # the step's error word: an out-of-range row index in any of the steps raises here
    losses = [float(v) for v in out]
    assert all(np.isfinite(losses)) and losses[0] < float(first[0]), (losses, first)   # it is learning its 8 batches, not idling
    flop = 3 * (2 * 64 * 9 * C * C * 2 * NB + 2 * 64 * 9 * 2 * C) * batch
    ach = flop / (dev_ms * 1e-3) / 1e12
    tr, tr_src = pmc_traffic(f"train_step@{C}x{NB}x{batch}", channels=C, blocks=NB, batch=batch)
    return {"metric": "train_steps_per_s", "value": K / dt, "unit": "steps/s", "n_gpus": ctx.world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"train_step_convnet6x128_batch{batch}_bf16", "optimizer": "Adam (as its own kernel)",
                       "launches_per_step": 10, "loss_first_last": [float(first[0]), losses[0]]},
            "roofline": {"bound": "mfma", "kernel": "k_train_fwd + k_train_bwd + k_train_wgrad (+ the six end kernels)",
                         "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                         "traffic": tr, "traffic_source": tr_src,
                         "avg_launch_ms": dev_ms, "flop_per_launch": flop,
                         "note": "whole step (10 kernels) timed with events on the launch stream; flop = convolution work only"}}


def run_env(ctx, n, K, W):
    """the batched board-env step on its own (bz_reversi_step_batch: legal-move mask, apply-move / flip, terminal /
    winner for n games per launch; 42 algorithmic bytes per step): every game plays the lowest legal
    move of a random position (pass where there is none)"""
    import torch
    from betazero_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device=ctx.dev).manual_seed(0)
    a = torch.randint(0, 2**62, (n,), generator=g, device=ctx.dev, dtype=torch.int64)
    b = torch.randint(0, 2**62, (n,), generator=g, device=ctx.dev, dtype=torch.int64)
    own, opp = a & ~b, b & ~a
    legal = torch.empty(n, dtype=torch.int64, device=ctx.dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.bz_reversi_legal_batch(own.data_ptr(), opp.data_ptr(), n, legal.data_ptr(), st))
    low = legal & -legal  # stones live in bits 0..61, so the lowest legal bit is a positive power of two (exact in f64)
    idx = torch.where(low < 0, torch.full_like(low, 63), torch.log2(low.clamp(min=1).double()).long())  # bit 63 is the sign
    action = torch.where(legal == 0, torch.full_like(legal, 64), idx).to(torch.uint8)
    on, pn, ln = (torch.empty(n, dtype=torch.int64, device=ctx.dev) for _ in range(3))
    status = torch.empty(n, dtype=torch.uint8, device=ctx.dev)
    winner = torch.empty(n, dtype=torch.int8, device=ctx.dev)

    def step():
        _lib.check(L.bz_reversi_step_batch(own.data_ptr(), opp.data_ptr(), action.data_ptr(), n, on.data_ptr(),
                                           pn.data_ptr(), ln.data_ptr(), status.data_ptr(), winner.data_ptr(), st))
    for _ in range(W):
        step()
    L.bz_profile_reset()
    L.bz_profile_reserve(_lib.PROF_SLOTS.index("env_step"), K + 8)
    L.bz_profile_enable(1)
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    ctx.barrier()
    dt = time.perf_counter() - t0
    L.bz_profile_enable(0)
    launches, timed, ms = _lib.profile_read()["env_step"]
    assert launches == timed == K and int((status == _lib.ST_ILLEGAL).sum()) == 0
    avg_ms = ms / timed
    tr, tr_src = pmc_traffic("k_reversi_step", games=n)
    ach = 42.0 * n / (avg_ms * 1e-3) / 1e9
    return {"metric": "env_steps_per_s", "value": n * K * ctx.world / dt, "unit": "steps/s", "n_gpus": ctx.world, "steps": K,
            "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"reversi8x8_env_step_{n}games", "positions": "random stones, lowest legal move"},
            "roofline": {"bound": "hbm", "kernel": "k_reversi_step", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": tr, "traffic_source": tr_src, "launches": launches, "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": 42.0 * n,
                         "note": "bound by vector-instruction issue, not HBM: 315 vector instructions per step (SQ_INSTS_VALU, "
                                 "profiles/r05_pmc_env_sq_pmc.csv; ~98 flips by carry propagation + ~134 next legal mask + selects, status "
                                 "and I/O) at 4 cycles per wave64 instruction; profiles/r05_env_isa.txt has the histogram"}}


def run_ttt(ctx, B, sims, K, W, ttt_lanes=0):
    """BASELINE cfg 2: TTT, uniform priors, the whole search of a move is one fused tree kernel"""
    from betazero_amd import _lib
    from betazero_amd.engine import SelfPlayEngine
    L = _lib.lib()
    eng = SelfPlayEngine("ttt", B, sims, "uniform", game_id_base=ctx.rank * B, game_id_stride=ctx.world * B,
                         device=ctx.dev, ttt_lanes=ttt_lanes)

    def step():
        eng.reset_games()
        for _ in range(9):  # a TTT game has at most 9 moves; finished slots idle
            eng.search()
            eng.play(False)
    for _ in range(W):
        step()
    ctx.sync()
    eng.reset_counters()
    L.bz_profile_reset()
    L.bz_profile_reserve(_lib.PROF_SLOTS.index("search_fused"), 9 * K + 8)
    L.bz_profile_enable(1)
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    ctx.barrier()
    dt = time.perf_counter() - t0
    L.bz_profile_enable(0)
    eng.status()  # raises on engine error flags
    cnt = eng.counters()
    launches, timed, ms = _lib.profile_read()["search_fused"]
    assert launches == timed
    avg_ms = ms / max(timed, 1)
    tb = tree_bytes(cnt)
    ach = tb / max(launches, 1) / (avg_ms * 1e-3) / 1e9
    tr, tr_src = pmc_traffic("k_search_fused_ttt", games=B, sims=sims, ttt_lanes=ttt_lanes if ttt_lanes > 0 else 4) if ttt_lanes >= 0 else (None, None)
    return {"metric": "selfplay_games_per_s", "value": K * B * ctx.world / dt, "unit": "games/s", "n_gpus": ctx.world,
            "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "data": "synthetic", "dtype": "u64+f32",
            "config": {"workload": f"ttt3x3_{B}games_{sims}sims_uniform_tree_only", "games_per_gpu": B,
                       "sims_per_move": sims, "step": "one complete self-play iteration of all games",
                       "ttt_lanes": ttt_lanes if ttt_lanes else 4},
            "roofline": {"bound": "hbm", "kernel": "k_search_fused_ttt<%s, uniform>" % (ttt_lanes if ttt_lanes > 0 else 4) if ttt_lanes >= 0
                         else "k_search_fused<TicTacToe> (generic)", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": tr, "traffic_source": tr_src, "launches": launches,
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": tb / max(launches, 1)},
            "sims_per_s": cnt["n_sims"] * ctx.world / dt, "counters": cnt}


def run_reversi(ctx, args, B, sims, K, W):
    import torch
    import torch.distributed as dist
    from betazero_amd import _lib
    from betazero_amd import distributed as bd
    from betazero_amd.engine import PipelinedSelfPlay, packed_block_header, pipeline_stream_info
    from betazero_amd.net import DeviceNet, PolicyValueNet
    L = _lib.lib()
    prec = args.precision or "bf16"
    torch.manual_seed(0)
    mod = PolicyValueNet(128, 6, 64).round_to_bf16_()
    if prec == "fp8":
        from betazero_amd.quant import fake_quantize_fp8_
        fake_quantize_fp8_(mod)
    net = DeviceNet.from_module(mod, B, ctx.dev)
    rounds = 2 + (W + K) // 40
    # NS independent pipelines of B/NS games, each on its own HIP stream (betazero_amd.engine.PipelinedSelfPlay -- the
    # package's own self-play loop; this file keeps no copy of it): the tree step of one overlaps the net kernel of the
    # other and their net launches fill each other's tail wave.
    NS = max(1, args.streams)
    steady = args.mode == "steady"
    sp = PipelinedSelfPlay("reversi", B, sims, "net_" + prec, net, pipelines=NS, game_id_base=ctx.rank * B,
                           game_id_stride=ctx.world * B, device=ctx.dev, temp_moves=8, openings=1, seed=0, rounds=rounds,
                           stagger=PLIES_PER_GAME if steady else 0, reuse_subtree=args.reuse_subtree, run_ahead=args.run_ahead,
                           eval_cache=False if args.no_eval_cache else ("search" if args.eval_cache == "search" else True),
                           dirichlet_alpha=0.5 if args.dirichlet_eps > 0 else 0.0, dirichlet_eps=args.dirichlet_eps)
    engs, Bs = sp.engines, sp.sizes[0]
    sp.reset_games()
    ctx.sync()

    def step():
        if steady:
            sp.step(restart=True)   # one move for every slot; a finished slot starts its next game at once
        else:
            sp.run_iteration()      # every game from its opening to the last finished one

    note(f"reversi: engines up ({NS} x {Bs} games, {sims} sims; stream probe {pipeline_stream_info(ctx.dev, NS)}), warm-up {W} steps")
    for _ in range(W):
        step()
    ctx.sync()
    note("warm-up done")
    fin0 = sp.status()[1]
    sp.reset_counters()
    L.bz_profile_reset()
    prof_on = not args.no_kernel_timers
    PROF_CAP = 1 << 18                      # launches a timer slot holds (bz_abi.h)
    prof_steps = K                          # steps whose launches carry timers: all of them unless a soak run overflows the slots
    if prof_on and steady:                  # events exist before the timed loop: it only records them
        prof_steps = max(1, min(K, (PROF_CAP - 64) // (NS * (sims + 2))))
        per = prof_steps * NS * (sims + 2) + 64
        for slot in ("tower", "select", "expand_backup", "play"):
            L.bz_profile_reserve(_lib.PROF_SLOTS.index(slot), per)
    L.bz_profile_enable(1 if prof_on else 0)
    buf = None
    if ctx.dist:
        # both ends of the one exchange exist before the clock starts.  The packed block holds the finished games only;
        # its capacity (the same on every rank) bounds what W + K steps of the steady-state pool can finish: a slot
        # finishes a game every ~58 moves -- sized for one every 50, plus 10 steps of slack; a whole iteration: every slot.
        cap_games = min(rounds * B, -(-B * (W + K + 10) // 50)) if steady else B
        cdev = ctx.dev if ctx.backend == "nccl" else "cpu"
        buf = bd.GatherBuffers(sp.na, sp.packed_capacity(cap_games), ctx.world, ctx.dev, cdev)
        # untimed: the first all-gather of a process group sets up RCCL's channels and buffers
        wsend = torch.zeros(1 << 20, dtype=torch.uint8, device=cdev)
        wrecv = torch.empty(ctx.world << 20, dtype=torch.uint8, device=cdev)
        dist.all_gather_into_tensor(wrecv, wsend)
        del wsend, wrecv
    ctx.barrier()
    note(f"timed region: {K} steps")
    c0t, c0p = time.thread_time(), time.process_time()
    t0 = time.perf_counter()
    for i in range(K):
        if i == prof_steps and prof_on:
            L.bz_profile_enable(0)          # (a relaxed atomic store: no synchronisation, nothing waits)
        step()
    t_issued = time.perf_counter() - t0     # the host has queued every launch of the K steps
    c1t = time.thread_time()
    ctx.sync()
    if ctx.dist:  # the one exchange step: pool the finished games' (s, pi, z) -- pack kernels + ONE all-gather
        bd.gather_packed(sp, buffers=buf)
    ctx.barrier()
    dt = time.perf_counter() - t0
    c2t, c2p = time.thread_time(), time.process_time()
    L.bz_profile_enable(0)
    note(f"timed region done: {dt:.2f} s")
    fin1 = sp.status()[1]  # status() raises on engine error flags
    if not steady:
        fin1, fin0 = K * B, 0
    games = float(fin1 - fin0)
    own_games, own_dt = games, dt
    # how much of a host core this rank needs (8 ranks share the box's cores with RCCL's proxy threads): CPU time of the
    # launch thread while it queues the K steps, and of the thread / the whole process over the timed region
    host = {"run_ahead_sims": args.run_ahead, "launch_thread_cpu_s_while_issuing": c1t - c0t, "issue_wall_s": t_issued,
            "host_launch_cpu_frac": (c2t - c0t) / dt, "process_cpu_frac": (c2p - c0p) / dt,
            "launch_cpu_frac_while_issuing": (c1t - c0t) / max(t_issued, 1e-9),
            "launches_per_s": K * NS * (2 * sims + 4) / dt if steady else None}
    pooled = None
    if ctx.dist:  # untimed: what the collective delivered (each rank's header carries its own counts)
        heads = [packed_block_header(buf.out[r], strict=False) for r in range(ctx.world)]
        pooled = {"bytes_received_per_rank": int(buf.out.numel()), "block_bytes_per_rank": int(buf.nbytes),
                  "cap_rows": int(buf.cap_rows), "rows": int(sum(h["n_rows"] for h in heads)),
                  "games": int(sum(h["n_games"] for h in heads)), "collectives_in_timed_region": 1,
                  # rows that did not fit a rank's fixed-capacity block (0 unless the capacity estimate above was too small
                  # for this --steps: the exchange was timed all the same, the line just says that it was short)
                  "dropped_rows": int(sum(h["dropped_rows"] for h in heads))}
        t = torch.tensor([dt, games], dtype=torch.float64, device=ctx.dev if ctx.backend == "nccl" else "cpu")
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, games = float(tmax[0]), float(tsum[1])
    cnt = sp.counters()
    ranks = None
    if ctx.dist:  # untimed: which device every rank really held, what it finished and how long it took
        pr = torch.cuda.get_device_properties(ctx.local)
        mine = {"rank": ctx.rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device_index": ctx.local,
                "device_name": pr.name, "device_uuid": str(getattr(pr, "uuid", "")),
                "pci_bus_id": f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}",
                "games_finished": own_games, "seconds": own_dt, "pid": os.getpid(),
                "host_launch_cpu_frac": host["host_launch_cpu_frac"], "process_cpu_frac": host["process_cpu_frac"]}
        ranks = [None] * ctx.world
        dist.all_gather_object(ranks, mine)
    if ctx.rank != 0:
        return None
    out = {"metric": "selfplay_games_per_s", "value": games / dt, "unit": "games/s", "n_gpus": ctx.world, "steps": K,
           "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "data": "synthetic", "dtype": prec,
           "config": {"workload": f"reversi8x8_{B}games_{sims}sims_convnet6x128_{prec}",
                      "games_per_gpu": B, "sims_per_move": sims, "c_puct": 1.5, "temp_moves": 8, "openings": 12,
                      "evaluator": "policy/value conv tower 6x128 (226.86 MFLOP per leaf), random init seed 0",
                      "evaluation_cache": ({"on": True, "scope": "inside a search only" if args.eval_cache == "search" else "inside a search + the slot's previous search",
                                            "evaluations_shared": cnt["n_cache_hits"],
                                            "of_which_from_the_previous_search": cnt["n_cache_hits_prev"], "evaluations_computed": cnt["n_net_leaves"],
                                            "shared_fraction": cnt["n_cache_hits"] / max(1, cnt["n_cache_hits"] + cnt["n_net_leaves"]),
                                            "note": "a position met again inside one search, or evaluated by the slot's previous search (after a move the "
                                                    "played child's old subtree is re-created node for node), takes that evaluation: bit-identical "
                                                    "results (tests/test_gpu_parity.py); roofline / net_evals_per_s count computed rows only"}
                                           if not args.no_eval_cache else {"on": False}),
                      "step": "one move for all concurrent games (steady-state pool, staggered starts)"
                      if args.mode == "steady" else "one complete self-play iteration (all games, start to end)",
                      "pipelines": f"{NS} x {Bs} games on separate HIP streams",
                      **({"supplementary_features": {"reuse_subtree": args.reuse_subtree, "dirichlet_eps": args.dirichlet_eps}}
                         if (args.reuse_subtree or args.dirichlet_eps > 0) else {}),
                      "stream_probe": pipeline_stream_info(ctx.dev, NS),
                      "parallelism": f"games sharded over {ctx.world} GPU(s), one all-gather of the packed example blocks"
                                     + (f" ({pooled['bytes_received_per_rank']} bytes received per rank)" if pooled else "")}}
    out["host"] = host
    if pooled:
        out["pooled"] = pooled
    peak = MFMA_PEAK_TFLOPS if prec == "bf16" else 2 * MFMA_PEAK_TFLOPS  # dense fp8 = 5 PF
    kname = ("k_tower_bf16" if prec == "bf16" else "f8::k_tower_fp8") + " (stem + 12 conv3x3 + heads, fused)"
    if prof_on:
        prof = _lib.profile_read()
        note("kernel timers read")
        launches, timed, ms = prof["tower"]
        assert timed == launches, f"kernel-timer capacity exceeded ({timed} of {launches} launches timed)"
        # leaves are packed before the net runs: a launch evaluates only the non-terminal leaves.  When only the first
        # prof_steps steps carried timers (soak runs), positions per launch is the whole run's mean: every step of the
        # steady-state pool issues the same NS x (sims + 1) net launches
        all_launches = K * NS * (sims + 1) if (args.mode == "steady" and prof_steps < K) else launches
        flop_per_launch = cnt["n_net_leaves"] / max(all_launches, 1) * NET_FLOP_PER_POS  # stem+tower+heads = ONE kernel
        union_ms, sum_ms = _lib.profile_union_ms("tower")
        ach = flop_per_launch * launches / (union_ms * 1e-3) / 1e12
        out["roofline"] = {
            "bound": "mfma", "kernel": kname, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "traffic": None, "launches": launches, "flop_per_launch": flop_per_launch,
            "positions_per_launch": flop_per_launch / NET_FLOP_PER_POS,
            **({"timed_steps": prof_steps} if prof_steps < K else {}),
            # the NS pipelines' launches overlap on the chip, so a launch's own event-to-event duration is
            # SHARED time.  `achieved` = flop_per_launch x launches / busy_ms, busy_ms = union of all launch
            # intervals (every launch timed, nothing extrapolated); avg_launch_ms = busy_ms / launches is the
            # chip time one launch costs.  The shared-time view is kept beside it.
            "busy_ms": union_ms, "avg_launch_ms": union_ms / max(launches, 1),
            "shared_time": {"avg_launch_ms": sum_ms / max(launches, 1), "concurrent_launches": sum_ms / union_ms,
                            "tflops_per_launch": flop_per_launch / (sum_ms / max(launches, 1) * 1e-3) / 1e12}}
        # HBM traffic of the tower: PMC counters cannot run inside this (timed) process, so the figure comes from the
        # committed counter passes of the SAME command (tools/profile_round.sh pmc_bench: FETCH_SIZE / WRITE_SIZE, each in
        # its own run; gfx950 fetch correction applied); it is attached only when it was measured at this run's shape
        # (precision, pipelines) and says at how many positions per launch
        ppl = flop_per_launch / NET_FLOP_PER_POS
        for path in PMC_TRAFFIC:
            try:
                tr = json.load(open(os.path.join(ROOT, path)))["k_tower_bf16" if prec == "bf16" else "k_tower_fp8@selfplay"]
            except Exception:
                continue
            tp = float(tr.get("positions_per_launch", 4096))
            if abs(tp - ppl) <= 0.1 * ppl:
                out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command ({path}: "
                                                     f"{tp:.0f} positions per launch; this run: {ppl:.0f})")
                break
        if out["roofline"]["traffic"] is None:
            out["roofline"]["traffic_note"] = "no --pmc FETCH_SIZE / WRITE_SIZE pass has been committed for this shape (precision, pipelines, positions per launch)"
        tb = tree_bytes(cnt) * (prof_steps / K if (args.mode == "steady" and prof_steps < K) else 1.0)  # the timed steps' share
        t_union, t_sum = _lib.profile_union_ms("select")
        t_tr, t_src = pmc_traffic("k_tree_step", games_per_launch=Bs, sims=sims) if prec == "bf16" else (None, None)
        out["roofline_tree"] = {"bound": "hbm", "kernels": "k_tree_step (expand + backup + select, 16 lanes per game)",
                                "achieved": tb / (t_sum * 1e-3) / 1e9 if t_sum else None, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": tb / (t_sum * 1e-3) / 1e9 / HBM_PEAK_GBS if t_sum else None,
                                "algorithmic_bytes": tb, "kernel_ms": t_sum, "launches": prof["select"][0],
                                "traffic": t_tr * prof["select"][0] if t_tr else None, "traffic_per_launch": t_tr, "traffic_source": t_src,
                                "note": "latency-bound pointer chase, overlapped with the other pipeline's net launch"}
        out["kernel_ms_total"] = {k: round(v[2], 3) for k, v in prof.items() if v[0]}
    else:
        out["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": cnt["n_net_leaves"] * NET_FLOP_PER_POS / dt / 1e12,
                           "peak": peak, "unit": "TFLOP/s", "frac": cnt["n_net_leaves"] * NET_FLOP_PER_POS / dt / 1e12 / peak,
                           "traffic": None, "note": "--no-kernel-timers: whole-run flops / wall time (lower bound)"}
    if ranks is not None:
        out["ranks"] = {"backend": ctx.backend + (" (RCCL)" if ctx.backend == "nccl" else ""), "world_size": ctx.world,
                        "distinct_devices": len({(r["device_uuid"] or r["pci_bus_id"], r["device_index"]) for r in ranks}),
                        "per_rank": ranks}
    out["sims_per_s"] = cnt["n_sims"] * ctx.world / dt
    out["net_evals_per_s"] = cnt["n_net_leaves"] * ctx.world / dt
    out["net_tflops_e2e"] = cnt["n_net_leaves"] * ctx.world * NET_FLOP_PER_POS / dt / 1e12
    out["counters"] = cnt
    return out


_JSON_FD = None


def claim_stdout():
    """keep the process's stdout for the ONE JSON line: RCCL prints a version banner on stdout when the first communicator is
    built (seen on the box: five lines before the JSON of a --force-collective run), and any other native library may do the
    same.  From here on file descriptor 1 is stderr for everybody; emit_json writes to the saved descriptor."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit_json(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def main():
    # multi-process GPU work on this image needs dmabuf IPC (the host driver does not support the legacy mode: RCCL set-up and
    # tensor sharing otherwise fail with "hipIpcGetMemHandle: invalid argument"); the image exports this already -- make sure a
    # launcher with a scrubbed environment still gets it, before anything initialises HIP
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="reversi", choices=["reversi", "ttt", "net", "env", "train"],
                    help="reversi = BASELINE cfg 3 (default, the metric), ttt = cfg 2, net = cfg 5 (net forward only), "
                         "env = the batched board-env step kernel alone, train = the training step of the benchmark net")
    ap.add_argument("--games", type=int, default=None, help="concurrent games per GPU (default: BASELINE config)")
    ap.add_argument("--sims", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short cfg-2 / cfg-5 runs after the headline one")
    ap.add_argument("--no-kernel-timers", action="store_true",
                    help="leave the in-library HIP-event timers off (A/B of their cost on `value`)")
    ap.add_argument("--precision", default=None, choices=["bf16", "fp8"],
                    help="net precision; default bf16 for the self-play workload (the BASELINE config; fp8 is a "
                         "supplementary line) and fp8 for --workload net (cfg 5)")
    ap.add_argument("--mode", default="steady", choices=["steady", "iteration"],
                    help="reversi: steady = one move per step on a staggered pool (default); iteration = a step is a\n"
                         "complete self-play iteration from the start position to the last finished game (cross-check)")
    ap.add_argument("--streams", type=int, default=2, help="independent half-batch pipelines per GPU (reversi)")
    ap.add_argument("--run-ahead", type=int, default=16,
                    help="simulations the host launch thread may queue ahead of the GPU (PipelinedSelfPlay.run_ahead; 0 = unbounded: "
                         "the thread then spins on the runtime's full queue, a whole core per rank)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = --games per GPU (default, the headline); strong = --games in total, split over the ranks")
    ap.add_argument("--eval-cache", default="carry", choices=["carry", "search", "off"],
                    help="the engine's evaluation cache: carry = inside a search + from the slot's previous search (default), search = "
                         "inside a search only, off = every leaf through the net (= --no-eval-cache); results are identical in all three")
    ap.add_argument("--no-eval-cache", action="store_true",
                    help="every non-terminal leaf through the net, repeats of a position inside one search included (the engine's "
                         "evaluation cache off; results are identical either way)")
    ap.add_argument("--reuse-subtree", action="store_true", help="supplementary: keep the chosen child's subtree (DESIGN 3.10)")
    ap.add_argument("--dirichlet-eps", type=float, default=0.0, help="supplementary: root noise weight (alpha 0.5; DESIGN 3.9)")
    ap.add_argument("--force-collective", action="store_true", default=os.environ.get("BZ_BENCH_FORCE_DIST") == "1",
                    help="take the N > 1 path at ANY world size, 1 included (also BZ_BENCH_FORCE_DIST=1): process group on the chosen "
                         "backend (nccl = RCCL with device_id), GatherBuffers on the device, pack kernels, the one all_gather_into_tensor, "
                         "the device-side all_reduces, ranks.per_rank, pooled -- so that the multi-GPU branch can be run on a one-GPU box")
    ap.add_argument("--ttt-lanes", type=int, default=0, choices=[-1, 0, 1, 2, 4, 8],
                    help="--workload ttt: lanes per game of the fused search (bz_engine_cfg.ttt_lanes; 0 = default, -1 = generic kernel)")
    args = ap.parse_args()
    claim_stdout()
    if args.eval_cache == "off":
        args.no_eval_cache = True

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
        if args.force_collective:  # a world of one rank, no launcher needed: this process is rank 0 of 1
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                               "MASTER_PORT": str(s.getsockname()[1])})
            s.close()
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: start me as "
                 f"`python bench.py --gpus N` or under torchrun with --nproc-per-node equal to --gpus")

    import torch
    import torch.distributed as dist
    ctx = Ctx()
    ctx.rank = int(os.environ.get("RANK", "0"))
    ctx.world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ctx.backend = os.environ.get("BZ_DIST_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 path on one GPU / on CPU
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        if os.environ.get("BZ_BENCH_REHEARSAL") != "1" or ctx.world < 2:
            sys.exit("bench.py: no GPU visible (the engine has no CPU path; BZ_BENCH_REHEARSAL=1 with --gpus >= 2 "
                     "rehearses the multi-rank plumbing only)")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        rehearsal(args, ctx.rank, ctx.world)
        dist.destroy_process_group()
        return
    if local >= n_dev:
        if ctx.backend == "nccl":
            sys.exit(f"bench.py: --gpus {ctx.world} but only {n_dev} GPU(s) visible: RCCL needs one device per rank "
                     f"(BZ_DIST_BACKEND=gloo rehearses the multi-rank path with ranks sharing a card)")
        local = local % n_dev  # gloo rehearsal: the ranks share the visible card(s) and say so in the line
    ctx.dev = f"cuda:{local}"
    ctx.local = local
    torch.cuda.set_device(local)
    ctx.dist = ctx.world > 1 or args.force_collective   # the multi-rank path (always at N > 1; on request at N = 1)
    if ctx.dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if ctx.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(ctx.dev))
        else:
            dist.init_process_group(ctx.backend)
        assert dist.get_world_size() == args.gpus
    ctx.sync = torch.cuda.synchronize

    def barrier():
        if ctx.dist:
            dist.barrier()
        torch.cuda.synchronize()
    ctx.barrier = barrier

    from betazero_amd import _lib
    _lib.require_gpu()

    if args.workload == "net":  # BASELINE cfg 5: leaf-eval batch 8192, fp8 e4m3 net, MFMA-utilisation run
        out = run_net(ctx, args.games or 8192, args.steps if args.steps is not None else 1000,
                      args.warmup if args.warmup is not None else 50, (args.precision or "fp8") == "fp8")
    elif args.workload == "train":
        out = run_train(ctx, args.games or 1024, args.steps if args.steps is not None else 200,
                        args.warmup if args.warmup is not None else 20)
    elif args.workload == "env":
        out = run_env(ctx, args.games or (1 << 26), args.steps if args.steps is not None else 20,
                      args.warmup if args.warmup is not None else 3)
    elif args.workload == "ttt":
        sims = args.sims or 50
        out = run_ttt(ctx, args.games or 65536, sims, args.steps if args.steps is not None else 20,
                      args.warmup if args.warmup is not None else 2, args.ttt_lanes)
        if ctx.rank == 0 and ctx.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_ttt(sims)
    else:
        sims = args.sims or 800
        games = args.games or 4096
        if args.scaling == "strong":
            assert games % ctx.world == 0, "--scaling strong: --games must divide by the number of ranks"
            games //= ctx.world
        out = run_reversi(ctx, args, games, sims, args.steps if args.steps is not None else 8,
                          args.warmup if args.warmup is not None else 1)
        if out is not None:
            out["scaling"] = args.scaling
        if ctx.rank == 0 and ctx.world == 1:
            if not args.no_secondary:  # cfg 2 and cfg 5 in the same driver-timed process (a few seconds)
                sec = {}
                a_fp8 = argparse.Namespace(**{**vars(args), "precision": "fp8", "mode": "steady"})
                a_iter = argparse.Namespace(**{**vars(args), "mode": "iteration"})
                a_nocache = argparse.Namespace(**{**vars(args), "no_eval_cache": True, "mode": "steady"})
                a_insearch = argparse.Namespace(**{**vars(args), "eval_cache": "search", "mode": "steady"})
                for name, fn in (
                        # cfg 5 as SURVEY 7 reads it: the fp8 net as the in-loop evaluator of 8192 concurrent games
                        ("cfg5_selfplay", lambda: run_reversi(ctx, a_fp8, 8192, sims, 6, 2)),
                        # the steady-state figure's cross-check: ONE complete iteration, every game from its
                        # opening to the last finished game, under the same clock
                        ("cfg3_iteration", lambda: run_reversi(ctx, a_iter, games, sims, 1, 0)),
                        # the headline workload with every leaf through the net (evaluation cache off): same results, more rows
                        ("cfg3_no_eval_cache", lambda: run_reversi(ctx, a_nocache, games, sims, 4, 1)),
                        # ... and with the cache restricted to repeats INSIDE one search (nothing taken from the previous search)
                        ("cfg3_eval_cache_in_search_only", lambda: run_reversi(ctx, a_insearch, games, sims, 4, 1)),
                        ("cfg2", lambda: run_ttt(ctx, 65536, 50, 20, 2)),
                        ("cfg5", lambda: run_net(ctx, 8192, 1000, 50, True)),
                        ("env_step", lambda: run_env(ctx, 1 << 26, 20, 3)),
                        # the training step of the benchmark net, all on hand-written kernels
                        ("train_step", lambda: run_train(ctx, 1024, 200, 20))):
                    try:
                        note(f"secondary {name}")
                        r = fn()
                        sec[name] = {k: r[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "dtype", "config",
                                                       "roofline", "cores", "games", "plies_per_game", "results_x_o_draw", "host",
                                                       "temp_moves", "distinct_move_sequences",
                                                       "all_games_legal_and_finished", "seconds") if k in r}
                    except Exception as e:
                        sec[name] = {"error": repr(e)}
                out["secondary"] = sec
            if not args.no_cpu_baseline:
                note("cpu baseline (oracle port on all cores and on one thread, then the Python loops on one core)")
                out["cpu_baseline"] = cpu_baseline_reversi(sims)
                # BASELINE cfg 1 IS a CPU configuration: it belongs to the CPU-baseline leg (the only part of this file that
                # may touch oracle/); `secondary.cfg1` carries the same object so that every config has its entry there
                try:
                    c1 = cfg1_python_loop()
                    c1["kind"] = "cpu_baseline leg: build-authored Python MCTS (oracle/py_twin.py) over betazero_amd.TicTacToeBoard"
                except Exception as e:
                    c1 = {"error": repr(e)}
                out["cpu_baseline"]["cfg1"] = c1
                if "secondary" in out:
                    out["secondary"]["cfg1"] = c1
                note("cpu baseline done")
    if ctx.rank == 0:
        out["n_gpus"] = dist.get_world_size() if ctx.dist else 1
        assert out["n_gpus"] == args.gpus
        if args.force_collective:
            out["forced_collective_path"] = True
        emit_json(out)
    if ctx.dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
