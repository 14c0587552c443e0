#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING the reference (this container only).

Test infrastructure, not product code.  Refers to the reference only by path
(/root/reference); writes data-only fixtures into tests/golden/.  On the GPU
box /root/reference does not exist and this script is never run there.

Fixtures:
  F1 reversi_random_games.npz   full random games, sizes 8/6/4, every ply
  F2 reversi_positions.npz      arbitrary (unreachable) positions: legal masks,
                                game-over, score, move results  (wrap probes)
  F3 ttt_exhaustive.npz         all reachable TTT positions + double-line case
  F4 ttt_csv.npz                the reference's tic_tac_toe_data.csv as arrays
  F5 strings.json               __str__/__repr__ text, demo sequences
  F6 illegal_moves.json         (position, move) pairs that raise ValueError
  F7 mcts_twin.npz / .json      build-authored MCTS twin over the REFERENCE's
                                board classes (every env transition inside the
                                search is reference-computed)
  F9 minimax_players.npz        decisions of the reference's OptimalPlayer classes
                                (Reversi depth-limited minimax, TTT full minimax)
  F10 mcts_twin_features.*      the twin over the reference's boards with Dirichlet
                                root noise / subtree reuse switched on
  F11 reversi_other_sizes.npz   F1 / F2 again for the board sizes the reference accepts besides
                                4/6/8 that fit a 64-bit board: 1, 2, 3, 5, 7 (reversi_board.py:4-14)
  F12 off_domain.json           cells and players outside {-1, 0, +1}: the reference treats any
                                non-zero cell as occupied (tic_tac_toe_board.py:20-21,
                                reversi_board.py:26) and plays `player` against `-player`

Run:  python oracle/gen_golden.py
"""
import json
import os
import random
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src/reversi/game_logic"))
sys.path.insert(0, os.path.join(REF, "src/tic_tac_toe"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reversi_board import ReversiBoard  # noqa: E402
from tic_tac_toe_board import TicTacToeBoard  # noqa: E402

OUT = os.environ.get("BZ_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


from py_twin import (M64, Twin as _Twin, eval_hash, expf_spec, f32, mix64, rev_bits, rev_mask, rng_draw,  # noqa: E402,F401
                     ttt_bits)


# ---------------------------------------------------------------- F1
def gen_f1():
    random.seed(1234)
    rows = []  # game, size, turn, to_move+1, x, o, legal, action(255=pass), flips, over_after
    finals = []  # game, size, winner+1, n_plus, n_minus, passes, x_final, o_final
    gid = 0
    for size, ngames in ((8, 200), (6, 60), (4, 60)):
        for _ in range(ngames):
            b = ReversiBoard(size=size)
            cur, over, turn, passes = 1, False, 0, 0
            while not over:  # mirrors reversi_terminal.py:16-38
                moves = b.generate_possible_moves(cur)
                x, o = rev_bits(b.board, 1), rev_bits(b.board, -1)
                if moves:
                    r, c = random.choice(moves)
                    nb = b.make_move(r, c, cur)
                    flips = rev_bits(nb.board, cur) & ~rev_bits(b.board, cur) & ~(1 << (8 * r + c))
                    act = 8 * r + c
                    b = nb
                else:
                    flips, act = 0, 255
                    passes += 1
                over = b.is_game_over()
                rows.append((gid, size, turn, cur + 1, x, o, rev_mask(moves), act, flips, int(over)))
                cur *= -1
                turn += 1
            w, (n1, n2) = b.get_score()
            finals.append((gid, size, w + 1, n1, n2, passes, rev_bits(b.board, 1), rev_bits(b.board, -1)))
            gid += 1
    rows = np.array(rows, dtype=np.uint64)
    finals = np.array(finals, dtype=np.uint64)
    np.savez_compressed(os.path.join(OUT, "reversi_random_games.npz"), rows=rows, finals=finals)
    print("F1", rows.shape, finals.shape)


# ---------------------------------------------------------------- F2
def gen_f2():
    rng = random.Random(99)
    pos = []  # size, x, o, legal(+1), legal(-1), over, winner+1, n_plus, n_minus
    moves = []  # pos_index, player(+1 -> 1, -1 -> 0), action, x_after, o_after
    for size in (8, 6, 4):
        for k in range(400):
            b = ReversiBoard(size=size)
            dens = rng.random()
            bias = rng.random()
            for r in range(size):
                for c in range(size):
                    u = rng.random()
                    b.board[r][c] = 0 if u > dens else (1 if rng.random() < bias else -1)
            if k % 7 == 0:  # dense edges: every a-/h-file and first/last row square occupied
                for i in range(size):
                    for (r, c) in ((i, 0), (i, size - 1), (0, i), (size - 1, i)):
                        if rng.random() < 0.8:
                            b.board[r][c] = rng.choice((1, -1))
            l1 = b.generate_possible_moves(1)
            l2 = b.generate_possible_moves(-1)
            w, (n1, n2) = b.get_score()
            pi = len(pos)
            pos.append((size, rev_bits(b.board, 1), rev_bits(b.board, -1), rev_mask(l1), rev_mask(l2),
                        int(b.is_game_over()), w + 1, n1, n2))
            for player, lst in ((1, l1), (-1, l2)):
                for (r, c) in lst:
                    nb = b.make_move(r, c, player)
                    moves.append((pi, 1 if player == 1 else 0, 8 * r + c, rev_bits(nb.board, 1),
                                  rev_bits(nb.board, -1)))
    np.savez_compressed(os.path.join(OUT, "reversi_positions.npz"),
                        pos=np.array(pos, dtype=np.uint64), moves=np.array(moves, dtype=np.uint64))
    print("F2", len(pos), len(moves))


# ---------------------------------------------------------------- F3
def gen_f3():
    seen = {}
    order = []
    stack = [(TicTacToeBoard(), 1)]
    while stack:
        b, cur = stack.pop()
        key = (ttt_bits(b.board, 1), ttt_bits(b.board, -1))
        if key in seen:
            continue
        over, w = b.is_game_over()
        legal = 0
        for r, c in b.generate_possible_moves():
            legal |= 1 << (3 * r + c)
        seen[key] = 1
        order.append((key[0], key[1], cur & 3, legal, int(over), 2 if w is None else w + 1))
        if not over:
            for r, c in b.generate_possible_moves():
                stack.append((b.make_move(r, c, cur), -cur))
    assert len(order) == 5478, len(order)
    # unreachable: both players own a line -> (True, 1) (+1 is tested first)
    b = TicTacToeBoard(np.array([[1, 1, 1], [-1, -1, -1], [0, 0, 0]]))
    over, w = b.is_game_over()
    extra = [(ttt_bits(b.board, 1), ttt_bits(b.board, -1), 1, 0b111000000, int(over), w + 1)]
    b = TicTacToeBoard(np.array([[-1, 1, 0], [-1, 1, 0], [-1, 1, 0]]))
    over, w = b.is_game_over()
    extra.append((ttt_bits(b.board, 1), ttt_bits(b.board, -1), 1, 0b100100100, int(over), w + 1))
    np.savez_compressed(os.path.join(OUT, "ttt_exhaustive.npz"),
                        pos=np.array(order, dtype=np.int64), extra=np.array(extra, dtype=np.int64))
    print("F3", len(order))


# ---------------------------------------------------------------- F4
def gen_f4():
    import csv
    st, ac = [], []
    with open(os.path.join(REF, "tic_tac_toe_data.csv")) as f:
        rd = csv.reader(f)
        next(rd)
        for row in rd:
            st.append([int(v) for v in row[0].split()])
            ac.append([int(v) for v in row[1].split()])
    np.savez_compressed(os.path.join(OUT, "ttt_csv.npz"), states=np.array(st, dtype=np.int64),
                        actions=np.array(ac, dtype=np.int64))
    print("F4", len(st))


# ---------------------------------------------------------------- F5 / F6
def gen_f5_f6():
    out = {"reversi": [], "ttt": []}
    b = ReversiBoard(size=4)  # demo, reversi_board.py:92-99
    seq = [(0, 2, 1), (0, 1, -1), (2, 0, 1)]
    boards = [b]
    for r, c, p in seq:
        b = b.make_move(r, c, p)
        boards.append(b)
    for bb in boards + [ReversiBoard(), ReversiBoard(size=6)]:
        out["reversi"].append({"size": bb.size, "board": bb.board.tolist(), "str": str(bb), "repr": repr(bb)})
    out["reversi_demo"] = {"moves": seq, "final": b.board.tolist(),
                           "valid_0_3_minus1": bool(b.is_valid_move(0, 3, -1))}
    t = TicTacToeBoard()  # demo, tic_tac_toe_board.py:45-52
    tb = [t]
    for r, c, p in [(0, 0, 1), (0, 1, -1), (0, 2, 1)]:
        t = t.make_move(r, c, p)
        tb.append(t)
    for tt in tb:
        out["ttt"].append({"board": tt.board.tolist(), "str": str(tt), "repr": repr(tt)})
    out["ttt_demo_valid_0_0"] = bool(t.is_valid_move(0, 0))
    with open(os.path.join(OUT, "strings.json"), "w") as f:
        json.dump(out, f, indent=1)

    rng = random.Random(5)
    ill = {"reversi": [], "ttt": []}
    for size in (8, 6, 4):
        b = ReversiBoard(size=size)
        cur = 1
        for _ in range(12):
            mv = b.generate_possible_moves(cur)
            bad = [(r, c) for r in range(-1, size + 1) for c in range(-1, size + 1) if (r, c) not in mv]
            for (r, c) in rng.sample(bad, 6):
                try:
                    b.make_move(r, c, cur)
                    raise AssertionError("reference accepted an illegal move")
                except ValueError as e:
                    ill["reversi"].append({"size": size, "board": b.board.tolist(), "player": cur,
                                           "move": [r, c], "msg": str(e)})
                except IndexError:
                    pass  # negative/out-of-range probes never index: reference returns False first
            if not mv:
                break
            b = b.make_move(*rng.choice(mv), cur)
            cur = -cur
    t = TicTacToeBoard()
    cur = 1
    for _ in range(5):
        mv = t.generate_possible_moves()
        for (r, c) in [(r, c) for r in range(-1, 4) for c in range(-1, 4) if (r, c) not in mv]:
            try:
                t.make_move(r, c, cur)
                raise AssertionError
            except ValueError as e:
                ill["ttt"].append({"board": t.board.tolist(), "player": cur, "move": [r, c], "msg": str(e)})
        t = t.make_move(*rng.choice(mv), cur)
        cur = -cur
    with open(os.path.join(OUT, "illegal_moves.json"), "w") as f:
        json.dump(ill, f)
    print("F5/F6", len(ill["reversi"]), len(ill["ttt"]))


# ---------------------------------------------------------------- F7: MCTS twin
class Twin(_Twin):
    """oracle/py_twin.py bound to the REFERENCE's board classes"""

    def __init__(self, game, eval_kind, c_puct=1.5, **kw):
        super().__init__(game, eval_kind, c_puct, boards=(ReversiBoard, TicTacToeBoard), **kw)



def gen_f7():
    out = {}
    meta = {"cases": []}
    ci = 0

    def add_search(game, eval_kind, b, p, sims, label):
        nonlocal ci
        tw = Twin(game, eval_kind)
        root = tw.search(b, p, sims)
        na = tw.na
        N = np.zeros(na, np.uint32); W = np.zeros(na, np.float32); P = np.zeros(na, np.float32)
        for e in root["edges"]:
            N[e["a"]], W[e["a"]], P[e["a"]] = e["N"], e["W"], e["P"]
        own, opp = tw.bits(b, p)
        out[f"s{ci}_N"], out[f"s{ci}_W"], out[f"s{ci}_P"] = N, W, P
        meta["cases"].append({"id": ci, "kind": "search", "game": game, "eval": eval_kind, "own": own,
                              "opp": opp, "to_move": p, "sims": sims, "label": label})
        ci += 1

    # TTT searches
    add_search("ttt", "uniform", TicTacToeBoard(), 1, 50, "empty, cfg2 setting")
    add_search("ttt", "hash", TicTacToeBoard(), 1, 200, "empty")
    t = TicTacToeBoard().make_move(1, 1, 1).make_move(0, 0, -1).make_move(2, 2, 1)
    add_search("ttt", "hash", t, -1, 120, "mid")
    add_search("ttt", "uniform", t, -1, 300, "mid, tree exhausts into terminals")
    # Reversi searches (env = reference ReversiBoard)
    add_search("reversi", "uniform", ReversiBoard(), 1, 40, "start")
    add_search("reversi", "hash", ReversiBoard(), 1, 150, "start")
    rnd = random.Random(77)
    b, p = ReversiBoard(), 1
    for k in range(52):  # late-game position: passes and terminals inside the tree
        mv = b.generate_possible_moves(p)
        if mv:
            b = b.make_move(*rnd.choice(mv), p)
        p = -p
        if k in (20, 40, 51) and b.generate_possible_moves(p) and not b.is_game_over():
            add_search("reversi", "hash", ReversiBoard(b), p, 120, f"random ply {k}")
    # hand-made pass position inside the tree: X to move, after X's move O must pass
    # self-play trajectories
    def add_selfplay(game, eval_kind, gid, sims, temp_moves, openings, seed):
        nonlocal ci
        tw = Twin(game, eval_kind)
        ex, w, passes = tw.selfplay(gid, sims, temp_moves, openings, seed)
        out[f"g{ci}_own"] = np.array([e[0] for e in ex], dtype=np.uint64)
        out[f"g{ci}_opp"] = np.array([e[1] for e in ex], dtype=np.uint64)
        out[f"g{ci}_pi"] = np.array([e[2] for e in ex], dtype=np.float32)
        out[f"g{ci}_mover"] = np.array([e[3] for e in ex], dtype=np.int8)
        out[f"g{ci}_act"] = np.array([e[4] for e in ex], dtype=np.uint8)
        meta["cases"].append({"id": ci, "kind": "selfplay", "game": game, "eval": eval_kind, "gid": gid,
                              "sims": sims, "temp_moves": temp_moves, "openings": openings, "seed": seed,
                              "winner": int(w), "passes": passes})
        ci += 1

    add_selfplay("ttt", "uniform", 0, 25, 0, 0, 0)      # BASELINE cfg 1 setting
    add_selfplay("ttt", "hash", 3, 25, 0, 0, 0)
    add_selfplay("ttt", "hash", 5, 40, 4, 0, 7)
    add_selfplay("reversi", "hash", 0, 12, 0, 0, 0)
    add_selfplay("reversi", "hash", 7, 10, 8, 1, 0)     # cfg 3 diversification rules
    add_selfplay("reversi", "uniform", 10, 8, 8, 1, 3)
    # the reference's demo board sizes (appended: earlier case ids stay stable)
    add_search("reversi6", "hash", ReversiBoard(size=6), 1, 120, "6x6 start")
    add_search("reversi4", "hash", ReversiBoard(size=4), 1, 200, "4x4 start: the tree runs into terminals")
    b6, p6 = ReversiBoard(size=6), 1
    for k in range(14):
        mv = b6.generate_possible_moves(p6)
        if mv:
            b6 = b6.make_move(*rnd.choice(mv), p6)
        p6 = -p6
    if b6.generate_possible_moves(p6) and not b6.is_game_over():
        add_search("reversi6", "uniform", b6, p6, 90, "6x6 random ply 14")
    add_selfplay("reversi6", "hash", 2, 14, 6, 0, 5)
    add_selfplay("reversi4", "hash", 1, 20, 2, 0, 0)
    add_selfplay("reversi4", "uniform", 3, 30, 0, 0, 0)
    np.savez_compressed(os.path.join(OUT, "mcts_twin.npz"), **out)
    with open(os.path.join(OUT, "mcts_twin.json"), "w") as f:
        json.dump(meta, f, indent=1)
    # expf spot values
    xs = np.concatenate([np.linspace(-90, 0, 721), -np.logspace(-8, 1.9, 200)]).astype(np.float32)
    ys = np.array([expf_spec(x) for x in xs], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "expf_spec.npz"), x=xs, y=ys)
    print("F7", ci, "cases")


# ---------------------------------------------------------------- F8: augmentation + CSV text
def gen_f8():
    """expand_with_transforms (SL/train.py:24-52) cannot be imported (train.py trains at import),
    so its 8 torch expressions are applied here to the reference's own CSV rows: the expected
    expanded + deduplicated dataset, in insertion order.  Also the CSV's checksum."""
    import csv
    import hashlib
    import torch
    transforms = [
        lambda x: x, lambda x: x.flip(dims=[0]), lambda x: x.flip(dims=[1]), lambda x: x.rot90(1, [0, 1]),
        lambda x: x.rot90(2, [0, 1]), lambda x: x.rot90(3, [0, 1]), lambda x: x.t(), lambda x: x.flip(dims=[0]).t(),
    ]
    path = os.path.join(REF, "tic_tac_toe_data.csv")
    raw = open(path, "rb").read()
    st, ac, src, tr = [], [], [], []
    seen = set()
    with open(path) as f:
        rd = csv.reader(f)
        next(rd)
        for i, row in enumerate(rd):
            s = torch.tensor([float(v) for v in row[0].split()]).view(3, 3)
            a = torch.tensor([float(v) for v in row[1].split()]).view(3, 3)
            for t, fn in enumerate(transforms):
                ts, ta = fn(s), fn(a)
                key = (",".join(map(str, ts.reshape(-1).tolist())), ",".join(map(str, ta.reshape(-1).tolist())))
                if key not in seen:
                    seen.add(key)
                    st.append(ts.reshape(-1).to(torch.int64).tolist()); ac.append(ta.reshape(-1).to(torch.int64).tolist())
                    src.append(i); tr.append(t)
    # the 8 index maps on a 3x3 and an 8x8 grid (out.flat[i] = x.flat[map[i]])
    maps3 = [fn(torch.arange(9).view(3, 3)).reshape(-1).tolist() for fn in transforms]
    maps8 = [fn(torch.arange(64).view(8, 8)).reshape(-1).tolist() for fn in transforms]
    np.savez_compressed(os.path.join(OUT, "augment.npz"), states=np.array(st, dtype=np.int64),
                        actions=np.array(ac, dtype=np.int64), src=np.array(src), tr=np.array(tr),
                        maps3=np.array(maps3), maps8=np.array(maps8))
    with open(os.path.join(OUT, "csv_meta.json"), "w") as f:
        json.dump({"sha256": hashlib.sha256(raw).hexdigest(), "bytes": len(raw), "rows": 180}, f)
    print("F8", len(st), "augmented rows")


# ---------------------------------------------------------------- F10: opt-in search features over the reference's boards
def gen_f10():
    """self-play games of the twin over the REFERENCE's board objects with the opt-in search features of DESIGN.md 3.9 /
    3.10 (Dirichlet root noise, subtree reuse, both), so that these too are pinned with every env transition
    reference-computed: mcts_twin_features.npz / .json"""
    out, meta = {}, {"cases": []}
    cases = [("ttt", "hash", 1, 30, 3, 0, 7, 0.0, 0.0, True), ("ttt", "uniform", 2, 40, 2, 0, 1, 0.3, 0.25, False),
             ("reversi", "hash", 4, 16, 6, 1, 3, 0.5, 0.25, True), ("reversi", "hash", 9, 20, 8, 1, 0, 0.0, 0.0, True),
             ("reversi6", "hash", 3, 20, 4, 0, 5, 1.0, 0.5, True), ("reversi4", "hash", 1, 40, 4, 0, 2, 0.3, 0.25, True),
             ("reversi4", "hash", 2, 40, 4, 0, 7, 0.0, 0.0, True)]
    for ci, (game, ev, gid, sims, tmv, openings, seed, alpha, eps, reuse) in enumerate(cases):
        tw = Twin(game, ev, dir_alpha=alpha, dir_eps=eps, reuse=reuse)
        ex, w, passes = tw.selfplay(gid, sims, tmv, openings, seed)
        out[f"g{ci}_own"] = np.array([e[0] for e in ex], dtype=np.uint64)
        out[f"g{ci}_pi"] = np.array([e[2] for e in ex], dtype=np.float32)
        out[f"g{ci}_act"] = np.array([e[4] for e in ex], dtype=np.uint8)
        meta["cases"].append({"id": ci, "game": game, "eval": ev, "gid": gid, "sims": sims, "temp_moves": tmv,
                              "openings": openings, "seed": seed, "alpha": alpha, "eps": eps, "reuse": reuse,
                              "winner": int(w), "passes": passes})
    np.savez_compressed(os.path.join(OUT, "mcts_twin_features.npz"), **out)
    with open(os.path.join(OUT, "mcts_twin_features.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("F10", len(cases), "cases; passes:", [c["passes"] for c in meta["cases"]])


# ---------------------------------------------------------------- F11: the other board sizes
def gen_f11():
    """ReversiBoard(size=N) is generic in the reference (reversi_board.py:4-14, :87-88): full random games (F1's row
    format) and arbitrary positions with all move results (F2's format) for sizes 1, 2, 3, 5 and 7."""
    random.seed(4321)
    rows, finals, gid = [], [], 0
    for size, ngames in ((7, 40), (5, 40), (3, 30), (2, 3), (1, 1)):
        for _ in range(ngames):
            b = ReversiBoard(size=size)
            cur, over, turn, passes = 1, False, 0, 0
            while not over:
                moves = b.generate_possible_moves(cur)
                x, o = rev_bits(b.board, 1), rev_bits(b.board, -1)
                if moves:
                    r, c = random.choice(moves)
                    nb = b.make_move(r, c, cur)
                    flips = rev_bits(nb.board, cur) & ~rev_bits(b.board, cur) & ~(1 << (8 * r + c))
                    act = 8 * r + c
                    b = nb
                else:
                    flips, act = 0, 255
                    passes += 1
                over = b.is_game_over()
                rows.append((gid, size, turn, cur + 1, x, o, rev_mask(moves), act, flips, int(over)))
                cur *= -1
                turn += 1
            w, (n1, n2) = b.get_score()
            finals.append((gid, size, w + 1, n1, n2, passes, rev_bits(b.board, 1), rev_bits(b.board, -1)))
            gid += 1
    rng = random.Random(77)
    pos, moves = [], []
    for size in (7, 5, 3, 2, 1):
        for k in range(300 if size > 3 else 120):
            b = ReversiBoard(size=size)
            dens, bias = rng.random(), rng.random()
            for r in range(size):
                for c in range(size):
                    b.board[r][c] = 0 if rng.random() > dens else (1 if rng.random() < bias else -1)
            if k % 7 == 0:
                for i in range(size):
                    for (r, c) in ((i, 0), (i, size - 1), (0, i), (size - 1, i)):
                        if rng.random() < 0.8:
                            b.board[r][c] = rng.choice((1, -1))
            l1, l2 = b.generate_possible_moves(1), b.generate_possible_moves(-1)
            w, (n1, n2) = b.get_score()
            pi = len(pos)
            pos.append((size, rev_bits(b.board, 1), rev_bits(b.board, -1), rev_mask(l1), rev_mask(l2),
                        int(b.is_game_over()), w + 1, n1, n2))
            for player, lst in ((1, l1), (-1, l2)):
                for (r, c) in lst:
                    nb = b.make_move(r, c, player)
                    moves.append((pi, 1 if player == 1 else 0, 8 * r + c, rev_bits(nb.board, 1), rev_bits(nb.board, -1)))
    starts = {str(n): {"board": ReversiBoard(size=n).board.tolist(), "str": str(ReversiBoard(size=n))} for n in range(1, 9)}
    np.savez_compressed(os.path.join(OUT, "reversi_other_sizes.npz"), rows=np.array(rows, dtype=np.uint64),
                        finals=np.array(finals, dtype=np.uint64), pos=np.array(pos, dtype=np.uint64),
                        moves=np.array(moves, dtype=np.uint64))
    with open(os.path.join(OUT, "reversi_start_positions.json"), "w") as f:
        json.dump(starts, f)
    print("F11", len(rows), len(finals), len(pos), len(moves))


# ---------------------------------------------------------------- F12: cells / players outside {-1, 0, +1}
def gen_f12():
    """The reference stores whatever `player` is (tic_tac_toe_board.py:28, reversi_board.py:48) and afterwards treats
    every non-zero cell as occupied (tic_tac_toe_board.py:21, reversi_board.py:26); Reversi plays `player` against
    `-player` whatever the number is (reversi_board.py:34-37).  Boards with such cells, every query answered by the
    reference: TTT valid / moves / game-over / the boards after every legal move; Reversi the same for several
    `player` values."""
    rng = random.Random(2024)
    out = {"ttt": [], "reversi": []}
    t = TicTacToeBoard().make_move(0, 0, 5)          # the VERDICT r4 case
    tb = [t, t.make_move(1, 1, -1), t.make_move(1, 1, -1).make_move(0, 1, 5).make_move(0, 2, 5)]
    for _ in range(60):
        b = TicTacToeBoard()
        for r in range(3):
            for c in range(3):
                b.board[r][c] = rng.choice((0, 0, 0, 1, -1, 5, -5, 2))
        tb.append(b)
    full = TicTacToeBoard(np.array([[5, 1, -1], [-1, 5, 1], [1, -1, 7]]))   # full without a line: (True, 0)
    tb.append(full)
    for b in tb:
        over, w = b.is_game_over()
        mv = b.generate_possible_moves()
        valid = [[bool(b.is_valid_move(r, c)) for c in range(3)] for r in range(3)]
        after = [b.make_move(r, c, 3).board.tolist() for (r, c) in mv[:2]]
        out["ttt"].append({"board": b.board.tolist(), "over": bool(over), "winner": w, "moves": [list(m) for m in mv],
                           "valid": valid, "after_player3": after})
    for size in (8, 6, 5, 4):
        for k in range(25):
            b = ReversiBoard(size=size)
            dens = 0.35 + 0.5 * rng.random()
            for r in range(size):
                for c in range(size):
                    b.board[r][c] = 0 if rng.random() > dens else rng.choice((1, 1, -1, -1, 5, -5, 2))
            ent = {"size": size, "board": b.board.tolist(), "over": bool(b.is_game_over()),
                   "score": [b.get_score()[0], list(b.get_score()[1])], "players": {}}
            for player in (1, -1, 5, -5, 2, 0):
                mv = b.generate_possible_moves(player)
                after = [b.make_move(r, c, player).board.tolist() for (r, c) in mv[:3]]
                ent["players"][str(player)] = {"moves": [list(m) for m in mv], "after": after}
            out["reversi"].append(ent)
    with open(os.path.join(OUT, "off_domain.json"), "w") as f:
        json.dump(out, f)
    print("F12", len(out["ttt"]), len(out["reversi"]))


# ---------------------------------------------------------------- F9: the reference's minimax players
def _load_by_path(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_f9():
    """decisions of the reference's OptimalPlayer classes (the arena's yard-stick):
    Reversi  reversi_players.py:35-77  minimax(board, True, 0) -> (score, best_move) on positions of seeded random
             games, sizes 4/6/8, depths 1..5 (deeper on the small boards), both colours;
    TTT      players.py:30-70          minimax(board, True) on every reachable, unfinished position for the side
             to move.  +-inf scores are stored as +-1000, best_move None as -1."""
    rp = _load_by_path("ref_reversi_players", os.path.join(REF, "src/reversi/players/reversi_players.py"))
    tp = _load_by_path("ref_ttt_players", os.path.join(REF, "src/tic_tac_toe/players.py"))
    rng = random.Random(99)
    rows = []  # size, depth, symbol, x, o, move (8r+c | -1), score
    plan = {4: [(1, 10), (2, 10), (3, 10), (4, 10), (5, 8), (6, 6)], 6: [(1, 12), (2, 12), (3, 10), (4, 6)],
            8: [(1, 14), (2, 12), (3, 10), (4, 4)]}
    for size, lst in plan.items():
        for depth, count in lst:
            got = 0
            while got < count:
                b, cur = ReversiBoard(size=size), 1
                target = rng.randrange(0, size * size - 4)
                for _ in range(target):
                    mv = b.generate_possible_moves(cur)
                    if not mv:
                        cur = -cur
                        mv = b.generate_possible_moves(cur)
                        if not mv:
                            break
                    b = b.make_move(*rng.choice(mv), cur)
                    cur = -cur
                sym = rng.choice((1, -1))
                if b.is_game_over():
                    continue
                score, mv = rp.OptimalPlayer(sym, max_depth=depth).minimax(b, True, 0)
                sc = 1000 if score == float("inf") else (-1000 if score == float("-inf") else int(score))
                rows.append((size, depth, sym + 1, rev_bits(b.board, 1), rev_bits(b.board, -1),
                             255 if mv is None else 8 * mv[0] + mv[1], sc + 2000))
                got += 1
    rev = np.array(rows, dtype=np.uint64)
    # one complete 6x6 game, OptimalPlayer(depth 2) as X against OptimalPlayer(depth 3) as O, through the reference's
    # own turn loop semantics (reversi_terminal.py:16-38)
    random.seed(7)
    pl = {1: rp.OptimalPlayer(1, max_depth=2), -1: rp.OptimalPlayer(-1, max_depth=3)}
    b, cur, seq = ReversiBoard(size=6), 1, []
    over = False
    while not over:
        if b.generate_possible_moves(cur):
            r, c = pl[cur].get_move(b)
            seq.append((cur + 1, 8 * r + c))
            b = b.make_move(r, c, cur)
        over = b.is_game_over()
        cur = -cur
    game = np.array(seq, dtype=np.int64)
    final = np.array([rev_bits(b.board, 1), rev_bits(b.board, -1)], dtype=np.uint64)
    # TTT: every reachable unfinished position, for the side to move
    seen, order = {}, []

    def rec(t, cur):
        key = (ttt_bits(t.board, 1), ttt_bits(t.board, -1))
        if key in seen:
            return
        seen[key] = cur
        order.append(key)
        if t.is_game_over()[0]:
            return
        for mv in t.generate_possible_moves():
            rec(t.make_move(*mv, cur), -cur)
    rec(TicTacToeBoard(), 1)
    trows = []
    memo = {}
    for (x, o) in order:
        cur = seen[(x, o)]
        t = TicTacToeBoard()
        for i in range(9):
            if x >> i & 1:
                t.board[i // 3][i % 3] = 1
            if o >> i & 1:
                t.board[i // 3][i % 3] = -1
        if t.is_game_over()[0] or (x | o) == 0:
            continue
        score, mv = tp.OptimalPlayer(cur).minimax(t, True)
        trows.append((x, o, cur + 1, 3 * mv[0] + mv[1], score + 1))
    ttt = np.array(trows, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "minimax_players.npz"), reversi=rev, game6=game, game6_final=final, ttt=ttt)
    print("F9", rev.shape, game.shape, ttt.shape)


if __name__ == "__main__":
    which = sys.argv[1:] or ["f1", "f2", "f3", "f4", "f5", "f7", "f8", "f9", "f10", "f11", "f12"]
    if "f1" in which: gen_f1()
    if "f2" in which: gen_f2()
    if "f3" in which: gen_f3()
    if "f4" in which: gen_f4()
    if "f5" in which: gen_f5_f6()
    if "f7" in which: gen_f7()
    if "f8" in which: gen_f8()
    if "f9" in which: gen_f9()
    if "f10" in which: gen_f10()
    if "f11" in which: gen_f11()
    if "f12" in which: gen_f12()
