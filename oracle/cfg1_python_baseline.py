#!/usr/bin/env python3
"""BASELINE config 1 (plumbing, CPU, build container only): the build-authored Python MCTS twin
(oracle/gen_golden.py Twin) over the IMPORTED reference TicTacToeBoard, 25 sims/move, uniform
priors, seed 0, 100 games, 1 core.  There is no reference MCTS loop to time; this is
the closest thing to "the reference Python CPU path".  Also times a Reversi sample (800 sims,
hash priors, 2 moves) for the cfg-3 yard-stick of BASELINE.md section 2."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402  (imports the reference by path)

t0 = time.time()
tw = gg.Twin("ttt", "uniform")
n, plies, res = 100, 0, {1: 0, -1: 0, 0: 0}
for g in range(n):
    ex, w, _ = tw.selfplay(g, 25, 0, 0, 0)
    plies += len(ex)
    res[w] += 1
dt = time.time() - t0
print(f"cfg 1: {n} TTT games, 25 sims/move, Python twin over the reference boards, 1 core: {n / dt:.2f} games/s "
      f"({plies / n:.1f} plies/game, results +1/-1/0 = {res[1]}/{res[-1]}/{res[0]})")

import random  # noqa: E402
from reversi_board import ReversiBoard  # noqa: E402
tw = gg.Twin("reversi", "hash")
t0 = time.time()
root = tw.search(ReversiBoard(), 1, 800)
dt = time.time() - t0
print(f"cfg 3 yard-stick: one 800-sim search from the Reversi start position over the reference ReversiBoard "
      f"(hash priors, no net): {dt:.2f} s -> {1 / dt / 58:.4f} games/s per core at 58 searched moves per game, before any net")
