#!/usr/bin/env python3
"""Latency of the Player.get_move plug-in (a12): one MCTS search of `sims` simulations from the Reversi start position
with the bf16 MFMA net in the loop, batch 1 (an interactive MCTSPlayer) and batch 64 (the arena).
python tools/bench_player.py [sims]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import betazero_amd as bz  # noqa: E402
from betazero_amd.engine import SelfPlayEngine  # noqa: E402
from betazero_amd.net import DeviceNet, PolicyValueNet  # noqa: E402

sims = int(sys.argv[1]) if len(sys.argv) > 1 else 800
torch.manual_seed(0)
net = DeviceNet.from_module(PolicyValueNet(128, 6, 64).round_to_bf16_(), 64)
pl = bz.MCTSPlayer(1, sims=sims, net=net)
b = bz.ReversiBoard()
pl.get_move(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    mv = pl.get_move(b)
dt = (time.perf_counter() - t0) / 5
print(f"MCTSPlayer.get_move, {sims} sims, batch 1, the SAME position again (every evaluation is found in the previous search's "
      f"arena, no net forward runs): {dt * 1e3:.1f} ms per move ({dt / sims * 1e6:.1f} us per simulation), move {mv}")
# the representative number: two players (own engines) play a game; every search starts two plies after the player's last one
players = {1: bz.MCTSPlayer(1, sims=sims, net=net), -1: bz.MCTSPlayer(-1, sims=sims, net=net)}
b, side, times = bz.ReversiBoard(), 1, []
for ply in range(24):
    if b.is_game_over():
        break
    if not b.generate_possible_moves(side):
        side = -side
        continue
    t0 = time.perf_counter()
    mv = players[side].get_move(b)
    times.append(time.perf_counter() - t0)
    b = b.make_move(mv[0], mv[1], side)
    side = -side
times = times[2:]  # each player's first search builds its engine
dt = sum(times) / len(times)
print(f"MCTSPlayer.get_move, {sims} sims, batch 1, along a game ({len(times)} moves, two players): {dt * 1e3:.1f} ms per move "
      f"({dt / sims * 1e6:.1f} us per simulation), min {min(times) * 1e3:.1f} max {max(times) * 1e3:.1f} ms")
eng = SelfPlayEngine("reversi", 64, sims, "net_bf16", net, temp_moves=8, openings=1)
eng.reset_games(); eng.search(); eng.play(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):   # search + move, so that every search starts from a new position (one ply after the last)
    eng.search()
    eng.play()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 6
print(f"engine search + move, {sims} sims, batch 64 (the arena's shape): {dt * 1e3:.1f} ms per move for all 64 games "
      f"({dt / sims * 1e6:.1f} us per simulation)")
