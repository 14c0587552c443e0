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
print(f"MCTSPlayer.get_move, {sims} sims, batch 1: {dt * 1e3:.1f} ms per move ({dt / sims * 1e6:.1f} us per simulation), move {mv}")
eng = SelfPlayEngine("reversi", 64, sims, "net_bf16", net, temp_moves=8, openings=1)
eng.reset_games(); eng.search(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng.search()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"engine search, {sims} sims, batch 64: {dt * 1e3:.1f} ms per move for all 64 games ({dt / sims * 1e6:.1f} us per simulation)")
