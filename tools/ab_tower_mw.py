#!/usr/bin/env python3
"""A/B of the bf16 tower's wave split (Split<MW>: MW = 1 -> 32 co x 4 positions per wave, MW = 2 -> 64 co x 2
positions per wave) in ONE process on ONE device, interleaved rounds on random (non-zero) data, as
cdna_hip_programming.md rule 24 asks.  Also checks that the two splits give bit-identical outputs (the k order of
every output element is the same).  python tools/ab_tower_mw.py [batch] [iters] [rounds]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd import _lib  # noqa: E402
from betazero_amd.net import DeviceNet, PolicyValueNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.manual_seed(0)
mod = PolicyValueNet(128, 6, 64).round_to_bf16_()
nets = {}
for mw in (1, 2):
    os.environ["BZ_TOWER_MW"] = str(mw)
    nets[mw] = DeviceNet.from_module(mod, B)
rng = np.random.default_rng(0)
x = rng.integers(0, 2**63, size=B, dtype=np.int64)
y = rng.integers(0, 2**63, size=B, dtype=np.int64)
own = torch.as_tensor(x & ~y).cuda()
opp = torch.as_tensor(y & ~x).cuda()
out = {mw: nets[mw].forward(own, opp) for mw in (1, 2)}
torch.cuda.synchronize()
same = torch.equal(out[1][0], out[2][0]) and torch.equal(out[1][1], out[2][1])
print("outputs bit-identical between the two splits:", same)
times = {1: [], 2: []}
for r in range(rounds):
    for mw in (1, 2):
        for _ in range(10):
            nets[mw].forward(own, opp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            nets[mw].forward(own, opp)
        e1.record()
        torch.cuda.synchronize()
        times[mw].append(e0.elapsed_time(e1) / iters * 1e3)
for mw in (1, 2):
    t = np.array(times[mw])
    print(f"MW={mw}: us per forward at batch {B}: median {np.median(t):.1f} min {t.min():.1f} max {t.max():.1f}  "
          f"-> {B * 226.86e6 / np.median(t) / 1e6:.0f} TFLOP/s")
sys.exit(0 if same else 1)
