#!/usr/bin/env python3
"""tower time vs number of residual blocks -> fixed cost and per-layer cost"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd import _lib
from betazero_amd.net import DeviceNet, PolicyValueNet
B = 4096
rng = np.random.default_rng(0)
x = rng.integers(0, 2**63, size=B, dtype=np.int64); y = rng.integers(0, 2**63, size=B, dtype=np.int64)
own = torch.as_tensor(x & ~y).cuda(); opp = torch.as_tensor(y & ~x).cuda()
L = _lib.lib()
res = {}
for NB in (1, 3, 6, 12):
    torch.manual_seed(0)
    net = DeviceNet.from_module(PolicyValueNet(128, NB, 64).round_to_bf16_(), B)
    for _ in range(3): net.forward(own, opp)
    L.bz_profile_reset(); L.bz_profile_enable(1)
    for _ in range(50): net.forward(own, opp)
    L.bz_profile_enable(0)
    n, t, ms = _lib.profile_read()["tower"]
    res[NB] = ms / t * 1e3
    print(f"NB={NB:2d} layers={2*NB:2d} tower {res[NB]:8.1f} us")
per = (res[12] - res[3]) / 18
print(f"per-layer {per:.2f} us (ideal 36.9 us @2.0GHz), fixed {res[6]-12*per:.1f} us")
