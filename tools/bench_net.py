#!/usr/bin/env python3
"""Micro-benchmark of the bf16 net forward (stem / tower / heads) on one GPU,
using the in-library HIP-event timers.  python tools/bench_net.py [batch] [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd import _lib  # noqa: E402
from betazero_amd.net import DeviceNet, PolicyValueNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fp8 = len(sys.argv) > 3 and sys.argv[3] == "fp8"
torch.manual_seed(0)
mod = PolicyValueNet(128, 6, 64).round_to_bf16_()
net = DeviceNet.from_module(mod, B)
rng = np.random.default_rng(0)
x = rng.integers(0, 2**63, size=B, dtype=np.int64)
y = rng.integers(0, 2**63, size=B, dtype=np.int64)
own = torch.as_tensor(x & ~y).cuda()
opp = torch.as_tensor(y & ~x).cuda()
for _ in range(5):
    net.forward(own, opp, fp8=fp8)
L = _lib.lib()
L.bz_profile_reset(); L.bz_profile_enable(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    net.forward(own, opp, fp8=fp8)
e1.record(); torch.cuda.synchronize()
L.bz_profile_enable(0)
pr = _lib.profile_read()
tot = e0.elapsed_time(e1) / iters
print(("fp8 " if fp8 else "bf16 ") + f"batch {B}: forward {tot*1e3:.1f} us  ({B*226.86e6/tot/1e9:.1f} TFLOP/s end-to-end)")
for k in ("stem", "tower", "heads"):
    n, t, ms = pr[k]
    print(f"  {k:6s} {ms/max(t,1)*1e3:8.1f} us" + (f"  {B*226.49e6/(ms/t)/1e9:.1f} TFLOP/s" if k == "tower" else ""))
