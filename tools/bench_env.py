#!/usr/bin/env python3
"""HBM-roofline microbenchmark of the batched board-env step kernels (42 algorithmic bytes per
Reversi step: 17 in, 25 out).  python tools/bench_env.py [n_games]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
L = _lib.lib()
_lib.require_gpu()
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randint(0, 2**62, (n,), generator=g, device="cuda", dtype=torch.int64)
b = torch.randint(0, 2**62, (n,), generator=g, device="cuda", dtype=torch.int64)
own, opp = a & ~b, b & ~a
legal = torch.empty(n, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
_lib.check(L.bz_reversi_legal_batch(own.data_ptr(), opp.data_ptr(), n, legal.data_ptr(), st))
# play the lowest legal move of every game (pass where there is none)
lg = legal.cpu().numpy().view(np.uint64)
low = (lg & (~lg + np.uint64(1))).astype(np.float64)
act = np.where(lg == 0, 64, np.log2(np.maximum(low, 1)).astype(np.int64)).astype(np.uint8)
action = torch.as_tensor(act).cuda()
on, pn, ln = (torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(3))
status = torch.empty(n, dtype=torch.uint8, device="cuda")
winner = torch.empty(n, dtype=torch.int8, device="cuda")


def step():
    _lib.check(L.bz_reversi_step_batch(own.data_ptr(), opp.data_ptr(), action.data_ptr(), n, on.data_ptr(), pn.data_ptr(),
                                       ln.data_ptr(), status.data_ptr(), winner.data_ptr(), st))


for _ in range(3):
    step()
L.bz_profile_reset(); L.bz_profile_enable(1)
for _ in range(20):
    step()
L.bz_profile_enable(0)
_, t, ms = _lib.profile_read()["env_step"]
us = ms / t * 1e3
gbs = 42.0 * n / (us * 1e-6) / 1e9
print(f"k_reversi_step: {n} games, {us:.1f} us per launch, {n / us:.0f} M steps/s, {gbs:.0f} GB/s algorithmic = "
      f"{gbs / 8000 * 100:.1f} % of 8 TB/s HBM peak; illegal={(status == 2).sum().item()}")
