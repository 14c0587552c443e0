#!/usr/bin/env python3
"""The ten kernels of a training step launched eagerly, a few steps (no graph, no torch kernel besides the set-up): the
target of `rocprofv3 --kernel-trace --pmc ...` passes (tools/prof_train_pmc.sh).
python3 tools/prof_train_kernels.py [C] [blocks] [batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd.net import PolicyValueNet  # noqa: E402
from betazero_amd.train_kernels import StepPlan  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
torch.manual_seed(0)
m = PolicyValueNet(C, NB, 64, fused_tower=True).cuda()
p = StepPlan(m, B)
rng = np.random.default_rng(0)
n = 4 * B
x = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
y = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
pi = rng.random((n, 65)).astype(np.float32)
pi /= pi.sum(1, keepdims=True)
t = lambda a: torch.as_tensor(a).cuda()  # noqa: E731
own, opp = t((x & ~y).view(np.int64)), t((y & ~x).view(np.int64))
idx = torch.randint(0, n, (B,), device="cuda:0")
p.set_batch(own, opp, t(pi), t(rng.integers(-1, 2, n).astype(np.int8)), idx)
p.enable_adam(1e-4)
for _ in range(12):
    out = p.step()
torch.cuda.synchronize()
print("done", C, NB, B, out.tolist(), flush=True)
