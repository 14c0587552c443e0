#!/usr/bin/env python3
"""One training configuration under rocprofv3 --kernel-trace: 200 replays of the captured step (the tower on the HIP
kernels), so that tools/summarize_prof.py's per-kernel table shows what a step is made of.
rocprofv3 --kernel-trace --stats -d gpurun_out/r04prof/train_step -o t -- python3 tools/prof_train_step.py [C] [NB] [batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd.engine import DeviceExamples, Examples  # noqa: E402
from betazero_amd.net import PolicyValueNet  # noqa: E402
from betazero_amd.train import GraphedTrainStep  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
rng = np.random.default_rng(0)
n = 4 * B
x = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
y = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
pi = rng.random((n, 65)).astype(np.float32); pi /= pi.sum(1, keepdims=True)
ex = DeviceExamples.from_host(Examples(x & ~y, y & ~x, pi, rng.integers(-1, 2, n).astype(np.int8), np.ones(n, np.int8),
                                       np.zeros(n, np.uint8), np.arange(n), np.zeros(n, np.int32), 8))
torch.manual_seed(0)
g = GraphedTrainStep(PolicyValueNet(C, NB, 64, fused_tower=True).cuda(), lr=1e-3, batch=B)
idx = torch.randint(0, n, (B,), device="cuda:0")
for _ in range(200):
    out = g(ex, idx)
torch.cuda.synchronize()
print("loss", out.tolist())
