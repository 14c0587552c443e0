#!/usr/bin/env python3
"""Where a training step's time goes (stock PyTorch-ROCm autograd): ms per Adam step of the
az_loop net (64 channels, 4 blocks, batch 1024) for eager / HIP-graph replay, fp32 / bf16 autocast, default /
benchmarked MIOpen solvers, NCHW / channels-last.  python tools/bench_train.py [channels] [blocks] [batch]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd.engine import DeviceExamples, Examples  # noqa: E402
from betazero_amd.net import PolicyValueNet  # noqa: E402
from betazero_amd.train import GraphedTrainStep, make_optimizer, train_step  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
rng = np.random.default_rng(0)
n = 8 * B
x = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
y = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
pi = rng.random((n, 65)).astype(np.float32); pi /= pi.sum(1, keepdims=True)
ex = DeviceExamples.from_host(Examples(x & ~y, y & ~x, pi, rng.integers(-1, 2, n).astype(np.int8), np.ones(n, np.int8),
                                       np.zeros(n, np.uint8), np.arange(n), np.zeros(n, np.int32), 8))


def run(label, autocast, graph, bench, cl):
    torch.backends.cudnn.benchmark = bench
    torch.manual_seed(0)
    m = PolicyValueNet(C, NB, 64).cuda()
    if cl:
        m = m.to(memory_format=torch.channels_last)
    idx = [torch.randint(0, n, (B,), device="cuda:0") for _ in range(8)]
    if graph:
        g = GraphedTrainStep(m, lr=1e-3, batch=B, autocast=autocast)
        step = lambda i: g(ex, idx[i % 8])  # noqa: E731
    else:
        opt = make_optimizer(m, lr=1e-3)
        step = lambda i: train_step(m, opt, ex, idx[i % 8], autocast=autocast)  # noqa: E731
    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    t0 = time.time()
    K = 100
    for i in range(K):
        out = step(i)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / K
    flop = 3 * (2 * 64 * 9 * C * C * 2 * NB + 2 * 64 * 9 * 2 * C) * B
    print(f"{label:58s} {dt * 1e3:7.3f} ms/step  {flop / dt / 1e12:6.1f} TFLOP/s (fwd+bwd convs)  loss {float(out[0]):.4f}", flush=True)


print(f"net {C} channels x {NB} blocks, batch {B}")
for bench in (False, True):
    for cl in (False, True):
        for autocast in (False, True):
            for graph in (False, True):
                try:
                    run(f"{'bf16 autocast' if autocast else 'fp32':14s} {'graph' if graph else 'eager':6s} "
                        f"{'miopen-benchmark' if bench else 'miopen-default':17s} {'channels_last' if cl else 'nchw'}",
                        autocast, graph, bench, cl)
                except Exception as e:
                    print("FAILED", autocast, graph, bench, cl, repr(e)[:200], flush=True)
