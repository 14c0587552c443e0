#!/usr/bin/env python3
"""Are all torch streams equal?  bench.py's two pipelines on the k-th pair of streams that torch hands out (it takes
them round-robin from a pool of 32), one process, same workload (cfg 3, 4 steps each).  Found while chasing a "second run
in a process is 12 % slow" effect: it was the pool's second pair.  python tools/exp_streams.py"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


class C:
    pass


ctx = C(); ctx.rank = 0; ctx.world = 1; ctx.backend = "nccl"; ctx.dev = "cuda:0"; ctx.local = 0
torch.cuda.set_device(0)
ctx.sync = torch.cuda.synchronize
ctx.barrier = torch.cuda.synchronize
args = argparse.Namespace(precision=None, mode="steady", streams=2, reuse_subtree=False, dirichlet_eps=0.0, no_kernel_timers=False)
for pair in range(6):
    st = [torch.cuda.Stream(device="cuda:0") for _ in range(2)]
    bench._STREAMS[("cuda:0", 2)] = st
    r = bench.run_reversi(ctx, args, 4096, 800, 4, 1)
    print(f"pool pair {pair} (cuda_stream {[hex(s.cuda_stream) for s in st]}): {r['value']:.1f} games/s, tower frac "
          f"{r['roofline']['frac']:.4f}, {r['ms_per_step']:.1f} ms per step", flush=True)
