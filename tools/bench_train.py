#!/usr/bin/env python3
"""Where a training step's time goes: ms per Adam step of the az_loop net (64 channels, 4 blocks,
batch 1024) with the residual tower on the hand-written HIP kernels (csrc/bz_train.hip) against stock PyTorch-ROCm
autograd (MIOpen) -- eager / HIP-graph replay, fp32 / bf16 autocast, default / benchmarked MIOpen solvers, NCHW /
channels-last -- plus the three tower kernels on their own.  python tools/bench_train.py [channels] [blocks] [batch] [--quick | --kernels-only]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd.engine import DeviceExamples, Examples  # noqa: E402
from betazero_amd.net import PolicyValueNet  # noqa: E402
from betazero_amd.train import GraphedTrainStep, make_optimizer, train_step  # noqa: E402

QUICK = "--quick" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
C = int(argv[0]) if len(argv) > 0 else 64
NB = int(argv[1]) if len(argv) > 1 else 4
B = int(argv[2]) if len(argv) > 2 else 1024
rng = np.random.default_rng(0)
n = 8 * B
x = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
y = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
pi = rng.random((n, 65)).astype(np.float32); pi /= pi.sum(1, keepdims=True)
ex = DeviceExamples.from_host(Examples(x & ~y, y & ~x, pi, rng.integers(-1, 2, n).astype(np.int8), np.ones(n, np.int8),
                                       np.zeros(n, np.uint8), np.arange(n), np.zeros(n, np.int32), 8))


def run(label, autocast, graph, bench, cl, kernels=False, step_kernels=None, fused_adam=None):
    torch.backends.cudnn.benchmark = bench
    torch.manual_seed(0)
    m = PolicyValueNet(C, NB, 64, fused_tower=kernels).cuda()
    if cl:
        m = m.to(memory_format=torch.channels_last)
    idx = [torch.randint(0, n, (B,), device="cuda:0") for _ in range(8)]
    if graph:
        g = GraphedTrainStep(m, lr=1e-3, batch=B, autocast=autocast, tower_kernels=kernels, step_kernels=step_kernels, fused_adam=fused_adam,
                             capture_autograd=True)   # ("graph" rows of this table time the CAPTURED forms, autograd ones included)
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


def tower_only():
    """the three tower kernels alone (no stem / heads / loss / optimiser): forward with saved activations,
    backward-data, backward-weights, HIP events around 50 launches each"""
    from betazero_amd import _lib
    from betazero_amd.train_kernels import TowerPlan
    L, Ly = _lib.lib(), 2 * NB
    p = TowerPlan(C, Ly, B)
    g = torch.Generator(device="cuda:0").manual_seed(0)
    W = torch.randn((Ly, C, C, 3, 3), device="cuda:0", generator=g) * (1.5 / (9 * C)) ** 0.5
    bias = torch.zeros((Ly, C), device="cuda:0")
    p.acts[0].copy_(torch.relu(torch.randn((B, 64, C), device="cuda:0", generator=g)))
    p.gs[Ly].copy_(torch.randn((B, 64, C), device="cuda:0", generator=g))
    st = torch.cuda.current_stream().cuda_stream
    calls = {
        "pack weights": lambda: L.bz_train_pack_weights(W.data_ptr(), C, Ly, p.wf_fwd.data_ptr(), p.wf_bwd.data_ptr(), st),
        "k_train_fwd": lambda: L.bz_train_tower_fwd(p.acts[0].data_ptr(), p.wf_fwd.data_ptr(), bias.data_ptr(), C, Ly, B, p.acts[1].data_ptr(), p.masks.data_ptr(), st),
        "k_train_bwd": lambda: L.bz_train_tower_bwd(p.gs[Ly].data_ptr(), p.wf_bwd.data_ptr(), p.zeros_c.data_ptr(), p.masks.data_ptr(), C, Ly, B, p.gs[0].data_ptr(), st),
        "k_train_wgrad": lambda: L.bz_train_wgrad(p.acts[0].data_ptr(), p.gs[1].data_ptr(), C, Ly, B, p.splits, p.partial.data_ptr(), p.db_partial.data_ptr(), st)}
    flop = 2 * 64 * 9 * C * C * Ly * B
    for name, fn in calls.items():
        for _ in range(5):
            _lib.check(fn())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(f"  {name:14s} {ms * 1e3:8.1f} us" + ("" if name == "pack weights" else f"  {flop / (ms * 1e-3) / 1e12:7.1f} TFLOP/s"), flush=True)


def ends_only():
    """the kernels of csrc/bz_train_ends.hip alone, HIP events around 50 launches each"""
    import ctypes
    from betazero_amd import _lib
    from betazero_amd.train_kernels import StepPlan
    L = _lib.lib()
    m = PolicyValueNet(C, NB, 64, fused_tower=True).cuda()
    p = StepPlan(m, B)
    Ly, st = p.L, torch.cuda.current_stream().cuda_stream
    idx = torch.randint(0, n, (B,), device="cuda:0")
    p.enable_adam(0.0)
    p.grads(ex.own, ex.opp, ex.pi, ex.z, idx)
    bd = p.batch_desc.data_ptr()
    calls = {
        "k_train_stem": lambda: L.bz_train_stem_fwd(bd, B, m.stem.weight.data_ptr(), m.stem.bias.data_ptr(), C, p.acts[0].data_ptr(), st),
        "k_train_heads": lambda: L.bz_train_heads(p.acts[Ly].data_ptr(), bd, B, C, 64, ctypes.byref(p._head), p.gs[Ly].data_ptr(),
                                                  p.hv.data_ptr(), p.dl.data_ptr(), p.dv1.data_ptr(), p.heads_partial.data_ptr(), st),
        "k_train_stem_wgrad": lambda: L.bz_train_stem_wgrad(bd, p.acts[0].data_ptr(), p.gs[0].data_ptr(), B, C, p.stem_partial.data_ptr(), st),
        "k_train_heads_wgrad": lambda: L.bz_train_heads_wgrad(p.hv.data_ptr(), p.dl.data_ptr(), p.dv1.data_ptr(), B, 64, p.heads_w_partial.data_ptr(), st),
        "k_train_finish": lambda: L.bz_train_finish(ctypes.byref(p._partials), ctypes.byref(p._grads), C, Ly, 64, B, p.losses.data_ptr(), None, st),
        "k_train_finish + k_train_adam": lambda: L.bz_train_finish(ctypes.byref(p._partials), ctypes.byref(p._grads), C, Ly, 64, B, p.losses.data_ptr(), ctypes.byref(p._adam), st)}
    print("the ends of the step alone (csrc/bz_train_ends.hip):")
    for name, fn in calls.items():
        for _ in range(5):
            _lib.check(fn())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"  {name:20s} {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us", flush=True)


print(f"net {C} channels x {NB} blocks, batch {B}")
print("tower kernels alone (csrc/bz_train.hip):")
tower_only()
if "--kernels-only" in sys.argv:
    sys.exit(0)
ends_only()
run("bf16 graph  whole step on HIP kernels incl. Adam (10 launches)", True, True, False, False, kernels=True, step_kernels=True)
run("bf16 graph  whole step on HIP kernels + torch's fused Adam", True, True, False, False, kernels=True, step_kernels=True, fused_adam=False)
run("bf16 graph  HIP tower kernels inside torch autograd", True, True, False, False, kernels=True, step_kernels=False)
run("bf16 autocast graph  miopen-default    nchw", True, True, False, False)
if not QUICK:
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
