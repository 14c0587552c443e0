#!/usr/bin/env python3
"""captured autograd steps returned wrong gradients once in a few hundred replays -- a bias gradient (a column sum
over 65,536 rows, torch's multi-block reduction) of 6e32 at replay 305 / 306, an all-zero one at replay 702 / 703, the replay numbers
repeating across seeds (profiles/r04_channels_last_cause.txt).  This isolates the suspect: ONE captured graph holding a few
elementwise kernels and that column sum over a STATIC input, replayed thousands of times without host synchronisation; every replay's
result is copied (outside the graph) into a row of a device log and compared with the expected constant at the end.

    python tools/exp_graph_reduction.py [replays] [rows] [cols]

Variants: `graph` (replays back to back), `graph+sync` (a stream synchronise after every replay), `eager` (the same ops launched
one by one), `graph+alloc` (a fresh allocation + free between replays, as a training loop's `out.clone()` does), `2graphs` (two
captured graphs replayed alternately on the same private pool inputs).  Prints the first replay whose result differs, if any."""
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
COLS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = "cuda:0"
torch.manual_seed(0)
x = torch.randn((ROWS, COLS), device=dev, dtype=torch.bfloat16)
scale = torch.ones((), device=dev)


MODE = "both"


def body(out):
    y = (x.float() * scale).to(torch.bfloat16)          # a cast and an elementwise op in front, like autograd's
    if MODE in ("both", "bf16"):
        g = y.sum(0)                                    # the multi-block column reduction (bf16 in, as under autocast)
        out[0].copy_(g.float())
    if MODE in ("both", "f32"):
        g2 = y.float().sum(0)
        out[1].copy_(g2)


def run(name, sync=False, graph=True, alloc=False, two=False, zero=True, nap=0.0):
    out = torch.zeros((2, COLS), device=dev)
    log = torch.zeros((N, 2, COLS), device=dev)
    body(out)
    torch.cuda.synchronize()
    want = out.clone()
    gs = []
    if graph:
        for _ in range(2 if two else 1):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    body(out)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body(out)
            gs.append(g)
    t0 = time.time()
    for i in range(N):
        if zero:
            out.zero_()
        if graph:
            gs[i % len(gs)].replay()
        else:
            body(out)
        log[i].copy_(out)
        if alloc:
            tmp = torch.empty((1 << 20,), device=dev)
            tmp.fill_(1.0)
            del tmp
        if sync:
            torch.cuda.synchronize()
        if nap:
            time.sleep(nap)
    torch.cuda.synchronize()
    dt = time.time() - t0
    bad = (log != want[None]).flatten(1).any(1).nonzero().flatten().tolist()
    print(f"{name:14s} {N} replays in {dt:.2f} s: {len(bad)} differ" + (f"; first at replay {bad[0]}: got {log[bad[0]].tolist()} want {want.tolist()}; all: {bad[:20]}" if bad else ""), flush=True)


print("torch", torch.__version__, "hip", torch.version.hip, "rows", ROWS, "cols", COLS, flush=True)
run("graph")
run("graph+sync", sync=True)
run("eager", graph=False)
run("eager+sync", graph=False, sync=True)
run("graph+alloc", alloc=True)
run("2graphs", two=True)
run("graph+sync/nz", sync=True, zero=False)
run("graph+nap", nap=0.0005)
for MODE in ("bf16", "f32"):
    run("sync " + MODE, sync=True)
    run("alloc " + MODE, alloc=True)
MODE = "both"
run("graph again")


# ---- the suspect by itself: a captured hipMemsetAsync in front of kernels that read and then dirty the buffer (what torch's
# multi-block reduction does with its semaphores: Reduce.cuh memsets them on the launch stream, i.e. as a MEMSET NODE under
# capture).  Every replay must see zeros.
def memset_probe(sync, eager_between):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    buf = torch.full((64,), 7, device=dev, dtype=torch.int32)
    seen = torch.zeros((64,), device=dev, dtype=torch.int32)
    other = torch.zeros((4,), device=dev)
    log = torch.zeros((N, 64), device=dev, dtype=torch.int32)

    def step():
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, buf.numel() * 4, st)
        assert rc == 0, rc
        seen.copy_(buf)          # must be all zeros
        buf.add_(7)              # dirty it again (the reduction leaves its semaphores at gridDim.y)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for i in range(N):
        if eager_between:
            other.zero_()
        g.replay()
        log[i].copy_(seen)
        if sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    bad = (log != 0).any(1).nonzero().flatten().tolist()
    print(f"memset node, sync={sync}, eager kernel between replays={eager_between}: {len(bad)} of {N} replays saw a dirty buffer"
          + (f"; first {bad[0]}: {log[bad[0]][:4].tolist()}" if bad else ""), flush=True)


for sync in (False, True):
    for eb in (False, True):
        memset_probe(sync, eb)
