#!/usr/bin/env python3
"""Are all torch streams equal?  The two pipelines of PipelinedSelfPlay on the k-th pair of streams torch hands out
(round-robin from a pool), one process, same workload (cfg 3, 4 steps each) -- next to what bz_stream_overlap_probe
says about that pair (two 0.3-ms single-wave kernels behind a common event: ~1.0 = they overlap, ~2.0 = they
serialise).  Round 3 found the pool's second pair 12 % slow; this is the evidence that the probe sees it, and the
last lines show six constructions through pipeline_streams() (which picks its pair by the probe).
python tools/exp_streams.py [--pairs 6] [--steps 4]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from betazero_amd.engine import PipelinedSelfPlay, pipeline_stream_info, pipeline_streams, stream_overlap_ratio  # noqa: E402
from betazero_amd.net import DeviceNet, PolicyValueNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=6)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--games", type=int, default=4096)
ap.add_argument("--sims", type=int, default=800)
args = ap.parse_args()
torch.cuda.set_device(0)
torch.manual_seed(0)
net = DeviceNet.from_module(PolicyValueNet(128, 6, 64).round_to_bf16_(), args.games)


def run(streams):
    sp = PipelinedSelfPlay("reversi", args.games, args.sims, "net_bf16", net, pipelines=2, streams=streams, temp_moves=8,
                           openings=1, seed=0, rounds=2, stagger=58)
    sp.reset_games()
    sp.step(True)
    sp.sync()
    f0 = sp.status()[1]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sp.step(True)
    sp.sync()
    dt = time.perf_counter() - t0
    return (sp.status()[1] - f0) / dt, dt / args.steps * 1e3


for pair in range(args.pairs):
    st = [torch.cuda.Stream(device="cuda:0") for _ in range(2)]
    ratio = stream_overlap_ratio(st[0], st[1])
    ratio_long = stream_overlap_ratio(st[0], st[1], spin_us=2000, reps=2)
    gps, ms = run(st)
    print(f"pool pair {pair} (cuda_stream {[hex(s.cuda_stream) for s in st]}): probe {ratio:.3f} (0.3 ms) / {ratio_long:.3f} (2 ms)  "
          f"{gps:.1f} games/s, {ms:.1f} ms per step", flush=True)
print("six constructions through pipeline_streams() (pair picked by the probe, created once per process):")
for k in range(6):
    gps, ms = run(None)
    print(f"  construction {k}: {gps:.1f} games/s, {ms:.1f} ms per step   probe info {pipeline_stream_info('cuda:0', 2)}", flush=True)
st = pipeline_streams("cuda:0", 2)
print("picked streams", [hex(s.cuda_stream) for s in st])
