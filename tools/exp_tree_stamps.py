#!/usr/bin/env python3
"""Where a k_tree_step launch spends its time: shader-clock stamps between the kernel's phases (diagnostic variant
-DBZ_EXP_TREE_STAMPS, built by betazero_amd.build.build_variant into build/variants/ and loaded through BZ_HIP_SO;
the product library is untouched).  cfg-3 shape: 4096 Reversi games, the bf16 net, a staggered pool.
usage (GPU box):  BZ_HIP_SO=build/variants/libbz_hip.treestamps.so BZ_ALLOW_EXPERIMENT=1 python tools/exp_tree_stamps.py [sims]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from betazero_amd.engine import SelfPlayEngine  # noqa: E402
from betazero_amd.net import DeviceNet, PolicyValueNet  # noqa: E402

sims = int(sys.argv[1]) if len(sys.argv) > 1 else 800
B = int(os.environ.get("GAMES", "4096"))
torch.manual_seed(0)
net = DeviceNet.from_module(PolicyValueNet(128, 6, 64).round_to_bf16_(), B)
CACHE = os.environ.get("EVAL_CACHE", "1") != "0"   # EVAL_CACHE=0: the engine's evaluation cache off (what the lookup costs the tree step)
eng = SelfPlayEngine("reversi", B, sims, "net_bf16", net, temp_moves=8, openings=1, rounds=4, stagger=58, eval_cache=CACHE)
eng.reset_games()
eng.search(); eng.play(True)          # warm-up move
torch.cuda.synchronize()
eng.reset_counters()
eng.search(); eng.play(True)
torch.cuda.synchronize()
eng._call(__import__("betazero_amd._lib", fromlist=["lib"]).lib().bz_engine_sum_counters)
c = eng._view(eng.lay.counters, torch.int64, (24,)).cpu().numpy()   # work counters 0..8, stamps 16..22, waves 23
waves = max(int(c[23]), 1)
launches = sims + 1
names = ["T0: per-game words + path (1 round trip)", "T1: evaluator row arrives", "expansion (softmax, edge stores) + backup stores",
         "select walk (all levels)", "child creation (apply / legal / terminal)", "tail (leaf words, slot atomic, path flush)",
         "work-counter flush"]
tot = 0
print(f"evaluation cache {'on' if CACHE else 'off'}: {c[8]} of {c[7] + c[8]} evaluations shared")
print(f"k_tree_step stamps: {waves} wave-executions over {launches} launches ({waves / launches:.0f} per launch), B = {B}, sims = {sims}")
for k, nm in enumerate(names):
    v = c[16 + k] / waves
    tot += v
    print(f"  {nm:60s} {v:9.0f} cycles per wave")
print(f"  {'sum of stamped phases':60s} {tot:9.0f} cycles per wave")
print(f"  mean path nodes per simulation: {c[1] / max(c[0], 1):.2f}")
