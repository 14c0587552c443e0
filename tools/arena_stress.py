import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from betazero_amd.arena import play_arena
from betazero_amd.net import DeviceNet, PolicyValueNet
torch.manual_seed(0)
for seed in range(6):
    for ev, depth, sims in (("uniform", 2, 16), ("hash", 3, 24)):
        try:
            r = play_arena("reversi", 2048, sims, opponent_depth=depth, evaluator=ev, seed=seed, opening_plies=6 + seed)
            print(seed, ev, r.summary(), flush=True)
        except RuntimeError as e:
            print("FAIL", seed, ev, e, flush=True)
