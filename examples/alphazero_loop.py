#!/usr/bin/env python3
"""A complete AlphaZero iteration loop on one MI355X with betazero_amd:
self-play (GPU MCTS + bf16 MFMA net) -> 8-fold augmentation + dedupe (GPU) -> a few Adam steps
(PyTorch autograd) -> push the weights back into the engine -> repeat.

    python examples/alphazero_loop.py --games 256 --sims 64 --iters 2
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd.augment import augment_examples  # noqa: E402
from betazero_amd.engine import SelfPlayEngine  # noqa: E402
from betazero_amd.examples_io import save_examples_csv  # noqa: E402
from betazero_amd.net import DeviceNet, PolicyValueNet  # noqa: E402
from betazero_amd.train import make_optimizer, refresh_device_net, train_step  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=256)
    ap.add_argument("--sims", type=int, default=64)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--train-steps", type=int, default=20)
    ap.add_argument("--csv", default=None, help="write the last iteration's examples (State,Action,Pi,Z)")
    ap.add_argument("--noise", type=float, default=0.0, help="Dirichlet root-noise weight eps (alpha 0.5); 0 = off")
    ap.add_argument("--reuse", action="store_true", help="keep the chosen child's subtree between moves")
    ap.add_argument("--arena", type=int, default=0, help="after training: this many arena games vs the reference's "
                                                         "depth-2 minimax player")
    args = ap.parse_args()
    torch.manual_seed(0)
    module = PolicyValueNet(128, 6, 64).round_to_bf16_()
    net = DeviceNet.from_module(module, args.games)
    opt = make_optimizer(module, lr=1e-3)
    rng = np.random.default_rng(0)
    for it in range(args.iters):
        t0 = time.time()
        eng = SelfPlayEngine("reversi", args.games, args.sims, "net_bf16", net, temp_moves=8, openings=1, seed=it,
                             game_id_base=it * args.games, dirichlet_alpha=0.5 if args.noise > 0 else 0.0,
                             dirichlet_eps=args.noise, reuse_subtree=args.reuse)
        plies = eng.run_iteration()
        ex = eng.examples()
        winners, _ = eng.winners()
        t1 = time.time()
        ex8 = augment_examples(ex)
        losses = []
        for _ in range(args.train_steps):
            idx = rng.choice(len(ex8), size=min(args.batch, len(ex8)), replace=False)
            losses.append(train_step(module, opt, ex8, idx))
        refresh_device_net(net, module)
        t2 = time.time()
        w = winners[0]
        print(f"iter {it}: {args.games} games / {plies} plies in {t1 - t0:.1f} s ({args.games / (t1 - t0):.1f} games/s), "
              f"X/O/draw = {(w == 1).sum()}/{(w == -1).sum()}/{(w == 0).sum()}, {len(ex)} -> {len(ex8)} examples, "
              f"loss {losses[0][0]:.3f} -> {losses[-1][0]:.3f} (CE {losses[-1][1]:.3f}, MSE {losses[-1][2]:.3f}), "
              f"train+refresh {t2 - t1:.1f} s")
        del eng
    if args.csv:
        save_examples_csv(ex8, args.csv)
        print("wrote", args.csv)
    if args.arena:
        from betazero_amd.arena import play_arena
        res = play_arena("reversi", args.arena, args.sims, opponent_depth=2, evaluator="net_bf16", net=net)
        print("arena vs OptimalPlayer(max_depth=2):", res.summary())


if __name__ == "__main__":
    main()
