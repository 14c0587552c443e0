#!/usr/bin/env python3
"""The closed AlphaZero loop on one GPU, end to end through the package's public pieces:

    self-play (PipelinedSelfPlay: two engines on two streams, bf16 MFMA net in the loop, Dirichlet root noise, temperature moves)
      -> example block -> 80 / 20 hold-out split of the new rows (the reference's split, SL/train.py:66-78)
      -> 8-fold D4 augmentation + exact dedupe of the training part on the device (augment_examples)
      -> policy cross-entropy + value MSE steps on the hand-written training kernels replayed as one HIP graph
         (GraphedTrainStep: stem, tower, heads, losses, every gradient and Adam as ten launches; --miopen-train /
         --eager-train / --fp32-train fall back to stock PyTorch autograd, run eagerly)
         -- the rows never leave the GPU between the engine's example block and the optimiser step
      -> validation on the held-out rows with the engine's own bf16 MFMA forward (validate: policy CE, value MSE, top-1
         agreement -- the reference's per-epoch validation line, SL/train.py:121-146)
      -> weights pushed back into the engine's net (refresh_device_net)
      -> batched arena against the reference's depth-limited minimax player (play_arena)

It prints one JSON line per iteration and a final summary; `--out` keeps them.  The yard-stick is the reference's
OptimalPlayer (src/reversi/players/reversi_players.py:35-77, stone-difference minimax) at `--depth`; the same arena
with the uniform evaluator (no net) is printed first as the untrained reference point.

    python tools/az_loop.py --iters 8            # a few minutes on one MI355X
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd.arena import play_arena  # noqa: E402
from betazero_amd.augment import augment_examples  # noqa: E402
from betazero_amd.engine import PipelinedSelfPlay, concat_device_examples  # noqa: E402
from betazero_amd.net import DeviceNet, PolicyValueNet  # noqa: E402
from betazero_amd.train import (GraphedTrainStep, holdout_split, make_optimizer, refresh_device_net, select_rows,  # noqa: E402
                                train_step, validate)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=64, choices=(64, 128, 256))
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--games", type=int, default=2048, help="concurrent self-play games per iteration")
    ap.add_argument("--sims", type=int, default=64)
    ap.add_argument("--pipelines", type=int, default=2, help="self-play pipelines on separate HIP streams (PipelinedSelfPlay)")
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--window", type=int, default=3, help="iterations of examples kept for training")
    ap.add_argument("--epochs", type=float, default=1.0, help="passes over the (augmented) window per iteration")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--lr-warmup", type=int, default=300, help="steps over which the learning rate ramps up (GraphedTrainStep.lr_warmup_steps): "
                    "without it Adam's first steps at this lr can kill the head ReLUs -- profiles/r04_channels_last_cause.txt")
    ap.add_argument("--val-split", type=float, default=0.2, help="fraction of every iteration's rows held out for validation "
                    "(the reference's validation_split, SL/train.py:66); 0 = train on everything")
    ap.add_argument("--temp-moves", type=int, default=10)
    ap.add_argument("--arena-games", type=int, default=256)
    ap.add_argument("--arena-sims", type=int, default=64)
    ap.add_argument("--depth", type=int, default=3, help="search depth of the minimax opponent")
    ap.add_argument("--final-depths", default="1,3,5", help="minimax depths the final net is also played against")
    ap.add_argument("--opening-plies", type=int, default=4, help="random legal moves before the arena players take over")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--fp32-train", action="store_true", help="train without bf16 autocast (A/B of the loss curve)")
    ap.add_argument("--eager-train", action="store_true", help="launch every kernel of a training step by itself instead of replaying the captured HIP graph")
    ap.add_argument("--miopen-train", action="store_true", help="train the tower through stock autograd (MIOpen) instead of the HIP training kernels (csrc/bz_train.hip)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    torch.manual_seed(args.seed)
    gen = torch.Generator(device="cuda:0").manual_seed(args.seed)
    kernels = not (args.miopen_train or args.eager_train or args.fp32_train or args.channels == 256)
    module = PolicyValueNet(args.channels, args.blocks, 64, fused_tower=kernels)
    opt = make_optimizer(module, lr=args.lr) if args.eager_train else None
    graphed = None if args.eager_train else GraphedTrainStep(module, lr=args.lr, batch=args.batch, autocast=not args.fp32_train, lr_warmup_steps=args.lr_warmup)
    bmax = max(args.games, args.arena_games)
    dnet = DeviceNet.from_module(module.round_to_bf16_(), bmax)
    lines = []

    def emit(d):
        lines.append(d)
        print(json.dumps(d), flush=True)

    def arena(evaluator, net, depth=None):
        t0 = time.time()
        try:
            res = play_arena("reversi", args.arena_games, args.arena_sims, opponent_depth=depth or args.depth, evaluator=evaluator,
                             net=net, seed=args.seed, opening_plies=args.opening_plies)
        except RuntimeError:  # keep the weights that were in play for a post-mortem
            if args.out:
                np.save(args.out + ".failed_params.npy", module.flat_params())
            raise
        s = res.summary()
        s["score"] = round((s["wins"] + 0.5 * s["draws"]) / s["games"], 4)
        s["seconds"] = round(time.time() - t0, 1)
        return s

    emit({"what": "arena, MCTS with the uniform evaluator (no net)", "sims": args.arena_sims, "opponent_depth": args.depth,
          **arena("uniform", None)})
    emit({"what": "arena, untrained net", "iter": 0, **arena("net_bf16", dnet)})

    window, val_window = [], []
    for it in range(1, args.iters + 1):
        t0 = time.time()
        sp = PipelinedSelfPlay("reversi", args.games, args.sims, "net_bf16", dnet, pipelines=args.pipelines, temp_moves=args.temp_moves,
                               openings=1, seed=args.seed * 1000 + it, dirichlet_alpha=0.3, dirichlet_eps=0.25)
        plies = sp.run_iteration()
        ex = sp.device_examples()    # finished games' rows, packed on the device
        winners, _ = sp.winners()
        cnt = sp.counters()
        torch.cuda.synchronize()
        t_play = time.time() - t0
        t1 = time.time()
        # hold-out split of this iteration's rows BEFORE augmentation (a row's symmetric copies must not sit on both sides)
        tr_idx, va_idx = holdout_split(len(ex), args.val_split, gen, ex.own.device)
        aug = augment_examples(select_rows(ex, tr_idx), dedupe=True)
        del sp
        window = (window + [aug])[-args.window:]
        val_window = (val_window + [select_rows(ex, va_idx)])[-args.window:]
        data, val = concat_device_examples(window), concat_device_examples(val_window)
        val_before = validate(dnet, val)   # the net that just played, on rows it has not been trained on
        steps = max(1, int(args.epochs * len(data) / args.batch))
        losses = []
        for _ in range(steps):
            idx = torch.randint(0, len(data), (args.batch,), device=data.own.device, generator=gen)
            if graphed is not None:
                losses.append(graphed(data, idx))
            else:
                losses.append(torch.stack(train_step(module, opt, data, idx, autocast=not args.fp32_train)))
        losses = torch.stack(losses).cpu().numpy()  # one transfer per iteration, after the last step
        if graphed is not None:
            graphed.check()                         # an out-of-range row index in any step of the iteration raises here
        if not np.isfinite(losses).all():
            raise RuntimeError(f"training diverged in iteration {it}: first non-finite loss at step "
                               f"{int(np.argmax(~np.isfinite(losses).all(1)))} of {steps}")
        refresh_device_net(dnet, module)
        t_train = time.time() - t1
        val_after = validate(dnet, val)    # the refreshed engine net (bf16 MFMA forward) on the same held-out rows
        head, tail = np.mean(losses[: max(1, steps // 10)], axis=0), np.mean(losses[-max(1, steps // 10):], axis=0)
        emit({"what": "iteration", "iter": it, "games": args.games, "plies": plies, "examples": int(len(ex)),
              "augmented_rows": int(len(aug)), "train_rows": int(len(data)), "steps": steps,
              "loss_first_tenth": [round(float(x), 4) for x in head], "loss_last_tenth": [round(float(x), 4) for x in tail],
              "validation": {"rows": val_after["rows"], "split": args.val_split,
                             "before": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in val_before.items() if k != "rows"},
                             "after": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in val_after.items() if k != "rows"}},
              "self_play_x_wins": int((winners > 0).sum()), "self_play_o_wins": int((winners < 0).sum()),
              "self_play_s": round(t_play, 1), "games_per_s": round(args.games / t_play, 1), "train_s": round(t_train, 1),
              "mean_walk_nodes": round(cnt["n_path_nodes"] / max(1, cnt["n_sims"]), 2),
              "evaluations_shared": round(cnt["n_cache_hits"] / max(1, cnt["n_cache_hits"] + cnt["n_net_leaves"]), 3),
              "arena": arena("net_bf16", dnet)})
    for d in (int(x) for x in args.final_depths.split(",") if x):
        emit({"what": "final arena", "opponent_depth": d, "sims": args.arena_sims, "trained": arena("net_bf16", dnet, d),
              "uniform_evaluator": arena("uniform", None, d)})
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"args": vars(args), "lines": lines}, f, indent=1)


if __name__ == "__main__":
    main()
