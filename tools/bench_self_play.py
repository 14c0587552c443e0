#!/usr/bin/env python3
"""The public entry point at the headline's size: betazero_amd.engine.self_play("reversi", 4096, 800, net=...) -- the
batched collect_game_data (src/tic_tac_toe/SL/generate_training_games.py:25-38) -- timed from the call to the returned
(s, pi, z), examples on the host included, next to bench.py's own whole-iteration figure (`--mode iteration`, the same
PipelinedSelfPlay object underneath).  VERDICT r3 item 1: the two must agree within 2 %.
python tools/bench_self_play.py [games] [sims]"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from betazero_amd.engine import self_play  # noqa: E402
from betazero_amd.net import DeviceNet, PolicyValueNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 800
torch.manual_seed(0)
net = DeviceNet.from_module(PolicyValueNet(128, 6, 64).round_to_bf16_(), B)
self_play("reversi", 64, 16, net=net, temp_moves=8, openings=1)  # library, streams, probe: not part of the measurement
torch.cuda.synchronize()
t0 = time.perf_counter()
s, pi, z, ex = self_play("reversi", B, sims, net=net, seed=0, temp_moves=8, openings=1)
dt = time.perf_counter() - t0
print(json.dumps({"what": "self_play() wall time, call to returned (s, pi, z) on the host", "games": B, "sims": sims, "seconds": dt,
                  "games_per_s": B / dt, "rows": int(len(ex)), "s_shape": list(s.shape), "pi_shape": list(pi.shape)}), flush=True)
del net
r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--mode", "iteration",
                    "--steps", "1", "--warmup", "0", "--games", str(B), "--sims", str(sims), "--no-cpu-baseline", "--no-secondary"],
                   capture_output=True, text=True)
d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
print(json.dumps({"what": "bench.py --mode iteration (same device, next process)", "games_per_s": d["value"], "seconds": d["ms_per_step"] / 1e3,
                  "self_play_over_bench": (B / dt) / d["value"]}))
