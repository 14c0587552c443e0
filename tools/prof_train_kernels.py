#!/usr/bin/env python3
"""Only the three tower training kernels (no torch step around them), a few launches each: the target of
`rocprofv3 --kernel-trace --pmc ...` passes (a whole training step under --pmc serialises hundreds of torch kernels).
python3 tools/prof_train_kernels.py [C] [blocks] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd import _lib  # noqa: E402
from betazero_amd.train_kernels import TowerPlan  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
L, Ly = _lib.lib(), 2 * NB
p = TowerPlan(C, Ly, B)
g = torch.Generator(device="cuda:0").manual_seed(0)
W = torch.randn((Ly, C, C, 3, 3), device="cuda:0", generator=g) * (1.5 / (9 * C)) ** 0.5
bias = torch.zeros((Ly, C), device="cuda:0")
p.acts[0].copy_(torch.relu(torch.randn((B, 64, C), device="cuda:0", generator=g)))
p.gs[Ly].copy_(torch.randn((B, 64, C), device="cuda:0", generator=g))
st = torch.cuda.current_stream().cuda_stream
for _ in range(12):
    _lib.check(L.bz_train_pack_weights(W.data_ptr(), C, Ly, p.wf_fwd.data_ptr(), p.wf_bwd.data_ptr(), st))
    _lib.check(L.bz_train_tower_fwd(p.acts[0].data_ptr(), p.wf_fwd.data_ptr(), bias.data_ptr(), C, Ly, B, p.acts[1].data_ptr(), p.masks.data_ptr(), st))
    _lib.check(L.bz_train_tower_bwd(p.gs[Ly].data_ptr(), p.wf_bwd.data_ptr(), p.zeros_c.data_ptr(), p.masks.data_ptr(), C, Ly, B, p.gs[0].data_ptr(), st))
    _lib.check(L.bz_train_wgrad(p.acts[0].data_ptr(), p.gs[1].data_ptr(), C, Ly, B, p.splits, p.partial.data_ptr(), p.db_partial.data_ptr(), st))
torch.cuda.synchronize()
print("done", C, NB, B, flush=True)
