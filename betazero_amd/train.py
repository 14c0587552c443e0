"""Closing the AlphaZero loop: one optimisation step on (s, pi, z)
with stock PyTorch-ROCm autograd.  Loop shape after src/tic_tac_toe/SL/train.py:85-136
(cross-entropy on the policy, Adam lr 1e-4), plus the value MSE the reference lacks."""
import numpy as np
import torch
import torch.nn.functional as F

from .net import bits_to_planes


def make_optimizer(module, lr=1e-4):
    return torch.optim.Adam(module.parameters(), lr=lr)  # train.py:87,192


def train_step(module, optimizer, ex, idx=None, device="cuda:0"):
    """one Adam step on the rows `idx` of Examples ex (Reversi 8x8 net); returns (loss, policy CE, value MSE)"""
    module.to(device).train()
    idx = np.arange(len(ex)) if idx is None else np.asarray(idx)
    x = bits_to_planes(ex.own[idx], ex.opp[idx]).to(device)
    pi = torch.as_tensor(ex.pi[idx]).to(device)
    z = torch.as_tensor(ex.z[idx].astype(np.float32)).to(device)
    logits, v = module(x)
    ce = -(pi * F.log_softmax(logits, dim=1)).sum(1).mean()
    mse = F.mse_loss(v, z)
    loss = ce + mse
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    optimizer.step()
    return float(loss.detach()), float(ce.detach()), float(mse.detach())


def refresh_device_net(device_net, module):
    """push the trained weights into the HIP engine's net (bf16-rounded copy for the MFMA path)"""
    import copy
    m = copy.deepcopy(module).cpu()
    m.round_to_bf16_()
    device_net.update(m.flat_params())
