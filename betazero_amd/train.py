"""Closing the AlphaZero loop: one optimisation step on (s, pi, z)
with stock PyTorch-ROCm autograd.  Loop shape after src/tic_tac_toe/SL/train.py:85-136
(cross-entropy on the policy, Adam lr 1e-4), plus the value MSE the reference lacks.

The data path is device-resident: DeviceExamples in (from gather_examples_device / augment_examples), batches are
index-selected on the GPU, the two input planes are expanded from the bitboards by torch ops on the GPU, forward and
backward run under bf16 autocast (fp32 master weights, fp32 losses).  Nothing is copied to the host; the three loss
values come back as device scalars (call .item() when you want to look at them, not every step)."""
import torch
import torch.nn.functional as F

from .engine import DeviceExamples, Examples

_SHIFTS = {}


def planes_from_bits(own, opp):
    """int64 tensors holding uint64 bitboards [n] -> float planes [n, 2, 8, 8] (bit 8*r+c), on their device.
    (>> on int64 is arithmetic, which still leaves bit s of the pattern at bit 0 for every s <= 63.)"""
    dev = own.device
    sh = _SHIFTS.get(dev)
    if sh is None:
        sh = _SHIFTS[dev] = torch.arange(64, dtype=torch.int64, device=dev)
    a = ((own[:, None] >> sh) & 1).to(torch.float32).view(-1, 8, 8)
    b = ((opp[:, None] >> sh) & 1).to(torch.float32).view(-1, 8, 8)
    return torch.stack([a, b], dim=1)


def make_optimizer(module, lr=1e-4):
    return torch.optim.Adam(module.parameters(), lr=lr)  # train.py:87,192


def train_step(module, optimizer, ex, idx=None, device="cuda:0", autocast=True):
    """one Adam step on the rows `idx` of ex (Reversi 8x8 net).  ex: DeviceExamples (stays on the GPU; idx a device
    index tensor or None = all rows) or host Examples (uploaded first).  Returns (loss, policy CE, value MSE) as
    detached device scalars."""
    if isinstance(ex, Examples):
        ex = DeviceExamples.from_host(ex, device)
    dev = ex.own.device
    if next(module.parameters()).device != dev:
        module.to(dev)
    module.train()
    if idx is None:
        own, opp, pi, z = ex.own, ex.opp, ex.pi, ex.z
    else:
        idx = torch.as_tensor(idx, device=dev)
        own, opp, pi, z = ex.own[idx], ex.opp[idx], ex.pi[idx], ex.z[idx]
    x = planes_from_bits(own, opp)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        logits, v = module(x)
    logits, v = logits.float(), v.float()
    ce = -(pi * F.log_softmax(logits, dim=1)).sum(1).mean()
    mse = F.mse_loss(v, z.to(torch.float32))
    loss = ce + mse
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    optimizer.step()
    return loss.detach(), ce.detach(), mse.detach()


def refresh_device_net(device_net, module):
    """push the trained weights into the HIP engine's net (bf16-rounded copy for the MFMA path)"""
    import copy
    m = copy.deepcopy(module).cpu()
    m.round_to_bf16_()
    device_net.update(m.flat_params())
