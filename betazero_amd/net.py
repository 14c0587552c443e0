"""Policy/value net: the torch module (fp32 definition + default init) and the
device-side handle around bz_net (weights repacked for the HIP kernels).

Architecture: SURVEY.md 8(d) "net" (build-authored; the reference only has a
policy MLP, SL/neural_networks.py:4-30).  The calling convention follows
AIPlayer.get_move (players.py:84-98): canonical side-to-move input, logits out."""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


_NBHD = {}  # device -> gather table of the stem's 3x3 neighbourhoods (PolicyValueNet._forward_kernels)


class PolicyValueNet(nn.Module):
    """fused_tower=True keeps the 2 x blocks conv3x3 layers of the residual tower as TWO stacked parameters (tower_w
    [L, C, C, 3, 3], tower_b [L, C]; same initial values, same flat_params() order as the per-layer modules) -- the form
    the hand-written training kernels take (train_kernels.tower_apply; pass forward() a TowerPlan to use them)."""

    def __init__(self, channels=128, blocks=6, value_hidden=64, fused_tower=False):
        super().__init__()
        self.C, self.NB, self.VH, self.fused_tower = channels, blocks, value_hidden, fused_tower
        self.stem = nn.Conv2d(2, channels, 3, padding=1)
        c1 = nn.ModuleList(nn.Conv2d(channels, channels, 3, padding=1) for _ in range(blocks))
        c2 = nn.ModuleList(nn.Conv2d(channels, channels, 3, padding=1) for _ in range(blocks))
        if fused_tower:  # (the per-layer modules were created first so that both forms draw the same initial weights)
            order = [m for a, b in zip(c1, c2) for m in (a, b)]
            self.tower_w = nn.Parameter(torch.stack([m.weight.detach() for m in order]))
            self.tower_b = nn.Parameter(torch.stack([m.bias.detach() for m in order]))
        else:
            self.c1, self.c2 = c1, c2
        self.pol = nn.Conv2d(channels, 2, 1)
        self.polfc = nn.Linear(128, 65)
        self.val = nn.Conv2d(channels, 1, 1)
        self.v1 = nn.Linear(64, value_hidden)
        self.v2 = nn.Linear(value_hidden, 1)

    def forward(self, planes, plan=None):  # [B,2,8,8] float (own, opp) -> logits [B,65], value [B]
        if self.fused_tower and plan is not None:
            return self._forward_kernels(planes, plan)
        x = F.relu(self.stem(planes))
        if self.fused_tower:
            for blk in range(self.NB):
                h = F.relu(F.conv2d(x, self.tower_w[2 * blk], self.tower_b[2 * blk], padding=1))
                x = F.relu(F.conv2d(h, self.tower_w[2 * blk + 1], self.tower_b[2 * blk + 1], padding=1) + x)
        else:
            assert plan is None, "a TowerPlan needs PolicyValueNet(..., fused_tower=True)"
            for a, b in zip(self.c1, self.c2):
                x = F.relu(b(F.relu(a(x))) + x)
        p = self.polfc(F.relu(self.pol(x)).flatten(1))
        v = torch.tanh(self.v2(F.relu(self.v1(F.relu(self.val(x)).flatten(1))))).squeeze(-1)
        return p, v

    def _forward_kernels(self, planes, plan):
        """the training forward around the HIP tower kernels (train_kernels.tower_apply_nhwc), everything in the
        kernels' [n, 64 cells, C] layout so that no tensor is permuted on the way in or out, and without a convolution
        solver anywhere: the stem is a K = 18 (padded to 32) matmul over the 3x3 neighbourhoods of the two planes, the
        two 1x1 head convolutions are ONE C -> 3 (padded to 32) matmul; the FCs as they are.  Same function as forward()."""
        from .train_kernels import rows_linear, tower_apply_nhwc
        n, C = planes.shape[0], self.C
        dev = planes.device
        tab = _NBHD.get(dev)
        if tab is None:  # cell, k = 9 plane + tap -> index into [plane 0 | plane 1 | one zero]; k >= 18 and off-board taps -> the zero
            idx = torch.full((64, 32), 128, dtype=torch.int64)
            for cell in range(64):
                for pl in range(2):
                    for t in range(9):
                        y, x = cell // 8 + t // 3 - 1, cell % 8 + t % 3 - 1
                        if 0 <= y < 8 and 0 <= x < 8:
                            idx[cell, 9 * pl + t] = 64 * pl + 8 * y + x
            tab = _NBHD[dev] = idx.reshape(-1).to(dev)
        flat = torch.cat([planes.reshape(n, 128), planes.new_zeros((n, 1))], dim=1)
        cols = flat.index_select(1, tab).view(n * 64, 32)                                        # [n x 64, 32]
        ws = F.pad(self.stem.weight.view(C, 18), (0, 14))                                        # [C, 32]
        x = F.relu(rows_linear(cols, ws, self.stem.bias)).view(n, 64, C)
        x = tower_apply_nhwc(x, self.tower_w, self.tower_b, plan)
        hw = F.pad(torch.cat([self.pol.weight.view(2, C), self.val.weight.view(1, C)]), (0, 0, 0, 29))   # [32, C]
        hb = F.pad(torch.cat([self.pol.bias, self.val.bias]), (0, 29))
        hv = F.relu(rows_linear(x.view(n * 64, C), hw, hb)).view(n, 64, 32)
        p = self.polfc(hv[:, :, :2].transpose(1, 2).reshape(n, 128))                             # channel-major, as .flatten(1) of NCHW
        v = torch.tanh(self.v2(F.relu(self.v1(hv[:, :, 2])))).squeeze(-1)
        return p, v

    def flat_params(self):
        """fp32 vector in the order bz_net_create / the oracle expect (torch layouts)."""
        parts = [self.stem.weight.detach().reshape(-1), self.stem.bias.detach().reshape(-1)]
        if self.fused_tower:
            for l in range(2 * self.NB):
                parts += [self.tower_w[l].detach().reshape(-1), self.tower_b[l].detach().reshape(-1)]
        else:
            for a, b in zip(self.c1, self.c2):
                for m in (a, b):
                    parts += [m.weight.detach().reshape(-1), m.bias.detach().reshape(-1)]
        for m in (self.pol, self.polfc, self.val, self.v1, self.v2):
            parts += [m.weight.detach().reshape(-1), m.bias.detach().reshape(-1)]
        return torch.cat(parts).to(torch.float32).cpu().numpy().copy()

    @torch.no_grad()
    def round_to_bf16_(self):
        """Make every parameter bf16-representable (the bf16 MFMA path then holds
        exactly these weights; oracle and GPU see identical values)."""
        for p in self.parameters():
            p.copy_(p.to(torch.bfloat16).to(torch.float32))
        return self


def bits_to_planes(own, opp):
    """uint64 bitboards -> float planes [N,2,8,8] (bit 8*r+c)."""
    own = np.asarray(own, dtype=np.uint64).reshape(-1, 1)
    opp = np.asarray(opp, dtype=np.uint64).reshape(-1, 1)
    sh = np.arange(64, dtype=np.uint64).reshape(1, 64)
    a = ((own >> sh) & np.uint64(1)).astype(np.float32).reshape(-1, 8, 8)
    b = ((opp >> sh) & np.uint64(1)).astype(np.float32).reshape(-1, 8, 8)
    return torch.from_numpy(np.stack([a, b], axis=1))


class DeviceNet:
    """bz_net handle; owns the torch workspace tensor that holds weights + activations."""

    def __init__(self, channels, blocks, value_hidden, params, max_batch, device="cuda:0"):
        _lib.require_gpu()
        L = _lib.lib()
        params = np.ascontiguousarray(params, dtype=np.float32)
        assert params.size == L.bz_net_param_count(channels, blocks, value_hidden)
        nbytes = L.bz_net_workspace_bytes(channels, blocks, value_hidden, max_batch)
        if nbytes < 0:
            raise RuntimeError(_lib.last_error())
        self.device = torch.device(device)
        self.ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        self._base = (self.ws.data_ptr() + 255) & ~255
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.bz_net_create(channels, blocks, value_hidden, max_batch, params.ctypes.data, self._base,
                                       nbytes, torch.cuda.current_stream().cuda_stream, C.byref(h)))
        self.h = h
        self.C, self.NB, self.VH, self.max_batch = channels, blocks, value_hidden, max_batch

    @classmethod
    def from_module(cls, module, max_batch, device="cuda:0"):
        return cls(module.C, module.NB, module.VH, module.flat_params(), max_batch, device)

    def forward(self, own, opp, bf16=True, fp8=False):
        """own/opp: uint64-as-int64 CUDA tensors [n] -> (logits [n,65] f32, value [n] f32)"""
        n = own.numel()
        logits = torch.empty((n, 65), dtype=torch.float32, device=self.device)
        value = torch.empty((n,), dtype=torch.float32, device=self.device)
        fn = _lib.lib().bz_net_forward_fp8 if fp8 else (_lib.lib().bz_net_forward_bf16 if bf16 else _lib.lib().bz_net_forward_f32)
        with torch.cuda.device(self.device):
            _lib.check(fn(self.h, own.data_ptr(), opp.data_ptr(), n, logits.data_ptr(), value.data_ptr(),
                          torch.cuda.current_stream().cuda_stream))
        return logits, value

    def update(self, params):
        """replace the weights in place (same architecture), e.g. after a training step"""
        params = np.ascontiguousarray(params, dtype=np.float32)
        assert params.size == _lib.lib().bz_net_param_count(self.C, self.NB, self.VH)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().bz_net_update(self.h, params.ctypes.data, torch.cuda.current_stream().cuda_stream))

    def __del__(self):
        try:
            _lib.lib().bz_net_destroy(self.h)
        except Exception:
            pass
