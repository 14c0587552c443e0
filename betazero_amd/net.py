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


class PolicyValueNet(nn.Module):
    def __init__(self, channels=128, blocks=6, value_hidden=64):
        super().__init__()
        self.C, self.NB, self.VH = channels, blocks, value_hidden
        self.stem = nn.Conv2d(2, channels, 3, padding=1)
        self.c1 = nn.ModuleList(nn.Conv2d(channels, channels, 3, padding=1) for _ in range(blocks))
        self.c2 = nn.ModuleList(nn.Conv2d(channels, channels, 3, padding=1) for _ in range(blocks))
        self.pol = nn.Conv2d(channels, 2, 1)
        self.polfc = nn.Linear(128, 65)
        self.val = nn.Conv2d(channels, 1, 1)
        self.v1 = nn.Linear(64, value_hidden)
        self.v2 = nn.Linear(value_hidden, 1)

    def forward(self, planes):  # [B,2,8,8] float (own, opp) -> logits [B,65], value [B]
        x = F.relu(self.stem(planes))
        for a, b in zip(self.c1, self.c2):
            x = F.relu(b(F.relu(a(x))) + x)
        p = self.polfc(F.relu(self.pol(x)).flatten(1))
        v = torch.tanh(self.v2(F.relu(self.v1(F.relu(self.val(x)).flatten(1))))).squeeze(-1)
        return p, v

    def flat_params(self):
        """fp32 vector in the order bz_net_create / the oracle expect (torch layouts)."""
        mods = [self.stem]
        for a, b in zip(self.c1, self.c2):
            mods += [a, b]
        mods += [self.pol, self.polfc, self.val, self.v1, self.v2]
        parts = []
        for m in mods:
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
