"""8-fold symmetry augmentation + dedupe of (s, pi, z) examples on the GPU.
Reference: TicTacToeDataset.expand_with_transforms, src/tic_tac_toe/SL/train.py:24-52
(the 8 transforms in its order, then keep the first occurrence of every distinct
(state, action) pair in insertion order)."""
import numpy as np
import torch

from . import _lib
from .engine import Examples


def augment_examples(ex, dedupe=True, device="cuda:0"):
    """Examples (n rows) -> Examples (<= 8n rows, row 8*i+t = transform t of row i before dedupe).
    z, mover and game carry over; `act` is permuted with the board; `ply` is kept."""
    _lib.require_gpu()
    n, na, size = len(ex), ex.pi.shape[1], ex.size
    dev = torch.device(device)
    own = torch.as_tensor(ex.own.view(np.int64)).to(dev)
    opp = torch.as_tensor(ex.opp.view(np.int64)).to(dev)
    onehot_cols = size * size
    pi = torch.as_tensor(np.ascontiguousarray(ex.pi, dtype=np.float32)).to(dev)
    own8 = torch.empty(8 * n, dtype=torch.int64, device=dev)
    opp8 = torch.empty(8 * n, dtype=torch.int64, device=dev)
    pi8 = torch.empty((8 * n, na), dtype=torch.float32, device=dev)
    key8 = torch.empty(8 * n, dtype=torch.int64, device=dev)
    # the move played, as a one-hot "policy", goes through the same kernel to permute `act`
    act1 = torch.zeros((n, na), dtype=torch.float32, device=dev)
    act1[torch.arange(n, device=dev), torch.as_tensor(ex.act.astype(np.int64)).to(dev)] = 1.0
    act8 = torch.empty((8 * n, na), dtype=torch.float32, device=dev)
    scr_a = torch.empty(8 * n, dtype=torch.int64, device=dev)  # two distinct buffers: the kernel's outputs
    scr_b = torch.empty(8 * n, dtype=torch.int64, device=dev)  # are __restrict__
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream().cuda_stream
        L = _lib.lib()
        _lib.check(L.bz_augment_d4_batch(own.data_ptr(), opp.data_ptr(), pi.data_ptr(), n, size, na, own8.data_ptr(),
                                         opp8.data_ptr(), pi8.data_ptr(), key8.data_ptr(), st))
        _lib.check(L.bz_augment_d4_batch(own.data_ptr(), opp.data_ptr(), act1.data_ptr(), n, size, na,
                                         scr_a.data_ptr(), scr_b.data_ptr(), act8.data_ptr(), None, st))
    keep = torch.arange(8 * n, device=dev)
    if dedupe:  # first occurrence of every distinct row, insertion order preserved
        sk, order = torch.sort(key8, stable=True)
        first = torch.ones_like(sk, dtype=torch.bool)
        first[1:] = sk[1:] != sk[:-1]
        keep = torch.sort(order[first]).values
    src = (keep // 8).cpu().numpy()
    k = keep
    return Examples(own=own8[k].cpu().numpy().view(np.uint64), opp=opp8[k].cpu().numpy().view(np.uint64),
                    pi=pi8[k].cpu().numpy(), z=ex.z[src], mover=ex.mover[src],
                    act=act8[k].argmax(1).cpu().numpy().astype(np.uint8), game=ex.game[src], ply=ex.ply[src], size=size)
