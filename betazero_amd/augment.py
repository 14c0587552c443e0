"""8-fold symmetry augmentation + dedupe of (s, pi, z) examples on the GPU.
Reference: TicTacToeDataset.expand_with_transforms, src/tic_tac_toe/SL/train.py:24-52
(the 8 transforms in its order, then keep the first occurrence of every distinct
(state, action) pair in insertion order).

Device in, device out: DeviceExamples (what gather_examples_device / SelfPlayEngine.device_examples hand over)
stay on the GPU through the transforms, the dedupe and into train_step; host Examples are accepted too and come
back as host Examples."""
import torch

from . import _lib
from .engine import DeviceExamples, Examples


def _first_occurrences(key8, own8, opp8, pi8):
    """indices (ascending = insertion order) of the rows to keep: the first of every set of EXACTLY equal
    (own, opp, pi) rows -- SL/train.py:45-50 keeps the first of every exactly-equal pair.  Rows are grouped by the
    kernel's 64-bit content key (stable sort: insertion order inside a group) and every row is compared with the HEAD
    of its group on the full content: the head is kept, rows equal to it are duplicates and dropped, rows that differ
    (another content sharing the key) go into the next pass among themselves -- so the result is exact whatever the
    keys do: a collision can neither drop a distinct row nor keep a duplicate.  With honest keys that is one pass plus
    one emptiness check.  pi is compared by BIT PATTERN (a row holding a NaN equals itself -- with float comparison a
    group head containing a NaN differed from itself and was handed to the next pass for ever), and a pass never hands
    its own heads on, so every pass removes at least one row: the loop ends."""
    kept = []
    pi8 = pi8.view(torch.int32)
    cur = torch.arange(key8.numel(), device=key8.device)
    while cur.numel():
        sk, o = torch.sort(key8[cur], stable=True)
        order = cur[o]
        newrun = torch.ones_like(sk, dtype=torch.bool)
        newrun[1:] = sk[1:] != sk[:-1]
        heads = newrun.nonzero(as_tuple=True)[0]                 # sorted positions of the group heads
        head_row = order[heads[torch.cumsum(newrun, 0) - 1]]     # for every sorted position: the row id of its group's head
        same = (own8[order] == own8[head_row]) & (opp8[order] == opp8[head_row]) & (pi8[order] == pi8[head_row]).all(1)
        kept.append(order[newrun])
        cur = torch.sort(order[~same & ~newrun]).values          # ascending again: the stable sort keeps insertion order
    return torch.sort(torch.cat(kept)).values if kept else cur


def augment_examples(ex, dedupe=True, device="cuda:0"):
    """Examples / DeviceExamples (n rows) -> the same type (<= 8n rows, row 8*i+t = transform t of row i before dedupe).
    z, mover and game carry over; `act` is permuted with the board; `ply` is kept."""
    _lib.require_gpu()
    host_in = isinstance(ex, Examples)
    if host_in:
        ex = DeviceExamples.from_host(ex, device)
    dev = ex.own.device
    n, na, size = len(ex), ex.pi.shape[1], ex.size
    own, opp, pi = ex.own.contiguous(), ex.opp.contiguous(), ex.pi.contiguous()
    own8 = torch.empty(8 * n, dtype=torch.int64, device=dev)
    opp8 = torch.empty(8 * n, dtype=torch.int64, device=dev)
    pi8 = torch.empty((8 * n, na), dtype=torch.float32, device=dev)
    key8 = torch.empty(8 * n, dtype=torch.int64, device=dev)
    # the move played, as a one-hot "policy", goes through the same kernel to permute `act`
    act1 = torch.zeros((n, na), dtype=torch.float32, device=dev)
    act1[torch.arange(n, device=dev), ex.act.to(torch.int64)] = 1.0
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
    keep = _first_occurrences(key8, own8, opp8, pi8) if dedupe else torch.arange(8 * n, device=dev)
    src = keep // 8
    out = DeviceExamples(own=own8[keep], opp=opp8[keep], pi=pi8[keep], z=ex.z[src], mover=ex.mover[src],
                         act=act8[keep].argmax(1).to(torch.uint8), game=ex.game[src], ply=ex.ply[src], size=size)
    return out.cpu() if host_in else out
