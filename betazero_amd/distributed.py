"""Pool (s, pi, z) examples across the GPUs of a node: one all-gather of the
fixed-capacity example buffers at iteration end (RCCL over xGMI when the
process group backend is "nccl"; gloo on CPU for tests).  Games shard
embarrassingly, so this is the only collective on the path."""
import numpy as np
import torch
import torch.distributed as dist

from .engine import pack_examples

_KEYS = ("own", "opp", "pi", "z", "mover", "act", "len", "winner")


def all_gather_example_tensors(tensors, group=None):
    """tensors: dict of [R,B,T,...] tensors (SelfPlayEngine.example_tensors()).
    Returns dict of [world, R, B, T, ...] tensors on the same device.  Every rank
    contributes the same fixed capacity, so no all-gather-v is needed; the `len`
    array carries the valid row counts."""
    world = dist.get_world_size(group)
    out = {}
    for k in _KEYS:
        t = tensors[k].contiguous()
        flat = t.view(torch.uint8).reshape(-1) if t.dtype != torch.uint8 else t.reshape(-1)
        buf = torch.empty((world,) + tuple(flat.shape), dtype=torch.uint8, device=flat.device)
        dist.all_gather_into_tensor(buf.view(-1), flat, group=group) if flat.is_cuda else \
            dist.all_gather(list(buf.unbind(0)), flat, group=group)
        out[k] = buf.view(world, -1).view(t.dtype).view((world,) + tuple(t.shape))
    return out


def gather_examples(engine, group=None):
    """All ranks get the pooled Examples of every rank's finished games (global game ids come from each
    rank's own game_id_base / game_id_stride, gathered alongside the buffers)."""
    g = all_gather_example_tensors(engine.example_tensors(), group)
    world = dist.get_world_size(group)
    dev = next(iter(g.values())).device
    meta = torch.tensor([int(engine.cfg.game_id_base), int(engine.cfg.game_id_stride)], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    parts = []
    size = getattr(engine, "size", 3 if engine.t_max == 9 else 8)
    for r in range(world):
        t = {k: g[k][r].cpu().numpy() for k in _KEYS}
        base, stride = (int(v) for v in metas[r].cpu())
        parts.append(pack_examples(t, base, stride, size))
    from .engine import Examples
    cat = lambda f: np.concatenate([getattr(p, f) for p in parts])  # noqa: E731
    return Examples(cat("own"), cat("opp"), cat("pi"), cat("z"), cat("mover"), cat("act"), cat("game"), cat("ply"), size)
