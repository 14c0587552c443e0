"""Pool (s, pi, z) examples across the GPUs of a node: ONE all-gather at iteration end (RCCL over xGMI when the
process group backend is "nccl"; gloo on CPU for tests).  Games shard embarrassingly, so this is the only collective
on the path.

What travels is each rank's *packed example block* (include/bz_abi.h "Packed examples"): the rows of the FINISHED
games only -- the reference pools complete games, src/tic_tac_toe/SL/generate_training_games.py:30-36 -- compacted on
the device by bz_engine_pack_examples behind a 256-byte header that carries the row count, so neither a second
collective for metadata nor an all-gather-v is needed.  The block has a fixed capacity (the same on every rank) and
both ends of the collective can be allocated once, before any timed region (GatherBuffers): the exchange itself
allocates nothing.

The engines' raw fixed-capacity blocks (SelfPlayEngine.example_block(): every round, finished or not) can still be
gathered with all_gather_example_blocks(); the packed path replaced it on the data path in round 4."""
import torch
import torch.distributed as dist

from .engine import (PipelinedSelfPlay, alloc_packed_block, concat_device_examples, concat_examples, packed_layout,
                     unpack_example_block, unpack_example_block_device, unpack_packed_block, unpack_packed_block_device)


class GatherBuffers:
    """both ends of the one all-gather, allocated once.  `pack` = the packed block the pack kernels fill on the
    engines' GPU (cap_rows rows), `send` = what the collective sends -- the same tensor under nccl, a host copy of it
    under gloo (whose collectives run on host memory) -- and `out` = [world, bytes] where the collective runs."""

    def __init__(self, na, cap_rows, world, engine_device, collective_device=None):
        self.na, self.cap_rows, self.world = na, int(cap_rows), world
        self.nbytes = packed_layout(na, self.cap_rows)[1]
        cdev = torch.device(collective_device if collective_device is not None else engine_device)
        self.pack = alloc_packed_block(na, self.cap_rows, engine_device)
        self.send = self.pack if cdev.type == "cuda" else torch.empty(self.nbytes, dtype=torch.uint8, device="cpu")
        self.out = torch.empty((world, self.nbytes), dtype=torch.uint8, device=cdev)


def _as_source(source):
    if isinstance(source, PipelinedSelfPlay):
        return source
    return list(source) if isinstance(source, (list, tuple)) else [source]


def pack_for_gather(source, cap_rows=None, send=None):
    """this rank's finished games -> ONE packed block on the engines' device (kernels only; no host visit).
    source: a PipelinedSelfPlay, a SelfPlayEngine or a list of engines of one geometry.  cap_rows must be the same on
    every rank (default: every slot of every round of the local engines, which are built alike on every rank)."""
    src = _as_source(source)
    if isinstance(src, PipelinedSelfPlay):
        return src.pack_examples(send, cap_rows)
    cap = int(cap_rows or sum(e.rounds * e.B * e.t_max for e in src))
    if send is None:
        send = alloc_packed_block(src[0].na, cap, src[0].device)
    for i, e in enumerate(src):
        e.pack_examples(send, cap, append=i > 0)
    return send


def all_gather_packed(send, out=None, group=None):
    """exactly ONE collective: every rank's packed block -> [world, bytes] on every rank.  `out` preallocated (e.g.
    GatherBuffers.out) keeps allocation out of the exchange."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world, send.numel()), dtype=torch.uint8, device=send.device)
    assert out.shape == (world, send.numel()) and out.device == send.device and send.is_contiguous()
    dist.all_gather_into_tensor(out.view(-1), send, group=group)
    return out


def _collective_device(group=None):
    return "cuda" if dist.get_backend(group) == "nccl" else "cpu"


def gather_packed(source, group=None, cap_rows=None, buffers=None):
    """pack this rank's finished games (device kernels) and run the ONE all-gather; returns the gathered
    [world, bytes] tensor (buffers.out when `buffers` is given: nothing is allocated)"""
    pack = pack_for_gather(source, cap_rows if buffers is None else buffers.cap_rows, None if buffers is None else buffers.pack)
    if _collective_device(group) == "cpu":  # gloo: the collective runs on host memory
        if buffers is not None:
            buffers.send.copy_(pack)
            pack = buffers.send
        else:
            pack = pack.cpu()
    return all_gather_packed(pack, None if buffers is None else buffers.out, group)


def gather_examples(source, group=None, cap_rows=None, buffers=None):
    """All ranks get the pooled Examples (host) of every rank's finished games: pack (device kernels), ONE
    all-gather, then only the valid rows of each rank's block are copied out."""
    out = gather_packed(source, group, cap_rows, buffers)
    return concat_examples([unpack_packed_block(out[r]) for r in range(out.shape[0])])


def gather_examples_device(source, group=None, cap_rows=None, buffers=None):
    """gather_examples() without the host: the pooled rows as DeviceExamples on this rank's GPU (backend "nccl" =
    RCCL over xGMI).  Per rank one 256-byte header read-back (the row count); no example data leaves the GPU."""
    out = gather_packed(source, group, cap_rows, buffers)
    if not out.is_cuda:
        dev = _as_source(source)
        dev = dev.device if isinstance(dev, PipelinedSelfPlay) else dev[0].device
        out = out.to(dev)
    return concat_device_examples([unpack_packed_block_device(out[r]) for r in range(out.shape[0])])


# ---- the engines' raw example blocks (every round, finished or not): kept for callers that want the fixed layout
def all_gather_example_blocks(blocks, group=None, send=None, out=None):
    """blocks: the raw example blocks (1-D uint8 tensors) of this rank's engines.  Issues exactly ONE collective and
    returns (gathered [world, bytes] uint8 tensor, per-engine block sizes).  A single block is sent in place; several
    are first laid side by side in `send` (allocated here unless given)."""
    sizes = [int(b.numel()) for b in blocks]
    if len(blocks) == 1:
        send = blocks[0]
    else:
        if send is None:
            send = torch.empty(sum(sizes), dtype=torch.uint8, device=blocks[0].device)
        torch.cat([b.reshape(-1) for b in blocks], out=send)
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world, send.numel()), dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(out.view(-1), send.contiguous(), group=group)
    return out, sizes


def split_gathered(gathered, sizes):
    """[world, bytes] -> list over ranks of lists over engines of raw block views"""
    res = []
    for r in range(gathered.shape[0]):
        off, row = 0, []
        for n in sizes:
            row.append(gathered[r, off:off + n])
            off += n
        res.append(row)
    return res


def gather_raw_examples(engines, group=None):
    """pooled Examples through the raw blocks (one collective)"""
    if not isinstance(engines, (list, tuple)):
        engines = [engines]
    gathered, sizes = all_gather_example_blocks([e.example_block() for e in engines], group)
    return concat_examples([unpack_example_block(b) for row in split_gathered(gathered, sizes) for b in row])


def gather_raw_examples_device(engines, group=None):
    if not isinstance(engines, (list, tuple)):
        engines = [engines]
    gathered, sizes = all_gather_example_blocks([e.example_block() for e in engines], group)
    geoms = [e.block_geometry() for e in engines]
    return concat_device_examples([unpack_example_block_device(b, g) for row in split_gathered(gathered, sizes)
                                   for b, g in zip(row, geoms)])
