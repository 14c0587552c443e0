"""Pool (s, pi, z) examples across the GPUs of a node: ONE all-gather of the fixed-capacity
example blocks at iteration end (RCCL over xGMI when the process group backend is "nccl"; gloo on
CPU for tests).  Games shard embarrassingly, so this is the only collective on the path
(SURVEY.md 8(e)).

What travels is each engine's example block: the contiguous byte range [ex_own .. ex_winner |
256-byte header] of its workspace (SelfPlayEngine.example_block(), include/bz_abi.h).  The header
carries the rank's game-id base / stride and the array geometry, and `ex_len` carries the valid
row counts, so neither a second collective for metadata nor an all-gather-v is needed."""
import torch
import torch.distributed as dist

from .engine import concat_device_examples, concat_examples, unpack_example_block, unpack_example_block_device


def all_gather_example_blocks(blocks, group=None):
    """blocks: the example blocks (1-D uint8 tensors) of this rank's engines -- one, or one per
    pipeline.  Issues exactly ONE collective and returns (gathered [world, bytes] uint8 tensor,
    per-engine block sizes).  A single block is sent in place (no copy); several are first laid
    side by side in one send buffer (a device-to-device copy)."""
    sizes = [int(b.numel()) for b in blocks]
    send = blocks[0] if len(blocks) == 1 else torch.cat([b.reshape(-1) for b in blocks])
    world = dist.get_world_size(group)
    out = torch.empty((world, send.numel()), dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(out.view(-1), send.contiguous(), group=group)
    return out, sizes


def split_gathered(gathered, sizes):
    """[world, bytes] -> list over ranks of lists over engines of block views"""
    res = []
    for r in range(gathered.shape[0]):
        off, row = 0, []
        for n in sizes:
            row.append(gathered[r, off:off + n])
            off += n
        res.append(row)
    return res


def gather_examples(engines, group=None):
    """All ranks get the pooled Examples of every rank's finished games.  `engines`: one
    SelfPlayEngine or a list of them (the pipelines of this rank).  One collective; the finished
    rows are selected where the gathered buffer lives, so only they are copied to the host."""
    if not isinstance(engines, (list, tuple)):
        engines = [engines]
    gathered, sizes = all_gather_example_blocks([e.example_block() for e in engines], group)
    return concat_examples([unpack_example_block(b) for row in split_gathered(gathered, sizes) for b in row])


def gather_examples_device(engines, group=None):
    """gather_examples() without the host: the pooled rows of every rank's finished games as DeviceExamples on this
    rank's GPU (one collective; backend "nccl" = RCCL over xGMI).  Every rank's engines are built alike (same game,
    slot count, rounds), so a peer's block has the geometry of the local engine at the same position."""
    if not isinstance(engines, (list, tuple)):
        engines = [engines]
    gathered, sizes = all_gather_example_blocks([e.example_block() for e in engines], group)
    geoms = [e.block_geometry() for e in engines]
    return concat_device_examples([unpack_example_block_device(b, g) for row in split_gathered(gathered, sizes)
                                   for b, g in zip(row, geoms)])
