"""SelfPlayEngine: torch-owned workspace + the bz_engine C ABI (batched MCTS
self-play on one GPU), and self_play(), the batched counterpart of the
reference's data-generation loop (SL/generate_training_games.py:25-38)."""
import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import (EVAL_EXTERNAL, EVAL_HASH, EVAL_NET_BF16, EVAL_NET_F32, EVAL_NET_FP8, EVAL_UNIFORM, GAME_REVERSI,
                   GAME_REVERSI4, GAME_REVERSI6, GAME_TTT, EngineCfg, EngineLayout)

_GAMES = {"ttt": GAME_TTT, "tic_tac_toe": GAME_TTT, "reversi": GAME_REVERSI, "reversi8": GAME_REVERSI,
          "reversi6": GAME_REVERSI6, "reversi4": GAME_REVERSI4, GAME_TTT: GAME_TTT, GAME_REVERSI: GAME_REVERSI,
          GAME_REVERSI6: GAME_REVERSI6, GAME_REVERSI4: GAME_REVERSI4}
_SIZES = {GAME_TTT: 3, GAME_REVERSI: 8, GAME_REVERSI6: 6, GAME_REVERSI4: 4}
_EVALS = {"uniform": EVAL_UNIFORM, "hash": EVAL_HASH, "net_f32": EVAL_NET_F32, "net_bf16": EVAL_NET_BF16,
          "external": EVAL_EXTERNAL, "net_fp8": EVAL_NET_FP8}


def _u64(t):
    """int64 tensor holding uint64 bit patterns -> numpy uint64"""
    return t.cpu().numpy().view(np.uint64)


@dataclass
class Examples:
    """(s, pi, z) rows.  own/opp: side-to-move canonical bitboards (the bitboard
    form of generate_training_games.py:17-18); pi [n, NA]; z in {-1,0,+1} for the
    mover; game = global game id; ply = row index inside its game."""
    own: np.ndarray
    opp: np.ndarray
    pi: np.ndarray
    z: np.ndarray
    mover: np.ndarray
    act: np.ndarray
    game: np.ndarray
    ply: np.ndarray
    size: int

    def __len__(self):
        return self.own.shape[0]

    def states(self):
        """canonical boards [n, size, size] int8: +1 = side to move, -1 = opponent"""
        s = self.size
        stride = 3 if s == 3 else 8
        sh = np.array([[stride * r + c for c in range(s)] for r in range(s)], dtype=np.uint64)
        a = ((self.own[:, None, None] >> sh) & np.uint64(1)).astype(np.int8)
        b = ((self.opp[:, None, None] >> sh) & np.uint64(1)).astype(np.int8)
        return a - b


@dataclass
class DeviceExamples:
    """The same rows as Examples, as torch tensors living on the GPU: what the device-resident pipeline
    gather -> augment -> train passes along (SL/train.py:24-52 then :85-136 in the reference's pipeline).
    own/opp: int64 tensors holding the uint64 bit patterns."""
    own: torch.Tensor
    opp: torch.Tensor
    pi: torch.Tensor
    z: torch.Tensor
    mover: torch.Tensor
    act: torch.Tensor
    game: torch.Tensor
    ply: torch.Tensor
    size: int

    def __len__(self):
        return int(self.own.shape[0])

    def cpu(self):
        """-> Examples (numpy, host)"""
        n = lambda t: t.cpu().numpy()  # noqa: E731
        return Examples(own=n(self.own).view(np.uint64), opp=n(self.opp).view(np.uint64), pi=n(self.pi), z=n(self.z),
                        mover=n(self.mover), act=n(self.act), game=n(self.game), ply=n(self.ply).astype(np.int32), size=self.size)

    @staticmethod
    def from_host(ex, device="cuda:0"):
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(device)  # noqa: E731
        return DeviceExamples(own=t(ex.own.view(np.int64)), opp=t(ex.opp.view(np.int64)), pi=t(ex.pi.astype(np.float32)),
                              z=t(ex.z.astype(np.int8)), mover=t(ex.mover.astype(np.int8)), act=t(ex.act.astype(np.uint8)),
                              game=t(np.asarray(ex.game, np.int64)), ply=t(np.asarray(ex.ply, np.int32)), size=ex.size)

    def states(self):
        """canonical boards [n, size, size] int8 on the device: +1 = side to move, -1 = opponent"""
        s = self.size
        stride = 3 if s == 3 else 8
        sh = torch.tensor([[stride * r + c for c in range(s)] for r in range(s)], dtype=torch.int64, device=self.own.device)
        a = ((self.own[:, None, None] >> sh) & 1).to(torch.int8)
        b = ((self.opp[:, None, None] >> sh) & 1).to(torch.int8)
        return a - b


def concat_device_examples(parts):
    cat = lambda f: torch.cat([getattr(p, f) for p in parts])  # noqa: E731
    return DeviceExamples(cat("own"), cat("opp"), cat("pi"), cat("z"), cat("mover"), cat("act"), cat("game"), cat("ply"),
                          parts[0].size)


MAX_SIMS, MAX_SIMS_REUSE = 8189, 2045  # BZ_ENGINE_MAX_SIMS / BZ_ENGINE_MAX_SIMS_REUSE (include/bz_abi.h)


def check_sims(sims, reuse_subtree=False):
    """the packed edge record holds node ids in 13 bits: refuse larger searches where they are asked for"""
    lim = MAX_SIMS_REUSE if reuse_subtree else MAX_SIMS
    if not 1 <= int(sims) <= lim:
        raise ValueError(f"sims must be in 1..{lim}" + (" with subtree reuse" if reuse_subtree else "") +
                         f" (got {sims}): the tree's packed edge record holds at most 8191 nodes per game "
                         "(BZ_ENGINE_MAX_SIMS / BZ_ENGINE_MAX_SIMS_REUSE in include/bz_abi.h)")


class SelfPlayEngine:
    def __init__(self, game, n_games, sims, evaluator="uniform", net=None, c_puct=1.5, temp_moves=0, openings=0,
                 seed=0, rounds=1, game_id_base=0, game_id_stride=None, device="cuda:0", stagger=0,
                 dirichlet_alpha=0.0, dirichlet_eps=0.0, reuse_subtree=False, ttt_lanes=0, eval_cache=True):
        """eval_cache (BZ_ENGINE_EVAL_CACHE, bz_abi.h): with a net evaluator, a leaf whose position was evaluated earlier in
        the same search shares that evaluation instead of running the net again; True / "carry" (the default) also takes
        evaluations from the slot's PREVIOUS search (BZ_ENGINE_EVAL_CACHE_CARRY: after a move, the played child's old subtree
        is re-created node for node by the new search), "search" only from the same search, False none.  Every result is bit
        for bit what it is without the cache (the net is a function of the position); counters()["n_cache_hits"] (of which
        "n_cache_hits_prev" from the previous search) says how often it fired.  Ignored for the synthetic / external
        evaluators and with reuse_subtree."""
        check_sims(sims, reuse_subtree)
        _lib.require_gpu()
        L = _lib.lib()
        self.game = _GAMES[game]
        self.device = torch.device(device)
        self._streams = {}  # every stream this engine's kernels were launched on (drained before the workspace dies)
        t_max = 9 if self.game == GAME_TTT else 64
        self.cfg = EngineCfg(self.game, n_games, sims, _EVALS[evaluator], c_puct, temp_moves, openings, rounds, t_max,
                             stagger, seed, game_id_base, n_games if game_id_stride is None else game_id_stride,
                             (_lib.ENGINE_REUSE_SUBTREE if reuse_subtree else 0) | (_lib.ENGINE_EVAL_CACHE if eval_cache else 0) |
                             (_lib.ENGINE_EVAL_CACHE_CARRY if eval_cache in (True, "carry") else 0),
                             dirichlet_alpha, dirichlet_eps, ttt_lanes)
        nbytes = L.bz_engine_workspace_bytes(C.byref(self.cfg))
        if nbytes < 0:
            raise RuntimeError(_lib.last_error())
        self.ws = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
        self._pad = (-self.ws.data_ptr()) & 255
        h = C.c_void_p()
        _lib.check(L.bz_engine_create(C.byref(self.cfg), self.ws.data_ptr() + self._pad, nbytes, C.byref(h)))
        self.h = h
        self.lay = EngineLayout()
        _lib.check(L.bz_engine_get_layout(self.h, C.byref(self.lay)))
        self.B, self.sims, self.rounds, self.na, self.t_max = n_games, sims, rounds, self.lay.na, t_max
        self.size = _SIZES[self.game]
        self.net = net
        if net is not None:
            _lib.check(L.bz_engine_set_net(self.h, net.h))

    # ---- views into the workspace
    def _view(self, off, dtype, shape):
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        o = self._pad + off
        return self.ws[o:o + n].view(dtype).view(*shape)

    def _stream(self):
        st = torch.cuda.current_stream(self.device)
        self._streams[st.cuda_stream] = st
        return st.cuda_stream

    def _call(self, fn, *args):
        with torch.cuda.device(self.device):
            _lib.check(fn(self.h, *args, self._stream()))

    # ---- ABI passthroughs
    def reset_games(self):
        self._call(_lib.lib().bz_engine_reset_games)

    def set_roots(self, own, opp, to_move):
        own = torch.as_tensor(np.asarray(own, dtype=np.uint64).view(np.int64)).to(self.device)
        opp = torch.as_tensor(np.asarray(opp, dtype=np.uint64).view(np.int64)).to(self.device)
        tm = torch.as_tensor(np.asarray(to_move, dtype=np.int8)).to(self.device)
        assert own.numel() == self.B
        self._call(_lib.lib().bz_engine_set_roots, own.data_ptr(), opp.data_ptr(), tm.data_ptr())
        torch.cuda.current_stream(self.device).synchronize()  # keep own/opp/tm alive until consumed

    def search(self):
        self._call(_lib.lib().bz_engine_search)

    def search_external(self, eval_fn):
        """One full search with a caller-supplied evaluator (engine built with evaluator="external"): eval_fn(own, opp,
        kind) gets the leaf positions as int64 CUDA tensors [B] (uint64 bit patterns, side-to-move canonical -- the input
        convention of the reference's AIPlayer, players.py:85) and kind (uint8 [B], 1 = needs evaluation) and returns
        (logits [B, NA] float32, value [B] float32) CUDA tensors; legality masking and the softmax happen in the expansion.
        Any torch module can sit here -- e.g. an MLP over the 9 tic-tac-toe cells like the reference's TicTacToeNet."""
        lb = self.leaf_buffers()

        def fill():
            lg, v = eval_fn(lb["own"], lb["opp"], lb["kind"])
            lb["logits"].copy_(lg.to(torch.float32).reshape(lb["logits"].shape))
            lb["value"].copy_(v.to(torch.float32).reshape(lb["value"].shape))
        self.root_begin(); fill(); self.expand_backup()
        self.root_noise()  # (no-op unless dirichlet_eps > 0) same place as in bz_engine_search
        for s in range(self.sims):
            self.select(s); fill(); self.expand_backup()

    def root_begin(self):
        self._call(_lib.lib().bz_engine_root_begin)

    def root_noise(self):
        """Dirichlet noise on the expanded roots' priors (no-op when dirichlet_eps == 0); step-API callers run it
        after the expand_backup() that follows root_begin() and before select(0)"""
        self._call(_lib.lib().bz_engine_root_noise)

    def select(self, sim_index):
        self._call(_lib.lib().bz_engine_select, sim_index)

    def evaluate(self):
        self._call(_lib.lib().bz_engine_evaluate)

    def expand_backup(self):
        self._call(_lib.lib().bz_engine_expand_backup)

    def play(self, restart=False):
        self._call(_lib.lib().bz_engine_play, int(restart))

    def reset_counters(self):
        self._call(_lib.lib().bz_engine_reset_counters)

    def status(self):
        a, f, e = C.c_int32(), C.c_int64(), C.c_int32()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().bz_engine_status(self.h, self._stream(), C.byref(a), C.byref(f), C.byref(e)))
        if e.value:
            names = [n for bit, n in ((1, "edge arena overflow"), (2, "terminal root"), (4, "example buffer overflow"),
                                      (8, "walk deeper than the path buffer"),
                                      (16, "the evaluator returned a non-finite logit or value")) if e.value & bit]
            raise RuntimeError(f"bz_engine error flags 0x{e.value:x}: " + ", ".join(names))
        return a.value, f.value

    def root_stats(self):
        self._call(_lib.lib().bz_engine_root_stats)
        shp = (self.B, self.na)
        N = self._view(self.lay.root_N, torch.int32, shp).cpu().numpy().view(np.uint32)
        W = self._view(self.lay.root_W, torch.float32, shp).cpu().numpy()
        P = self._view(self.lay.root_P, torch.float32, shp).cpu().numpy()
        return N, W, P

    def counters(self):
        self._call(_lib.lib().bz_engine_sum_counters)
        c = self._view(self.lay.counters, torch.int64, (24,)).cpu().numpy()
        return dict(zip(_lib.COUNTER_NAMES, (int(v) for v in c[:len(_lib.COUNTER_NAMES)])))

    # leaf buffers for BZ_EVAL_EXTERNAL callers (torch tensors aliasing the workspace)
    def leaf_buffers(self):
        return {"own": self._view(self.lay.leaf_own, torch.int64, (self.B,)),
                "opp": self._view(self.lay.leaf_opp, torch.int64, (self.B,)),
                "kind": self._view(self.lay.leaf_kind, torch.uint8, (self.B,)),
                "logits": self._view(self.lay.logits, torch.float32, (self.B, self.na)),
                "value": self._view(self.lay.value, torch.float32, (self.B,))}

    def positions(self):
        return (_u64(self._view(self.lay.g_own, torch.int64, (self.B,))),
                _u64(self._view(self.lay.g_opp, torch.int64, (self.B,))),
                self._view(self.lay.g_to_move, torch.int8, (self.B,)).cpu().numpy(),
                self._view(self.lay.g_state, torch.uint8, (self.B,)).cpu().numpy())

    # ---- example buffers
    def example_tensors(self):
        """fixed-capacity device tensors [rounds, B, t_max, ...] (what the all-gather ships)"""
        R, B, T = self.rounds, self.B, self.t_max
        return {"own": self._view(self.lay.ex_own, torch.int64, (R, B, T)),
                "opp": self._view(self.lay.ex_opp, torch.int64, (R, B, T)),
                "pi": self._view(self.lay.ex_pi, torch.float32, (R, B, T, self.na)),
                "z": self._view(self.lay.ex_z, torch.int8, (R, B, T)),
                "mover": self._view(self.lay.ex_mover, torch.int8, (R, B, T)),
                "act": self._view(self.lay.ex_act, torch.uint8, (R, B, T)),
                "len": self._view(self.lay.ex_len, torch.int32, (R, B)),
                "winner": self._view(self.lay.ex_winner, torch.int8, (R, B))}

    def example_block(self):
        """the ONE contiguous, self-describing byte range [ex_own .. ex_winner | 256-B header] of the
        workspace (uint8 view, no copy) -- what the iteration-end all-gather ships"""
        o = self._pad + self.lay.ex_begin
        return self.ws[o:o + self.lay.ex_bytes]

    def examples(self):
        """finished games' rows, compacted on the device; only the valid rows cross PCIe"""
        return unpack_example_block(self.example_block())

    def block_geometry(self):
        """host-side description of this engine's example block (what its 256-byte header says), so that a block of the
        same geometry -- this engine's, or a peer rank's copy of it after the all-gather -- can be unpacked on the
        device without reading the header back"""
        L = self.lay
        offs = [o - L.ex_begin for o in (L.ex_own, L.ex_opp, L.ex_pi, L.ex_z, L.ex_mover, L.ex_act, L.ex_len, L.ex_winner)]
        return {"B": self.B, "rounds": self.rounds, "t_max": self.t_max, "na": self.na, "game": self.game,
                "size": self.size, "offs": offs, "ex_bytes": int(L.ex_bytes)}

    def device_examples(self):
        """finished games' rows as DeviceExamples: nothing leaves the GPU"""
        return unpack_example_block_device(self.example_block(), self.block_geometry())

    def winners(self):
        t = self.example_tensors()
        return t["winner"].cpu().numpy(), t["len"].cpu().numpy()

    def run_iteration(self, max_plies=None):
        """play every slot's game to termination (one self-play iteration)"""
        self.reset_games()
        plies = 0
        limit = max_plies or (12 if self.game == GAME_TTT else 140)
        while True:
            self.search()
            self.play(False)
            plies += 1
            active, _ = self.status()
            if active == 0 or plies >= limit:
                return plies

    def pack_examples(self, out=None, cap_rows=None, append=False):
        """finished games' rows -> a packed example block (bz_abi.h "Packed examples": header + compacted rows in
        (round, slot, ply) order), by two kernels on the current stream; nothing is copied to the host.  `out`: a block
        from alloc_packed_block() (append=True: add this engine's rows behind those already in it)."""
        cap = int(cap_rows or self.rounds * self.B * self.t_max)
        if out is None:
            assert not append
            out = alloc_packed_block(self.na, cap, self.device)
        self._call(_lib.lib().bz_engine_pack_examples, out.data_ptr(), out.numel(), cap, int(append))
        return out

    def drain(self):
        """wait for every stream this engine's kernels were launched on"""
        for st in self._streams.values():
            st.synchronize()

    def __del__(self):
        try:
            self.drain()  # the caching allocator only knows the stream the workspace was allocated on
            _lib.lib().bz_engine_destroy(self.h)
        except Exception:
            pass


EX_MAGIC = 0x425A455841000002
_EX_FIELDS = (("own", torch.int64, 8), ("opp", torch.int64, 8), ("pi", torch.float32, 4), ("z", torch.int8, 1),
              ("mover", torch.int8, 1), ("act", torch.uint8, 1), ("len", torch.int32, 4), ("winner", torch.int8, 1))


def example_block_views(block):
    """uint8 example block (device or host tensor) -> (dict of [R,B,T,...] tensor views, meta dict).
    The block describes itself through its trailing 256-byte header (include/bz_abi.h)."""
    meta = block[-256:].cpu().numpy().view(np.uint64)
    if int(meta[0]) != EX_MAGIC:
        raise ValueError("not a betazero_amd example block (bad magic)")
    base, stride, B, R, T, na, game = (int(v) for v in meta[1:8])
    offs = [int(v) for v in meta[8:16]]
    shapes = {"own": (R, B, T), "opp": (R, B, T), "pi": (R, B, T, na), "z": (R, B, T), "mover": (R, B, T),
              "act": (R, B, T), "len": (R, B), "winner": (R, B)}
    out = {}
    for (name, dt, esz), off in zip(_EX_FIELDS, offs):
        n = int(np.prod(shapes[name])) * esz
        out[name] = block[off:off + n].view(dt).view(*shapes[name])
    return out, {"game_id_base": base, "game_id_stride": stride, "B": B, "rounds": R, "t_max": T, "na": na,
                 "game": game, "size": _SIZES[game]}


def build_example_block(arrays, game_id_base, game_id_stride, game, device="cpu"):
    """host-side constructor of an example block with the engine's exact layout (256-byte aligned
    arrays + header): `arrays` = dict of [R,B,T,...] tensors as example_tensors() returns them.
    Used to re-load saved examples and by the CPU rehearsal of the multi-GPU path."""
    R, B, T = arrays["own"].shape
    na = arrays["pi"].shape[-1]
    offs, off = [], 0
    for name, dt, esz in _EX_FIELDS:
        offs.append(off)
        off += (arrays[name].numel() * esz + 255) & ~255
    block = torch.zeros(off + 256, dtype=torch.uint8, device=device)
    for (name, dt, esz), o in zip(_EX_FIELDS, offs):
        a = arrays[name].to(device=device, dtype=dt).contiguous()
        block[o:o + a.numel() * esz] = a.view(torch.uint8).reshape(-1)
    meta = np.zeros(32, np.uint64)
    meta[0:8] = [EX_MAGIC, game_id_base, game_id_stride, B, R, T, na, _GAMES[game]]
    meta[8:16] = offs
    block[off:off + 256] = torch.from_numpy(meta.view(np.uint8).copy()).to(device)
    return block


def unpack_example_block(block):
    """example block -> compact Examples of the finished games; the row selection runs where the
    block lives (on the GPU for a device block), so only valid rows are copied to the host"""
    t, m = example_block_views(block)
    T = m["t_max"]
    valid = torch.arange(T, device=block.device)[None, None, :] < t["len"][:, :, None]
    r, b, k = valid.nonzero(as_tuple=True)
    pick = lambda a: a[r, b, k].cpu().numpy()  # noqa: E731
    gid = m["game_id_base"] + r.cpu().numpy().astype(np.int64) * m["game_id_stride"] + b.cpu().numpy()
    return Examples(own=pick(t["own"]).view(np.uint64), opp=pick(t["opp"]).view(np.uint64), pi=pick(t["pi"]),
                    z=pick(t["z"]), mover=pick(t["mover"]), act=pick(t["act"]), game=gid,
                    ply=k.cpu().numpy().astype(np.int32), size=m["size"])


def unpack_example_block_device(block, geom):
    """device example block -> DeviceExamples of the finished games, entirely on the device: the array views come from
    `geom` (SelfPlayEngine.block_geometry() of an engine with the same configuration -- every rank's engines are built
    alike), the game-id base / stride are read from the block's own header AS DEVICE SCALARS, the row selection is a
    device-side nonzero().  No example data is copied to the host: the only host visits are the 112-byte header check and
    the size read-back of nonzero()."""
    R, B, T, na = geom["rounds"], geom["B"], geom["t_max"], geom["na"]
    assert block.numel() == geom["ex_bytes"], "example block of another geometry"
    # the block's own header must say what `geom` says (a peer built with another configuration can have the same byte
    # size): magic and words 3..15 (B, rounds, t_max, NA, game, the 8 array offsets) compared on the device
    want = torch.tensor([EX_MAGIC, B, R, T, na, geom["game"]] + list(geom["offs"]), dtype=torch.int64)
    got = block[-256:].view(torch.int64)
    if not torch.equal(torch.cat([got[0:1], got[3:16]]).cpu(), want):
        raise ValueError("example block does not have the local engine's geometry (header words differ)")
    shapes = {"own": (R, B, T), "opp": (R, B, T), "pi": (R, B, T, na), "z": (R, B, T), "mover": (R, B, T),
              "act": (R, B, T), "len": (R, B), "winner": (R, B)}
    t = {}
    for (name, dt, esz), off in zip(_EX_FIELDS, geom["offs"]):
        n = int(np.prod(shapes[name])) * esz
        t[name] = block[off:off + n].view(dt).view(*shapes[name])
    meta = block[-256:].view(torch.int64)  # header words: [1] = game-id base, [2] = stride (device scalars)
    valid = torch.arange(T, device=block.device)[None, None, :] < t["len"][:, :, None]
    r, b, k = valid.nonzero(as_tuple=True)
    pick = lambda a: a[r, b, k]  # noqa: E731
    return DeviceExamples(own=pick(t["own"]), opp=pick(t["opp"]), pi=pick(t["pi"]), z=pick(t["z"]), mover=pick(t["mover"]),
                          act=pick(t["act"]), game=meta[1] + r * meta[2] + b, ply=k.to(torch.int32), size=geom["size"])


def concat_examples(parts):
    cat = lambda f: np.concatenate([getattr(p, f) for p in parts])  # noqa: E731
    return Examples(cat("own"), cat("opp"), cat("pi"), cat("z"), cat("mover"), cat("act"), cat("game"), cat("ply"),
                    parts[0].size)


# ---------------------------------------------------------------- packed example blocks (include/bz_abi.h)
PACKED_MAGIC = 0x425A50414B000001
_PK_FIELDS = (("own", torch.int64), ("opp", torch.int64), ("pi", torch.float32), ("game", torch.int64),
              ("z", torch.int8), ("mover", torch.int8), ("act", torch.uint8), ("ply", torch.uint8))


def packed_layout(na, cap_rows):
    """(byte offsets of the 8 arrays, total bytes) of a packed example block: 256-byte header, then own, opp, pi,
    game id, z, mover, act, ply -- each at a multiple of 256 bytes (== bz_examples_packed_bytes)"""
    offs, off = [], 256
    for esz in (8, 8, 4 * na, 8, 1, 1, 1, 1):
        offs.append(off)
        off += (cap_rows * esz + 255) & ~255
    return offs, off


def alloc_packed_block(na, cap_rows, device="cuda:0"):
    """an (uninitialised) packed example block: 1-D uint8 tensor, 256-byte aligned"""
    _, total = packed_layout(na, cap_rows)
    t = torch.empty(total + 256, dtype=torch.uint8, device=device)
    pad = (-t.data_ptr()) & 255
    return t[pad:pad + total]


def packed_block_header(block, strict=True):
    """the block's header as a dict (ONE 256-byte copy to the host when the block lives on a GPU).  strict: raise when
    the block overflowed (rows of finished games did not fit) -- unpacking such a block would silently lose games;
    strict=False only reports `dropped_rows` (bench.py's post-mortem of its own capacity estimate)."""
    h = block[:256].cpu().numpy().view(np.uint64)
    if int(h[0]) != PACKED_MAGIC:
        raise ValueError("not a betazero_amd packed example block (bad magic)")
    d = {"n_rows": int(h[1]), "n_games": int(h[2]), "cap_rows": int(h[3]), "na": int(h[4]), "game": int(h[5]),
         "dropped_rows": int(h[6]), "bytes": int(h[7]), "offs": [int(v) for v in h[8:16]]}
    if d["dropped_rows"] and strict:
        raise RuntimeError("packed example block overflow: " +
                           ("an engine of another geometry was appended" if d["dropped_rows"] == 2**64 - 1 else
                            f"{d['dropped_rows']} rows of finished games did not fit into cap_rows = {d['cap_rows']}"))
    return d


def _packed_views(block, h):
    n, na = h["n_rows"], h["na"]
    out = {}
    for (name, dt), off in zip(_PK_FIELDS, h["offs"]):
        per = na if name == "pi" else 1
        esz = torch.empty((), dtype=dt).element_size()
        v = block[off:off + n * per * esz].view(dt)
        out[name] = v.view(n, na) if name == "pi" else v
    return out


def unpack_packed_block_device(block):
    """packed example block on a GPU -> DeviceExamples (views of its first n_rows rows; one header read-back)"""
    h = packed_block_header(block)
    v = _packed_views(block, h)
    return DeviceExamples(own=v["own"], opp=v["opp"], pi=v["pi"], z=v["z"], mover=v["mover"], act=v["act"], game=v["game"],
                          ply=v["ply"].to(torch.int32), size=_SIZES[h["game"]])


def unpack_packed_block(block):
    """packed example block (device or host) -> host Examples; only the valid rows are copied"""
    h = packed_block_header(block)
    v = {k: t.cpu().numpy() for k, t in _packed_views(block, h).items()}
    return Examples(own=v["own"].view(np.uint64), opp=v["opp"].view(np.uint64), pi=v["pi"], z=v["z"], mover=v["mover"],
                    act=v["act"], game=v["game"], ply=v["ply"].astype(np.int32), size=_SIZES[h["game"]])


def build_packed_block(ex, cap_rows, game, device="cpu"):
    """host-side constructor of a packed block from Examples (saved examples, and the CPU rehearsal of the multi-GPU
    path): the same bytes bz_engine_pack_examples writes for these rows"""
    n, na = len(ex), int(ex.pi.shape[1])
    assert n <= cap_rows
    offs, total = packed_layout(na, cap_rows)
    block = torch.zeros(total, dtype=torch.uint8)
    hdr = np.zeros(32, np.uint64)
    hdr[0:8] = [PACKED_MAGIC, n, len(np.unique(ex.game)), cap_rows, na, _GAMES[game], 0, total]
    hdr[8:16] = offs
    block[:256] = torch.from_numpy(hdr.view(np.uint8).copy())
    src = {"own": ex.own.view(np.int64), "opp": ex.opp.view(np.int64), "pi": ex.pi.astype(np.float32),
           "game": np.asarray(ex.game, np.int64), "z": ex.z.astype(np.int8), "mover": ex.mover.astype(np.int8),
           "act": ex.act.astype(np.uint8), "ply": np.asarray(ex.ply).astype(np.uint8)}
    for (name, dt), off in zip(_PK_FIELDS, offs):
        a = torch.from_numpy(np.ascontiguousarray(src[name])).view(torch.uint8).reshape(-1)
        block[off:off + a.numel()] = a
    return block.to(device)


# ---------------------------------------------------------------- pipelined self-play
_PIPE_STREAMS, _PIPE_INFO, _PIPE_REJECTED = {}, {}, []


def stream_overlap_ratio(a, b, spin_us=300, reps=2):
    """bz_stream_overlap_probe on two torch streams: ~1.0 = they run side by side, ~2.0 = one waits for the other"""
    r = C.c_float()
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().bz_stream_overlap_probe(a.cuda_stream, b.cuda_stream, spin_us, reps, C.byref(r)))
    return float(r.value)


def pipeline_streams(device="cuda:0", n=2, max_tries=4):
    """the n HIP streams the pipelines of this process run on: created ONCE per (device, n) and reused by every
    PipelinedSelfPlay.  torch hands streams out round-robin from a pool and not every pair runs side by side on this
    stack (profiles/r03_torch_stream_pool_pairs.txt: the third and fourth stream of a process serialise -- 12 % of the
    headline), so a candidate set is accepted only when bz_stream_overlap_probe sees every two of its streams overlap
    (two 0.3-ms single-wave kernels behind a common event finish in ~0.3 ms, not ~0.6); otherwise the next set is
    tried and the best one seen is kept."""
    dev = torch.device(device)
    key = (str(dev), n)
    if key not in _PIPE_STREAMS:
        tried = []
        for _ in range(max_tries if n > 1 else 1):
            st = [torch.cuda.Stream(device=dev) for _ in range(n)]
            worst = max([stream_overlap_ratio(st[i], st[j]) for i in range(n) for j in range(i + 1, n)], default=1.0)
            tried.append((worst, st))
            if worst < 1.5:
                break
        best = min(range(len(tried)), key=lambda i: tried[i][0])
        _PIPE_REJECTED.extend(t[1] for i, t in enumerate(tried) if i != best)  # kept alive: torch would hand them out again
        _PIPE_STREAMS[key] = tried[best][1]
        _PIPE_INFO[key] = {"overlap_ratio_of_candidates": [round(t[0], 3) for t in tried], "picked": best}
    return _PIPE_STREAMS[key]


def pipeline_stream_info(device="cuda:0", n=2):
    """what the probe saw when pipeline_streams() chose its streams (None before the first call)"""
    return _PIPE_INFO.get((str(torch.device(device)), n))


class PipelinedSelfPlay:
    """The batched self-play of `n_games` concurrent games on one GPU as `pipelines` independent SelfPlayEngines of
    n_games / pipelines slots, each on its own HIP stream, sharing one net: the tree step of one pipeline overlaps the
    net launch of the other and their net launches fill each other's tail (the shape the headline is
    measured on).  Games keep their global ids (slot s of pipeline i = id base + offset_i + s), so the examples are
    the same rows whatever the number of pipelines.  Batched counterpart of collect_game_data,
    src/tic_tac_toe/SL/generate_training_games.py:25-38.

    Stream contract: work is issued behind whatever the caller's current stream holds at that moment (weights just
    updated there are seen); join() makes the caller's current stream wait for the pipelines without blocking the
    host, and everything that hands data out (status, counters, pack_examples, examples ...) joins first."""

    def __init__(self, game, n_games, sims, evaluator="uniform", net=None, pipelines=2, streams=None, game_id_base=0,
                 game_id_stride=None, device="cuda:0", run_ahead=16, **engine_kwargs):
        assert 1 <= pipelines <= n_games
        # simulations the host thread may queue ahead of the GPU (0 = unbounded: it then spins on the runtime's full queue,
        # 2 cores per rank against 0.18 -- profiles/r04_host_run_ahead.txt)
        self.run_ahead = run_ahead
        self.device = torch.device(device)
        self.sizes = [n_games // pipelines + (1 if i < n_games % pipelines else 0) for i in range(pipelines)]
        self.streams = list(streams) if streams is not None else pipeline_streams(self.device, pipelines)
        assert len(self.streams) == pipelines
        stride = n_games if game_id_stride is None else game_id_stride
        self.engines = [SelfPlayEngine(game, self.sizes[i], sims, evaluator, net, game_id_base=game_id_base + sum(self.sizes[:i]),
                                       game_id_stride=stride, device=device, **engine_kwargs) for i in range(pipelines)]
        e0 = self.engines[0]
        self.B, self.sims, self.na, self.t_max, self.rounds, self.size, self.game = n_games, sims, e0.na, e0.t_max, e0.rounds, e0.size, e0.game

    # ---- stream plumbing
    def _fork(self):
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            st.wait_stream(cur)

    def join(self):
        """the caller's current stream waits for everything issued on the pipelines so far (no host block)"""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)

    def sync(self):
        for st in self.streams:
            st.synchronize()

    def _each(self, fn):
        out = []
        for e, st in zip(self.engines, self.streams):
            with torch.cuda.stream(st):
                out.append(fn(e))
        return out

    # ---- the loop
    def reset_games(self):
        self._fork()
        self._each(lambda e: e.reset_games())

    def step(self, restart=False):
        """one move for every active slot of every pipeline: search (root expansion + sims simulations) + play, issued
        by bz_engines_step: ONE host thread, the pipelines interleaved simulation by simulation, at most `run_ahead`
        simulations ahead of the streams (it sleeps on blocking events instead of spinning on a full queue).  Returns
        with the tail of the move still queued."""
        self._fork()
        n = len(self.engines)
        for e, st in zip(self.engines, self.streams):
            e._streams[st.cuda_stream] = st
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().bz_engines_step((C.c_void_p * n)(*[e.h for e in self.engines]),
                                                  (C.c_void_p * n)(*[st.cuda_stream for st in self.streams]), n,
                                                  int(restart), int(self.run_ahead)))

    def status(self):
        """(active slots, finished games) over all pipelines; waits for them; raises on engine error flags"""
        r = self._each(lambda e: e.status())
        return sum(a for a, _ in r), sum(f for _, f in r)

    def run_iteration(self, max_plies=None):
        """play every slot's game to termination (one self-play iteration); returns the number of moves"""
        self.reset_games()
        plies, limit = 0, max_plies or (12 if self.game == GAME_TTT else 140)
        while True:
            self.step(False)
            plies += 1
            if self.status()[0] == 0 or plies >= limit:
                return plies

    def reset_counters(self):
        self._each(lambda e: e.reset_counters())

    def counters(self):
        tot = {}
        for c in self._each(lambda e: e.counters()):
            for k, v in c.items():
                tot[k] = tot.get(k, 0) + v
        return tot

    # ---- examples
    def example_blocks(self):
        """the engines' raw fixed-capacity example blocks (views into their workspaces)"""
        self.join()
        return [e.example_block() for e in self.engines]

    def packed_capacity(self, games=None):
        """rows a packed block needs for `games` finished games (default: every slot of every round)"""
        return int((games if games is not None else self.rounds * self.B) * self.t_max)

    def pack_examples(self, out=None, cap_rows=None):
        """ONE packed example block with the finished games of all pipelines (pipeline order, then round, slot, ply),
        written by the pack kernels on the caller's current stream behind the pipelines' work; no host visit."""
        cap = int(cap_rows or self.packed_capacity())
        if out is None:
            out = alloc_packed_block(self.na, cap, self.device)
        self.join()
        for i, e in enumerate(self.engines):
            e.pack_examples(out, cap, append=i > 0)
        return out

    def device_examples(self):
        return unpack_packed_block_device(self.pack_examples())

    def examples(self):
        return unpack_packed_block(self.pack_examples())

    def winners(self):
        self.join()
        w = [e.winners() for e in self.engines]
        return np.concatenate([a for a, _ in w], axis=1), np.concatenate([b for _, b in w], axis=1)

    def __del__(self):
        try:
            self.sync()
        except Exception:
            pass


def self_play(game, n_games, sims, net=None, seed=0, evaluator=None, temp_moves=0, openings=0, c_puct=1.5,
              device="cuda:0", game_id_base=0, game_id_stride=None, dirichlet_alpha=0.0, dirichlet_eps=0.0,
              reuse_subtree=False, pipelines=None):
    """Play n_games concurrent self-play games to the end on one GPU and return
    (s, pi, z): canonical states int8 [n, size, size], visit-count policies
    f32 [n, NA], outcomes for the mover int8 [n] -- plus the Examples object.
    With a net evaluator the games run as two pipelines on two HIP streams (PipelinedSelfPlay: the shape bench.py
    measures); `pipelines` overrides.  The rows do not depend on it."""
    if evaluator is None:
        evaluator = "net_bf16" if net is not None else "uniform"
    if pipelines is None:
        pipelines = 2 if (evaluator.startswith("net_") and n_games >= 2) else 1
    sp = PipelinedSelfPlay(game, n_games, sims, evaluator, net, pipelines, game_id_base=game_id_base,
                           game_id_stride=game_id_stride, device=device, c_puct=c_puct, temp_moves=temp_moves,
                           openings=openings, seed=seed, rounds=1, dirichlet_alpha=dirichlet_alpha,
                           dirichlet_eps=dirichlet_eps, reuse_subtree=reuse_subtree)
    sp.run_iteration()
    ex = sp.examples()
    return ex.states(), ex.pi, ex.z, ex
