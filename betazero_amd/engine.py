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


class SelfPlayEngine:
    def __init__(self, game, n_games, sims, evaluator="uniform", net=None, c_puct=1.5, temp_moves=0, openings=0,
                 seed=0, rounds=1, game_id_base=0, game_id_stride=None, device="cuda:0", stagger=0,
                 dirichlet_alpha=0.0, dirichlet_eps=0.0, reuse_subtree=False, ttt_lanes=0):
        _lib.require_gpu()
        L = _lib.lib()
        self.game = _GAMES[game]
        self.device = torch.device(device)
        t_max = 9 if self.game == GAME_TTT else 64
        self.cfg = EngineCfg(self.game, n_games, sims, _EVALS[evaluator], c_puct, temp_moves, openings, rounds, t_max,
                             stagger, seed, game_id_base, n_games if game_id_stride is None else game_id_stride,
                             _lib.ENGINE_REUSE_SUBTREE if reuse_subtree else 0, dirichlet_alpha, dirichlet_eps, ttt_lanes)
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
        return torch.cuda.current_stream(self.device).cuda_stream

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
        c = self._view(self.lay.counters, torch.int64, (16,)).cpu().numpy()
        return dict(zip(_lib.COUNTER_NAMES, (int(v) for v in c[:8])))

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

    def __del__(self):
        try:
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
    device-side nonzero().  No byte of the block visits the host."""
    R, B, T, na = geom["rounds"], geom["B"], geom["t_max"], geom["na"]
    assert block.numel() == geom["ex_bytes"], "example block of another geometry"
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


def self_play(game, n_games, sims, net=None, seed=0, evaluator=None, temp_moves=0, openings=0, c_puct=1.5,
              device="cuda:0", game_id_base=0, game_id_stride=None, dirichlet_alpha=0.0, dirichlet_eps=0.0,
              reuse_subtree=False):
    """Play n_games concurrent self-play games to the end on one GPU and return
    (s, pi, z): canonical states int8 [n, size, size], visit-count policies
    f32 [n, NA], outcomes for the mover int8 [n] -- plus the Examples object."""
    if evaluator is None:
        evaluator = "net_bf16" if net is not None else "uniform"
    eng = SelfPlayEngine(game, n_games, sims, evaluator, net, c_puct, temp_moves, openings, seed, 1, game_id_base,
                         game_id_stride, device, dirichlet_alpha=dirichlet_alpha, dirichlet_eps=dirichlet_eps,
                         reuse_subtree=reuse_subtree)
    eng.run_iteration()
    ex = eng.examples()
    return ex.states(), ex.pi, ex.z, ex
