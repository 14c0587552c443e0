"""ctypes wrapper around oracle/libbz_oracle.so  -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The product package (betazero_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("BZ_ORACLE_SO", os.path.join(_HERE, "libbz_oracle.so"))  # BZ_ORACLE_SO: e.g. the ASan build

GAME_TTT, GAME_REVERSI, GAME_REVERSI6, GAME_REVERSI4 = 0, 1, 2, 3
EVAL_UNIFORM, EVAL_HASH, EVAL_NET_F32, EVAL_NET_BF16, EVAL_NET_FP8 = 0, 1, 2, 3, 5
PASS = 64
COUNTER_NAMES = ("n_sims", "n_path_nodes", "n_child_scored", "n_edges_backed", "n_expanded",
                 "n_child_written", "n_env_steps", "n_net_leaves")


def build():
    subprocess.check_call(["make", "-C", _HERE, "libbz_oracle.so"], stdout=subprocess.DEVNULL)


class SpCfg(C.Structure):
    _fields_ = [("game", C.c_int), ("sims", C.c_int), ("eval_kind", C.c_int), ("temp_moves", C.c_int),
                ("openings", C.c_int), ("max_moves", C.c_int), ("c_puct", C.c_float), ("seed", C.c_uint64),
                ("flags", C.c_uint), ("dir_alpha", C.c_float), ("dir_eps", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u64, i32, f32p = C.c_uint64, C.c_int, C.POINTER(C.c_float)
        L.orc_reversi_legal.restype = u64
        L.orc_reversi_legal.argtypes = [u64, u64, i32]
        L.orc_reversi_apply.restype = i32
        L.orc_reversi_apply.argtypes = [u64, u64, i32, i32, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
        L.orc_reversi_game_over.restype = i32
        L.orc_reversi_game_over.argtypes = [u64, u64, i32]
        L.orc_reversi_score.restype = i32
        L.orc_reversi_score.argtypes = [u64, u64, C.POINTER(i32), C.POINTER(i32)]
        L.orc_ttt_game_over.restype = i32
        L.orc_ttt_game_over.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(i32)]
        L.orc_ttt_legal.restype = C.c_uint32
        L.orc_ttt_legal.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_ttt_apply.restype = i32
        L.orc_ttt_apply.argtypes = [C.c_uint32, C.c_uint32, i32, i32, C.POINTER(C.c_uint32)]
        L.orc_expf.restype = C.c_float
        L.orc_expf.argtypes = [C.c_float]
        L.orc_e4m3_round.restype = C.c_float
        L.orc_e4m3_round.argtypes = [C.c_float]
        L.orc_tanhf.restype = C.c_float
        L.orc_tanhf.argtypes = [C.c_float]
        L.orc_eval_hash.restype = None
        L.orc_eval_hash.argtypes = [u64, u64, i32, f32p, f32p]
        L.orc_rng.restype = u64
        L.orc_rng.argtypes = [u64, u64, u64]
        L.orc_net_param_count.restype = C.c_size_t
        L.orc_net_param_count.argtypes = [i32, i32, i32]
        L.orc_net_create.restype = C.c_void_p
        L.orc_net_create.argtypes = [i32, i32, i32, f32p]
        L.orc_net_destroy.restype = None
        L.orc_net_destroy.argtypes = [C.c_void_p]
        L.orc_net_forward.restype = None
        L.orc_net_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, i32, i32, C.c_void_p, C.c_void_p]
        L.orc_logf.restype = C.c_float
        L.orc_logf.argtypes = [C.c_float]
        L.orc_gamma.restype = C.c_float
        L.orc_gamma.argtypes = [C.c_float, u64, u64, u64, i32]
        L.orc_mcts_search_noise.restype = i32
        L.orc_mcts_search_noise.argtypes = [i32, u64, u64, i32, i32, i32, C.c_float, C.c_void_p, C.c_float, C.c_float, u64, u64,
                                            u64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mcts_search.restype = i32
        L.orc_mcts_search.argtypes = [i32, u64, u64, i32, i32, i32, C.c_float, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_selfplay_game.restype = i32
        L.orc_selfplay_game.argtypes = [C.POINTER(SpCfg), C.c_void_p, u64, i32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(i32), C.POINTER(i32),
                                        C.c_void_p]
        _lib = L
    return _lib


# ---------------------------------------------------------------- env
def reversi_legal(own, opp, size=8):
    return int(lib().orc_reversi_legal(own, opp, size))


def reversi_apply(own, opp, size, row, col):
    """-> (own_after, opp_after, flips) or None where the reference raises ValueError."""
    a, b, f = C.c_uint64(), C.c_uint64(), C.c_uint64()
    if lib().orc_reversi_apply(own, opp, size, row, col, C.byref(a), C.byref(b), C.byref(f)):
        return None
    return a.value, b.value, f.value


def reversi_game_over(a, b, size=8):
    return bool(lib().orc_reversi_game_over(a, b, size))


def reversi_score(x, o):
    nx, no = C.c_int(), C.c_int()
    w = lib().orc_reversi_score(x, o, C.byref(nx), C.byref(no))
    return w, (nx.value, no.value)


def ttt_game_over(x, o):
    w = C.c_int()
    over = lib().orc_ttt_game_over(x, o, C.byref(w))
    return bool(over), (w.value if over else None)


def ttt_legal(x, o):
    return int(lib().orc_ttt_legal(x, o))


def ttt_apply(own, opp, row, col):
    a = C.c_uint32()
    if lib().orc_ttt_apply(own, opp, row, col, C.byref(a)):
        return None
    return a.value


def expf(x):
    return np.float32(lib().orc_expf(float(np.float32(x))))


def tanhf(x):
    return np.float32(lib().orc_tanhf(float(np.float32(x))))


def eval_hash(own, opp, na):
    lg = np.zeros(na, np.float32)
    v = C.c_float()
    lib().orc_eval_hash(own, opp, na, lg.ctypes.data_as(C.POINTER(C.c_float)), C.byref(v))
    return lg, np.float32(v.value)


def rng(seed, gid, ply):
    return int(lib().orc_rng(seed, gid, ply))


# ---------------------------------------------------------------- net
class Net:
    """flat fp32 parameter vector in the order documented in bz_oracle.c"""

    def __init__(self, C_, NB, VH, params):
        params = np.ascontiguousarray(params, dtype=np.float32)
        assert params.size == lib().orc_net_param_count(C_, NB, VH), (params.size, C_, NB, VH)
        self.h = lib().orc_net_create(C_, NB, VH, params.ctypes.data_as(C.POINTER(C.c_float)))
        self.C, self.NB, self.VH = C_, NB, VH

    def forward(self, own, opp, bf16=False):
        """bf16: False/0 = fp32, True/1 = bf16 emulation, 2 = fp8 (e4m3) emulation"""
        own = np.ascontiguousarray(own, dtype=np.uint64)
        opp = np.ascontiguousarray(opp, dtype=np.uint64)
        n = own.size
        lg = np.zeros((n, 65), np.float32)
        v = np.zeros(n, np.float32)
        lib().orc_net_forward(self.h, own.ctypes.data, opp.ctypes.data, n, int(bf16), lg.ctypes.data, v.ctypes.data)
        return lg, v

    def __del__(self):
        try:
            lib().orc_net_destroy(self.h)
        except Exception:
            pass


# ---------------------------------------------------------------- mcts / self-play
def logf(x):
    return np.float32(lib().orc_logf(float(np.float32(x))))


def gamma(alpha, seed, gid, ply, edge):
    return np.float32(lib().orc_gamma(float(np.float32(alpha)), seed, gid, ply, edge))


def _counters(cnt):
    """the oracle's 8 work counters under the engine's names; the oracle has no evaluation cache (every leaf is evaluated),
    so the engine's ninth counter reads 0 here"""
    d = dict(zip(COUNTER_NAMES, (int(c) for c in cnt)))
    d["n_cache_hits"] = d["n_cache_hits_prev"] = 0
    return d


def mcts_search(game, own, opp, to_move, sims, eval_kind, c_puct=1.5, net=None, dir_alpha=0.0, dir_eps=0.0, seed=0,
                gid=0, ply=0):
    na = 9 if game == GAME_TTT else 65
    N = np.zeros(na, np.uint32)
    W = np.zeros(na, np.float32)
    P = np.zeros(na, np.float32)
    cnt = np.zeros(8, np.uint64)
    rc = lib().orc_mcts_search_noise(game, own, opp, to_move, sims, eval_kind, c_puct, net.h if net else None,
                                     dir_alpha, dir_eps, seed, gid, ply,
                                     N.ctypes.data, W.ctypes.data, P.ctypes.data, cnt.ctypes.data)
    if rc:
        raise ValueError("terminal root")
    return N, W, P, _counters(cnt)


def mcts_search_nodes(game, own, opp, to_move, sims, eval_kind, c_puct=1.5, net=None):
    """diagnostic: (own[n], opp[n], terminal[n]) of every node one search created, in creation order (node 0 = root)"""
    cap = sims + 2
    o, p, t = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64), np.zeros(cap, np.uint8)
    L = lib()
    L.orc_mcts_search_nodes.restype = C.c_int
    L.orc_mcts_search_nodes.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
    n = L.orc_mcts_search_nodes(game, int(own), int(opp), to_move, sims, eval_kind, c_puct, net.h if net else None, cap,
                                o.ctypes.data, p.ctypes.data, t.ctypes.data)
    if n < 0:
        raise ValueError("terminal root")
    return o[:n], p[:n], t[:n]


def mcts_search_walkstats(game, own, opp, to_move, sims, eval_kind, c_puct=1.5, net=None):
    """diagnostic: per-level statistics of one search's walks (bz_oracle.c orc_mcts_search_walkstats)"""
    out = np.zeros(7, np.uint64)
    L = lib()
    L.orc_mcts_search_walkstats.restype = C.c_int
    L.orc_mcts_search_walkstats.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    if L.orc_mcts_search_walkstats(game, int(own), int(opp), to_move, sims, eval_kind, c_puct, net.h if net else None, out.ctypes.data) < 0:
        raise ValueError("terminal root")
    k = ("sims", "levels", "fav_hits", "levels_below_root", "fav_hits_below_root", "round_trips_with_prefetch")
    return {n: int(out[i]) for i, n in enumerate(k)}


def selfplay_game(game, gid, sims, eval_kind, temp_moves=0, openings=0, seed=0, c_puct=1.5, net=None,
                  max_moves=0, dir_alpha=0.0, dir_eps=0.0, reuse=False):
    na = 9 if game == GAME_TTT else 65
    tmax = 16 if game == GAME_TTT else 64
    cfg = SpCfg(game, sims, eval_kind, temp_moves, openings, max_moves, c_puct, seed, 1 if reuse else 0, dir_alpha, dir_eps)
    own = np.zeros(tmax, np.uint64)
    opp = np.zeros(tmax, np.uint64)
    pi = np.zeros((tmax, na), np.float32)
    mover = np.zeros(tmax, np.int8)
    act = np.zeros(tmax, np.uint8)
    cnt = np.zeros(8, np.uint64)
    w, p = C.c_int(), C.c_int()
    n = lib().orc_selfplay_game(C.byref(cfg), net.h if net else None, gid, tmax, own.ctypes.data, opp.ctypes.data,
                                pi.ctypes.data, mover.ctypes.data, act.ctypes.data, C.byref(w), C.byref(p),
                                cnt.ctypes.data)
    z = (w.value * mover[:n]).astype(np.int8) if w.value in (-1, 0, 1) else np.zeros(n, np.int8)
    return {"own": own[:n], "opp": opp[:n], "pi": pi[:n], "mover": mover[:n], "act": act[:n], "z": z,
            "winner": w.value, "passes": p.value, "counters": _counters(cnt)}
