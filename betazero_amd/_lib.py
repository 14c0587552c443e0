"""ctypes binding of libbz_hip.so (include/bz_abi.h).  No CPU fallback: if the
library is missing this raises; batched calls need a HIP device."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# BZ_HIP_SO: load a diagnostic variant built by betazero_amd.build.build_variant() instead of the
# product library (needs BZ_ALLOW_EXPERIMENT=1 as well: such builds may time but not compute)
SO = os.environ.get("BZ_HIP_SO") or os.path.join(HERE, "libbz_hip.so")
ABI_VERSION = 6

BZ_OK, BZ_EINVAL, BZ_EILLEGAL_MOVE, BZ_EHIP, BZ_ENOMEM, BZ_ENOGPU, BZ_ESTATE = range(7)
GAME_TTT, GAME_REVERSI, GAME_REVERSI6, GAME_REVERSI4 = 0, 1, 2, 3
EVAL_UNIFORM, EVAL_HASH, EVAL_NET_F32, EVAL_NET_BF16, EVAL_EXTERNAL, EVAL_NET_FP8 = range(6)
ST_RUNNING, ST_TERMINAL, ST_ILLEGAL, ST_MUST_PASS = range(4)
PASS_ACTION = 64
ENGINE_REUSE_SUBTREE = 1
ENGINE_EVAL_CACHE = 2   # BZ_ENGINE_EVAL_CACHE: a position met again inside one search shares its first evaluation
ENGINE_EVAL_CACHE_CARRY = 4   # BZ_ENGINE_EVAL_CACHE_CARRY: ... and the previous search's evaluations serve the next search too
PROF_SLOTS = ("tower", "stem", "heads", "select", "expand_backup", "search_fused", "play", "env_step")
COUNTER_NAMES = ("n_sims", "n_path_nodes", "n_child_scored", "n_edges_backed", "n_expanded",
                 "n_child_written", "n_env_steps", "n_net_leaves", "n_cache_hits", "n_cache_hits_prev")

u64, u32, i32, i64, vp = C.c_uint64, C.c_uint32, C.c_int32, C.c_int64, C.c_void_p


class EngineCfg(C.Structure):
    _fields_ = [("game", i32), ("n_games", i32), ("sims", i32), ("eval_kind", i32), ("c_puct", C.c_float),
                ("temp_moves", i32), ("openings", i32), ("rounds", i32), ("t_max", i32), ("stagger", i32),
                ("seed", u64), ("game_id_base", u64), ("game_id_stride", u64),
                ("flags", u32), ("dirichlet_alpha", C.c_float), ("dirichlet_eps", C.c_float), ("ttt_lanes", i32)]


class EngineLayout(C.Structure):
    _fields_ = [(n, i64) for n in ("ex_own", "ex_opp", "ex_pi", "ex_z", "ex_mover", "ex_act", "ex_len", "ex_winner",
                                   "root_N", "root_W", "root_P", "leaf_own", "leaf_opp", "leaf_kind", "logits",
                                   "value", "g_own", "g_opp", "g_to_move", "g_state", "counters")] + \
               [("na", i32), ("t_max", i32)] + [(n, i64) for n in ("ex_begin", "ex_bytes", "ex_meta")]


class TrainHeadParams(C.Structure):   # bz_train_head_params: one fp32 device pointer per head parameter tensor
    _fields_ = [(n, vp) for n in ("pol_w", "pol_b", "polfc_w", "polfc_b", "val_w", "val_b", "v1_w", "v1_b", "v2_w", "v2_b")]


class TrainTensors(C.Structure):      # bz_train_tensors: one fp32 device pointer per parameter tensor (gradients / parameters / an Adam moment)
    _fields_ = [(n, vp) for n in ("stem_w", "stem_b", "tower_w", "tower_b", "pol_w", "pol_b", "polfc_w", "polfc_b", "val_w", "val_b",
                                  "v1_w", "v1_b", "v2_w", "v2_b")]


class TrainBatch(C.Structure):        # bz_train_batch (the kernels read it from DEVICE memory)
    _fields_ = [("own", vp), ("opp", vp), ("pi", vp), ("z", vp), ("idx", vp), ("n_rows", i64)]


class TrainAdam(C.Structure):         # bz_train_adam
    _fields_ = [("hyper", vp), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("p", TrainTensors), ("m", TrainTensors), ("v", TrainTensors)]


class TrainPartials(C.Structure):     # bz_train_partials
    _fields_ = [(n, vp) for n in ("tower", "tower_b", "stem", "heads", "heads_w")] + [("splits", i32)]


_SIGS = {
    "bz_abi_version": (i32, []),
    "bz_last_error": (C.c_char_p, []),
    "bz_build_info": (C.c_char_p, []),
    "bz_device_count": (i32, []),
    "bz_reversi_legal": (i32, [u64, u64, i32, C.POINTER(u64)]),
    "bz_reversi_apply": (i32, [u64, u64, i32, i32, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
    "bz_reversi_game_over": (i32, [u64, u64, i32, C.POINTER(i32)]),
    "bz_reversi_score": (i32, [u64, u64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "bz_ttt_legal": (i32, [u32, u32, C.POINTER(u32)]),
    "bz_ttt_apply": (i32, [u32, u32, i32, i32, C.POINTER(u32)]),
    "bz_ttt_game_over": (i32, [u32, u32, C.POINTER(i32), C.POINTER(i32)]),
    "bz_reversi_step_batch": (i32, [vp, vp, vp, i64, vp, vp, vp, vp, vp, vp]),
    "bz_reversi_step_batch_sized": (i32, [vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp]),
    "bz_reversi_legal_batch": (i32, [vp, vp, i64, vp, vp]),
    "bz_reversi_score_batch": (i32, [vp, vp, i64, vp, vp, vp]),
    "bz_ttt_step_batch": (i32, [vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp]),
    "bz_reversi_minimax": (i32, [u64, u64, i32, i32, C.POINTER(i32), C.POINTER(i32)]),
    "bz_ttt_minimax": (i32, [u32, u32, i32, C.POINTER(i32), C.POINTER(i32)]),
    "bz_reversi_minimax_batch": (i32, [vp, vp, vp, i64, i32, i32, vp, vp, vp]),
    "bz_ttt_minimax_batch": (i32, [vp, vp, vp, vp, i64, vp, vp, vp]),
    "bz_augment_d4_batch": (i32, [vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, vp]),
    "bz_net_update": (i32, [vp, vp, vp]),
    "bz_net_param_count": (i64, [i32, i32, i32]),
    "bz_net_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "bz_net_create": (i32, [i32, i32, i32, i32, vp, vp, i64, vp, C.POINTER(vp)]),
    "bz_net_destroy": (i32, [vp]),
    "bz_net_forward_f32": (i32, [vp, vp, vp, i32, vp, vp, vp]),
    "bz_net_forward_bf16": (i32, [vp, vp, vp, i32, vp, vp, vp]),
    "bz_net_forward_fp8": (i32, [vp, vp, vp, i32, vp, vp, vp]),
    "bz_engine_workspace_bytes": (i64, [C.POINTER(EngineCfg)]),
    "bz_engine_create": (i32, [C.POINTER(EngineCfg), vp, i64, C.POINTER(vp)]),
    "bz_engine_destroy": (i32, [vp]),
    "bz_engine_get_layout": (i32, [vp, C.POINTER(EngineLayout)]),
    "bz_engine_set_net": (i32, [vp, vp]),
    "bz_engine_debug_set_search_seq": (i32, [vp, C.c_uint32]),
    "bz_engine_reset_games": (i32, [vp, vp]),
    "bz_engine_set_roots": (i32, [vp, vp, vp, vp, vp]),
    "bz_engine_search": (i32, [vp, vp]),
    "bz_engine_root_begin": (i32, [vp, vp]),
    "bz_engine_select": (i32, [vp, u32, vp]),
    "bz_engine_evaluate": (i32, [vp, vp]),
    "bz_engine_expand_backup": (i32, [vp, vp]),
    "bz_engine_root_noise": (i32, [vp, vp]),
    "bz_engine_root_stats": (i32, [vp, vp]),
    "bz_engine_play": (i32, [vp, i32, vp]),
    "bz_engine_status": (i32, [vp, vp, C.POINTER(i32), C.POINTER(i64), C.POINTER(i32)]),
    "bz_mcts_select": (i32, [vp, u32, vp]),
    "bz_mcts_expand_backup": (i32, [vp, vp]),
    "bz_selfplay_run": (i32, [vp, i32, vp]),
    "bz_engines_step": (i32, [C.POINTER(vp), C.POINTER(vp), i32, i32, i32]),
    "bz_engine_reset_counters": (i32, [vp, vp]),
    "bz_engine_sum_counters": (i32, [vp, vp]),
    "bz_examples_packed_bytes": (i64, [i32, i64]),
    "bz_engine_pack_examples": (i32, [vp, vp, i64, i64, i32, vp]),
    "bz_stream_overlap_probe": (i32, [vp, vp, i32, i32, C.POINTER(C.c_float)]),
    "bz_train_wf_bytes": (i64, [i32, i32]),
    "bz_train_positions_per_workgroup": (i32, [i32]),
    "bz_train_mask_bytes": (i64, [i32, i32, i32]),
    "bz_train_pack_weights": (i32, [vp, i32, i32, vp, vp, vp]),
    "bz_train_tower_fwd": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "bz_train_tower_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "bz_train_wgrad_splits": (i32, [i32, i32, i32]),
    "bz_train_wgrad_bias_rows": (i32, [i32, i32]),
    "bz_train_wgrad": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "bz_train_ends_sizes": (i32, [i32, i32, C.POINTER(i32)]),
    "bz_train_stem_fwd": (i32, [vp, i32, vp, vp, i32, vp, vp]),
    "bz_train_stem_wgrad": (i32, [vp, vp, vp, i32, i32, vp, vp]),
    "bz_train_heads": (i32, [vp, vp, i32, i32, i32, C.POINTER(TrainHeadParams), vp, vp, vp, vp, vp, vp]),
    "bz_train_heads_wgrad": (i32, [vp, vp, vp, i32, i32, vp, vp]),
    "bz_train_finish": (i32, [C.POINTER(TrainPartials), C.POINTER(TrainTensors), i32, i32, i32, i32, vp, C.POINTER(TrainAdam), vp]),
    "bz_profile_enable": (i32, [i32]),
    "bz_profile_reserve": (i32, [i32, i64]),
    "bz_profile_read": (i32, [i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_double)]),
    "bz_profile_reset": (i32, []),
    "bz_profile_intervals": (i32, [i32, vp, vp, i64, C.POINTER(i64)]),
}
ABI_SYMBOLS = tuple(_SIGS)

_lib = None


def lib():
    """Load libbz_hip.so; raise (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise RuntimeError(f"{SO} is missing: build it with `python -m betazero_amd.build` "
                               "(there is no CPU fallback for the HIP engine)")
        # torch first: its wheel bundles a HIP runtime (libamdhip64.so.7); loading it before
        # ours makes the dynamic linker bind libbz_hip.so to that same copy (same SONAME), so
        # the process has ONE HIP runtime and torch tensors / streams are valid in our kernels.
        import torch  # noqa: F401
        L = C.CDLL(SO)
        # the version first: a stale library must fail with the rebuild message, not with an AttributeError on
        # whichever newer symbol it happens to lack
        ver = getattr(L, "bz_abi_version", None)
        L_ver = None
        if ver is not None:
            ver.restype, ver.argtypes = _SIGS["bz_abi_version"]
            L_ver = ver()
        if L_ver != ABI_VERSION:
            raise RuntimeError(f"{SO}: ABI version {L_ver}, expected {ABI_VERSION} -- rebuild it "
                               "(python -m betazero_amd.build)")
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        info = (L.bz_build_info() or b"").decode()
        if info != "product" and os.environ.get("BZ_ALLOW_EXPERIMENT") != "1":
            raise RuntimeError(f"{SO} was built with [{info}], not the product flags; refusing to load it "
                               "(set BZ_ALLOW_EXPERIMENT=1 for a diagnostic run)")
        _lib = L
    return _lib


def last_error():
    return (lib().bz_last_error() or b"").decode()


def check(rc):
    if rc == BZ_OK:
        return
    msg = last_error()
    if rc == BZ_EILLEGAL_MOVE:
        raise ValueError("Invalid move")  # reversi_board.py:45, tic_tac_toe_board.py:25
    raise RuntimeError(f"libbz_hip error {rc}: {msg}")


def require_gpu():
    if lib().bz_device_count() <= 0:
        raise RuntimeError("betazero_amd: no HIP device visible; the batched engine has no CPU fallback")


def profile_read():
    """{slot: (launches, timed, total_ms)} from the in-library HIP-event timers"""
    out = {}
    for i, name in enumerate(PROF_SLOTS):
        a, b, t = i64(), i64(), C.c_double()
        check(lib().bz_profile_read(i, C.byref(a), C.byref(b), C.byref(t)))
        out[name] = (a.value, b.value, t.value)
    return out


def profile_union_ms(slot_name, cap=1 << 18):
    """(union busy time, summed duration) in ms of a slot's timed launches (they may overlap across streams)"""
    import numpy as np
    st = np.zeros(cap, np.float64)
    en = np.zeros(cap, np.float64)
    n = i64()
    check(lib().bz_profile_intervals(PROF_SLOTS.index(slot_name), st.ctypes.data, en.ctypes.data, cap, C.byref(n)))
    st, en = st[:n.value], en[:n.value]
    order = np.argsort(st)
    union, cur_s, cur_e = 0.0, None, None
    for a, b in zip(st[order], en[order]):
        if cur_e is None or a > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    if cur_e is not None:
        union += cur_e - cur_s
    return union, float((en - st).sum())
