"""Batched arena: B concurrent games of the GPU MCTS player against the reference's minimax player
 -- the only in-repo yard-stick of playing strength.

Reference anchors: the turn loop with its pass rule is ReversiTerminal.play (reversi_terminal.py:16-38)
resp. TicTacToeHeadless.play (tic_tac_toe.py:13-34); the opponents are OptimalPlayer
(src/reversi/players/reversi_players.py:35-77, depth-limited) and OptimalPlayer (src/tic_tac_toe/players.py:30-70,
full depth), both as gfx950 kernels (one game per lane, csrc/bz_arena.hip).  Every ply of all games is three
launches: one MCTS search for the games whose mover is the MCTS side, one minimax kernel for the others, one
batched env step for all.  torch holds the state tensors; nothing is computed on the host except the
random choices the reference makes with `random` (opening move of the TTT minimax, its Reversi fallback)."""
import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .engine import SelfPlayEngine


@dataclass
class ArenaResult:
    winner: np.ndarray       # absolute winner per game: +1 / -1 / 0
    mcts_colour: np.ndarray  # colour the MCTS side played in that game
    plies: np.ndarray        # moves played (passes not counted)
    moves: list              # per ply: (actions uint8 [B] (255 = no move this ply), mover int8 [B])

    @property
    def score(self):
        """outcome for the MCTS side per game: +1 win, 0 draw, -1 loss"""
        return self.winner * self.mcts_colour

    def summary(self):
        s = self.score
        return {"games": int(len(s)), "wins": int((s > 0).sum()), "draws": int((s == 0).sum()), "losses": int((s < 0).sum())}


def play_arena(game, n_games, sims, opponent_depth=4, evaluator="uniform", net=None, c_puct=1.5, seed=0, size=8,
               device="cuda:0", max_plies=200, opening_plies=0):
    """MCTS (`sims` simulations, `evaluator`) vs minimax for n_games concurrent games; the MCTS side plays X
    (moves first) in the even-numbered games and O in the odd ones.  game: "ttt" | "reversi" (size 8, 6 or 4).
    Both players are deterministic, so without help there are only two distinct games (one per colour):
    `opening_plies` > 0 plays that many uniformly random legal moves (seeded) before the players take over, which
    makes the B games B different tests."""
    if game != "ttt" and not 0 <= int(opponent_depth) <= 8:
        raise ValueError(f"play_arena: opponent_depth must be in 0..8 (got {opponent_depth}): the minimax kernel keeps an "
                         "explicit stack of that depth")
    _lib.require_gpu()
    L = _lib.lib()
    dev = torch.device(device)
    ttt = game == "ttt"
    B = n_games
    ename = "ttt" if ttt else {8: "reversi", 6: "reversi6", 4: "reversi4"}[size]
    eng = SelfPlayEngine(ename, B, sims, evaluator, net, c_puct, device=device)
    rng = np.random.default_rng(seed)
    st = lambda: torch.cuda.current_stream(dev).cuda_stream  # noqa: E731
    if ttt:
        own = torch.zeros(B, dtype=torch.int16, device=dev)
        opp = torch.zeros(B, dtype=torch.int16, device=dev)
    else:
        p = size // 2 - 1
        x0 = (1 << (8 * p + p)) | (1 << (8 * (p + 1) + p + 1))
        o0 = (1 << (8 * p + p + 1)) | (1 << (8 * (p + 1) + p))
        own = torch.full((B,), x0, dtype=torch.int64, device=dev)
        opp = torch.full((B,), o0, dtype=torch.int64, device=dev)
    to_move = torch.ones(B, dtype=torch.int8, device=dev)
    mcts_colour = torch.as_tensor(np.where(np.arange(B) % 2 == 0, 1, -1).astype(np.int8)).to(dev)
    active = torch.ones(B, dtype=torch.bool, device=dev)
    winner = torch.zeros(B, dtype=torch.int8, device=dev)
    plies = torch.zeros(B, dtype=torch.int32, device=dev)
    mm_move = torch.empty(B, dtype=torch.int8, device=dev)
    mm_score = torch.empty(B, dtype=torch.int16, device=dev)
    wdt = torch.int16 if ttt else torch.int64
    own_n, opp_n, legal_n = (torch.empty(B, dtype=wdt, device=dev) for _ in range(3))
    status = torch.empty(B, dtype=torch.uint8, device=dev)
    win_n = torch.empty(B, dtype=torch.int8, device=dev)
    log = []
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    valid = (1 << 9) - 1 if ttt else sum(((1 << size) - 1) << (8 * r) for r in range(size))
    valid_t = torch.tensor(valid - (1 << 64) if valid >= 1 << 63 else valid, dtype=torch.int64, device=dev)
    shifts = torch.arange(64, dtype=torch.int64, device=dev)
    for ply in range(max_plies):
        if not bool(active.any()):
            break
        mcts_turn = active & (to_move == mcts_colour)
        mm_turn = active & ~mcts_turn
        action = torch.full((B,), 255, dtype=torch.uint8, device=dev)
        if ply < opening_plies:  # a random legal move for every game, by neither player
            if ttt:
                legal = ~(own | opp).to(torch.int64) & valid_t
            else:
                legal = torch.empty(B, dtype=torch.int64, device=dev)
                with torch.cuda.device(dev):
                    _lib.check(L.bz_reversi_legal_batch(own.data_ptr(), opp.data_ptr(), B, legal.data_ptr(), st()))
                legal = legal & valid_t
            bits = ((legal.unsqueeze(1) >> shifts) & 1).bool()
            draw = torch.rand((B, 64), generator=gen, device=dev)
            pick = torch.where(bits, draw, torch.full_like(draw, -1.0)).argmax(1).to(torch.uint8)
            action = torch.where(active, pick, action)
            mcts_turn = mm_turn = torch.zeros_like(active)
        if bool(mcts_turn.any()):
            roots_tm = torch.where(mcts_turn, to_move, torch.zeros_like(to_move))
            o64 = own.to(torch.int64) if ttt else own
            p64 = opp.to(torch.int64) if ttt else opp
            with torch.cuda.device(dev):
                _lib.check(L.bz_engine_set_roots(eng.h, o64.data_ptr(), p64.data_ptr(), roots_tm.data_ptr(), st()))
            eng.search()
            eng._call(L.bz_engine_root_stats)
            N = eng._view(eng.lay.root_N, torch.int32, (B, eng.na))
            pick = N.argmax(1).to(torch.uint8)  # first maximum = lowest action (MCTSPlayer's rule)
            eng.status()
            action = torch.where(mcts_turn, pick, action)
        if bool(mm_turn.any()):
            act8 = mm_turn.to(torch.uint8)
            with torch.cuda.device(dev):
                if ttt:
                    _lib.check(L.bz_ttt_minimax_batch(own.data_ptr(), opp.data_ptr(), to_move.data_ptr(), act8.data_ptr(), B,
                                                      mm_move.data_ptr(), mm_score.data_ptr(), st()))
                else:
                    _lib.check(L.bz_reversi_minimax_batch(own.data_ptr(), opp.data_ptr(), act8.data_ptr(), B, size,
                                                          opponent_depth, mm_move.data_ptr(), mm_score.data_ptr(), st()))
            mv = mm_move.clone()
            need = mm_turn & (mv < 0)  # None (Reversi) / empty board (TTT): the reference draws with `random`
            if bool(need.any()):
                idx = need.nonzero().flatten().cpu().numpy()
                oc, pc = own[need].cpu().numpy(), opp[need].cpu().numpy()
                draws = []
                for a, b in zip(oc.tolist(), pc.tolist()):
                    if ttt:
                        lg = ~(a | b) & 0x1FF
                    else:
                        lgc = C.c_uint64()
                        _lib.check(L.bz_reversi_legal(a & (2**64 - 1), b & (2**64 - 1), size, C.byref(lgc)))
                        lg = lgc.value
                    cells = [i for i in range(64) if lg >> i & 1]
                    draws.append(cells[int(rng.integers(len(cells)))])
                mv[torch.as_tensor(idx, device=dev)] = torch.as_tensor(np.array(draws, dtype=np.int8)).to(dev)
            action = torch.where(mm_turn, mv.to(torch.uint8), action)
        log.append((action.cpu().numpy(), torch.where(active, to_move, torch.zeros_like(to_move)).cpu().numpy()))
        safe = torch.where(active, action, torch.zeros_like(action))  # finished games step a dummy action, ignored below
        with torch.cuda.device(dev):
            if ttt:
                _lib.check(L.bz_ttt_step_batch(own.data_ptr(), opp.data_ptr(), safe.data_ptr(), to_move.data_ptr(), B,
                                               own_n.data_ptr(), opp_n.data_ptr(), legal_n.data_ptr(), status.data_ptr(),
                                               win_n.data_ptr(), st()))
            else:
                _lib.check(L.bz_reversi_step_batch_sized(own.data_ptr(), opp.data_ptr(), safe.data_ptr(), B, size,
                                                         own_n.data_ptr(), opp_n.data_ptr(), legal_n.data_ptr(),
                                                         status.data_ptr(), win_n.data_ptr(), st()))
        bad = active & (status == _lib.ST_ILLEGAL)
        if bool(bad.any()):
            g = int(bad.nonzero()[0])
            who = "MCTS" if bool((to_move == mcts_colour)[g]) and ply >= opening_plies else ("opening" if ply < opening_plies else "minimax")
            raise RuntimeError(f"arena: a player produced an illegal move: game {g}, ply {ply}, {who} to move, own "
                               f"{int(own[g]) & (2**64 - 1):#018x} opp {int(opp[g]) & (2**64 - 1):#018x} action {int(action[g])}")
        term = active & (status == _lib.ST_TERMINAL)
        # TTT reports the absolute winner; Reversi the result for the player who just moved
        winner = torch.where(term, win_n if ttt else (win_n * to_move).to(torch.int8), winner)
        plies = plies + active.to(torch.int32)
        must_pass = active & (status == _lib.ST_MUST_PASS)  # next mover cannot move: the same side moves again
        run = active & (status == _lib.ST_RUNNING)
        new_own = torch.where(run, own_n, torch.where(must_pass, opp_n, own))
        new_opp = torch.where(run, opp_n, torch.where(must_pass, own_n, opp))
        to_move = torch.where(run, -to_move, to_move)
        own, opp = new_own, new_opp
        active = active & ~term
    if bool(active.any()):
        raise RuntimeError("arena: games still running after max_plies")
    return ArenaResult(winner.cpu().numpy(), mcts_colour.cpu().numpy(), plies.cpu().numpy(), log)
