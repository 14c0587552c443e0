"""The reference's minimax players (arena yard-stick) through the C ABI's scalar entry
points -- the host build of the exact code the arena kernels run -- against fixture F9, which was generated
by IMPORTING the reference's OptimalPlayer classes (oracle/gen_golden.py gen_f9)."""
import ctypes as C
import os
import random

import numpy as np

import betazero_amd as bz
from betazero_amd import _lib

G = os.path.join(os.path.dirname(__file__), "golden")


def _rev(x, o, size, depth):
    mv, sc = C.c_int32(), C.c_int32()
    _lib.check(_lib.lib().bz_reversi_minimax(x, o, size, depth, C.byref(mv), C.byref(sc)))
    return mv.value, sc.value


def test_reversi_minimax_matches_reference_optimal_player():
    d = np.load(os.path.join(G, "minimax_players.npz"))
    rows = d["reversi"]
    assert len(rows) >= 120 and set(rows[:, 0].tolist()) == {4, 6, 8}
    n_none = n_inf = 0
    for size, depth, sym1, x, o, move, score in rows.tolist():
        sym = sym1 - 1
        own, opp = (x, o) if sym == 1 else (o, x)
        mv, sc = _rev(own, opp, size, depth)
        exp_mv = -1 if move == 255 else move
        assert (mv, sc) == (exp_mv, score - 2000), (size, depth, sym, hex(x), hex(o))
        n_none += exp_mv == -1
        n_inf += abs(score - 2000) == 1000
    print("fixture rows", len(rows), "best_move None:", n_none, "+-inf scores:", n_inf)


def test_reversi_optimal_player_class_replays_the_reference_game():
    """OptimalPlayer(depth 2) as X vs OptimalPlayer(depth 3) as O on 6x6 through the reference's turn-loop
    semantics: the same move sequence and the same final board as the reference produced (random.seed(7))."""
    d = np.load(os.path.join(G, "minimax_players.npz"))
    random.seed(7)
    pl = {1: bz.ReversiOptimalPlayer(1, max_depth=2), -1: bz.ReversiOptimalPlayer(-1, max_depth=3)}
    b, cur, seq, over = bz.ReversiBoard(size=6), 1, [], False
    while not over:
        if b.generate_possible_moves(cur):
            r, c = pl[cur].get_move(b)
            seq.append((cur + 1, 8 * r + c))
            b = b.make_move(r, c, cur)
        over = b.is_game_over()
        cur = -cur
    assert np.array_equal(np.array(seq, dtype=np.int64), d["game6"])
    assert b.bits(1) == tuple(int(v) for v in d["game6_final"])
    p = bz.ReversiOptimalPlayer(1, 4)  # the API of the reference class: minimax(board, True, 0) -> (score, move)
    score, mv = p.minimax(bz.ReversiBoard(size=4), True, 0)
    assert mv in bz.ReversiBoard(size=4).generate_possible_moves(1) and isinstance(score, int)


def test_ttt_minimax_matches_reference_optimal_player_on_every_position():
    d = np.load(os.path.join(G, "minimax_players.npz"))
    rows = d["ttt"]
    assert len(rows) == 4519  # every reachable, unfinished, non-empty position
    mv, sc = C.c_int32(), C.c_int32()
    L = _lib.lib()
    for x, o, cur1, move, score1 in rows.tolist():
        cur = cur1 - 1
        own, opp = (x, o) if cur == 1 else (o, x)
        _lib.check(L.bz_ttt_minimax(own, opp, cur, C.byref(mv), C.byref(sc)))
        assert (mv.value, sc.value) == (move, score1 - 1), (x, o, cur)
    _lib.check(L.bz_ttt_minimax(0, 0, 1, C.byref(mv), C.byref(sc)))
    assert mv.value == -2  # empty board: the caller draws the opening move (players.py:35-36)
    _lib.check(L.bz_ttt_minimax(0b000000111, 0b000011000, 1, C.byref(mv), C.byref(sc)))
    assert (mv.value, sc.value) == (-1, 1)  # finished board: no move, the player has won
    # unreachable double-line position: X lines are tested first (tic_tac_toe_board.py:31-40), for either player
    _lib.check(L.bz_ttt_minimax(0b000000111, 0b111000000, 1, C.byref(mv), C.byref(sc)))
    assert sc.value == 1
    _lib.check(L.bz_ttt_minimax(0b111000000, 0b000000111, -1, C.byref(mv), C.byref(sc)))
    assert sc.value == -1


class _GameApiMinimax:
    """full-depth minimax over the Game API only (generate_possible_moves / make_move / is_game_over), memoised: an
    independent statement of the reference's decision rule (src/tic_tac_toe/players.py:30-70: the first move with the
    best score in generate_possible_moves() order) to hold the library's bz_ttt_minimax against"""

    def __init__(self, symbol):
        self.symbol, self._memo = symbol, {}

    def get_move(self, board):
        return self._minimax(board, True)[1]

    def _minimax(self, board, is_max):
        key = (board.board.tobytes(), is_max)
        if key in self._memo:
            return self._memo[key]
        over, winner = board.is_game_over()
        if over:
            res = ((1 if winner == self.symbol else -1 if winner == -self.symbol else 0), None)
        else:
            best, best_move = (-2, None) if is_max else (2, None)
            for mv in board.generate_possible_moves():
                sc, _ = self._minimax(board.make_move(*mv, self.symbol if is_max else -self.symbol), not is_max)
                if (is_max and sc > best) or (not is_max and sc < best):
                    best, best_move = sc, mv
            res = (best, best_move)
        self._memo[key] = res
        return res


def test_ttt_optimal_player_class_never_loses_to_itself_and_matches_memoised_minimax():
    random.seed(3)
    for _ in range(5):
        positions, winner = bz.TicTacToeHeadless(bz.OptimalPlayer(1), bz.OptimalPlayer(-1)).play()
        assert winner == 0 and len(positions) == 10  # two perfect players draw in 9 plies (as in the reference's CSV)
    # same decision as a minimax written against the Game API alone, on random reachable positions
    rng = np.random.default_rng(0)
    for _ in range(40):
        t, cur = bz.TicTacToeBoard(), 1
        for _ in range(int(rng.integers(1, 6))):
            mv = t.generate_possible_moves()
            if t.is_game_over()[0] or not mv:
                break
            t = t.make_move(*mv[int(rng.integers(len(mv)))], cur)
            cur = -cur
        if t.is_game_over()[0]:
            continue
        assert bz.OptimalPlayer(cur).get_move(t) == _GameApiMinimax(cur).get_move(t)


def _py_minimax(board, symbol, max_depth, is_max=True, depth=0):
    """the decision rule of reversi_players.py:41-69 written over the Game API (no pass rule inside the search, -inf / +inf
    for a side without a move, first strictly better move wins) -- an independent yard-stick for random positions"""
    if depth >= max_depth or board.is_game_over():
        w, (n1, n2) = board.get_score()
        return (n1 - n2 if symbol == 1 else n2 - n1), None
    best, best_move = (float("-inf"), None) if is_max else (float("inf"), None)
    for mv in board.generate_possible_moves(symbol if is_max else -symbol):
        sc, _ = _py_minimax(board.make_move(*mv, symbol if is_max else -symbol), symbol, max_depth, not is_max, depth + 1)
        if (is_max and sc > best) or (not is_max and sc < best):
            best, best_move = sc, mv
    return best, best_move


def test_reversi_minimax_random_positions_vs_game_api_restatement():
    """260 random positions (sizes 4 / 6 / 8, depths 0-3, both colours, incl. positions where the player has no move):
    bz_reversi_minimax == the rule restated over the API-compatible board (itself pinned to the reference by F1-F6)"""
    rng = random.Random(11)
    n_none = 0
    for trial in range(260):
        size = rng.choice((4, 4, 6, 6, 8))
        b, cur = bz.ReversiBoard(size=size), 1
        for _ in range(rng.randrange(0, size * size - 4)):
            mv = b.generate_possible_moves(cur)
            if not mv:
                cur = -cur
                mv = b.generate_possible_moves(cur)
                if not mv:
                    break
            b = b.make_move(*rng.choice(mv), cur)
            cur = -cur
        sym, depth = rng.choice((1, -1)), rng.choice((0, 1, 2, 3)) if size > 4 else rng.choice((0, 1, 2, 3, 4))
        sc, mv = _py_minimax(b, sym, depth)
        own, opp = b.bits(sym)
        gm, gs = _rev(own, opp, size, depth)
        exp_sc = 1000 if sc == float("inf") else (-1000 if sc == float("-inf") else sc)
        assert (gm, gs) == (-1 if mv is None else 8 * mv[0] + mv[1], exp_sc), (trial, size, depth, sym)
        n_none += mv is None
    assert n_none > 10


def test_reversi_optimal_player_takes_any_depth_like_the_reference_and_the_batched_kernel_says_its_limit():
    """the reference's OptimalPlayer recurses to whatever max_depth it is given (reversi_players.py:36-38).  The scalar
    entry point keeps 60 stack frames -- no line of play is longer -- so the player accepts every depth: depth 12 on a
    4x4 board (a game there has at most 12 plies) is the full-depth search and equals the Game-API minimax move for move;
    depth 1000 searches the same tree.  Only the BATCHED kernel (the arena's opponent) holds 8 levels and says so."""
    import pytest
    with pytest.raises(ValueError, match="max_depth"):
        bz.ReversiOptimalPlayer(1, max_depth=-1)
    b = bz.ReversiBoard(size=4)
    player, moves = 1, 0
    while not b.is_game_over() and moves < 4:
        if b.generate_possible_moves(player):
            sc12, mv12 = bz.ReversiOptimalPlayer(player, max_depth=12).minimax(b)
            sc_big, mv_big = bz.ReversiOptimalPlayer(player, max_depth=1000).minimax(b)
            ref_sc, ref_mv = _py_minimax(b, player, 12)
            assert (sc12, mv12) == (ref_sc, ref_mv) == (sc_big, mv_big), (moves, sc12, mv12, ref_sc, ref_mv)
            # (None = the reference's quirk: every line ends in a forced pass, which its search scores -inf -> random.choice)
            b = b.make_move(*(mv12 or b.generate_possible_moves(player)[0]), player)
            moves += 1
        player = -player
    assert bz.ReversiOptimalPlayer(1, max_depth=9).max_depth == 9
    from betazero_amd.arena import play_arena
    with pytest.raises(ValueError, match="opponent_depth"):
        play_arena("reversi", 4, 8, opponent_depth=9)
