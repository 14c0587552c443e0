"""Drop-in demonstration (build container only: needs /root/reference, skipped elsewhere).
The REFERENCE's own game loop and players run unchanged over OUR board classes, and produce
exactly what they produce over the reference's boards under the same seed."""
import importlib
import os
import random
import sys
import types

import numpy as np
import pytest

import betazero_amd as bz

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present on this machine")


def _fresh_import(name, path, monkeypatch, shims=None):
    sys.dont_write_bytecode = True
    for k in list(sys.modules):
        if k in (name, "reversi_board", "players", "players.reversi_players", "tic_tac_toe_board"):
            monkeypatch.delitem(sys.modules, k, raising=False)
    for k, v in (shims or {}).items():
        monkeypatch.setitem(sys.modules, k, v)
    monkeypatch.syspath_prepend(path)
    return importlib.import_module(name)


def _shim(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


@pytest.mark.parametrize("size", [4, 6, 8])
def test_reference_reversi_terminal_loop_over_our_board(size, monkeypatch, capsys):
    path = os.path.join(REF, "src/reversi/game_logic")
    rt = _fresh_import("reversi_terminal", path, monkeypatch)          # the reference, untouched
    random.seed(100 + size)
    rt.ReversiTerminal(rt.RandomPlayer(1), rt.RandomPlayer(-1), size=size).play()
    expected = capsys.readouterr().out
    rt2 = _fresh_import("reversi_terminal", path, monkeypatch,         # same loop + players, OUR ReversiBoard
                        {"reversi_board": _shim("reversi_board", ReversiBoard=bz.ReversiBoard)})
    assert rt2.ReversiBoard is bz.ReversiBoard
    random.seed(100 + size)
    rt2.ReversiTerminal(rt2.RandomPlayer(1), rt2.RandomPlayer(-1), size=size).play()
    got = capsys.readouterr().out
    assert got == expected and ("wins!" in got or "tie" in got)       # every printed board, pass line and score


def test_reference_ttt_minimax_player_over_our_board(monkeypatch):
    """the reference's OptimalPlayer (players.py:30-70) searches through OUR TicTacToeBoard: same moves"""
    tpath = os.path.join(REF, "src/tic_tac_toe")
    ref_board = _fresh_import("tic_tac_toe_board", tpath, monkeypatch).TicTacToeBoard
    players = importlib.import_module("players")
    for seed in range(3):
        seqs = []
        for cls in (ref_board, bz.TicTacToeBoard):
            random.seed(seed)
            b, cur, moves = cls(), 1, []
            p = {1: players.OptimalPlayer(1), -1: players.OptimalPlayer(-1)}
            while not b.is_game_over()[0]:
                mv = p[cur].get_move(b)
                moves.append(tuple(int(v) for v in mv))
                b = b.make_move(*mv, cur)
                cur = -cur
            seqs.append((moves, b.is_game_over(), b.board.tolist()))
        assert seqs[0] == seqs[1] and seqs[0][1] == (True, 0)
    monkeypatch.delitem(sys.modules, "players", raising=False)


def test_reference_reversi_minimax_player_over_our_board(monkeypatch):
    """the reference's depth-limited OptimalPlayer (reversi_players.py:35-77) over OUR ReversiBoard"""
    monkeypatch.delitem(sys.modules, "players", raising=False)
    rpath = os.path.join(REF, "src/reversi")
    monkeypatch.syspath_prepend(rpath)
    rp = importlib.import_module("players.reversi_players")
    rb = _fresh_import("reversi_board", os.path.join(rpath, "game_logic"), monkeypatch).ReversiBoard
    out = []
    for cls in (rb, bz.ReversiBoard):
        random.seed(7)
        b, cur, moves = cls(size=6), 1, []
        pl = {1: rp.OptimalPlayer(1, max_depth=2), -1: rp.RandomPlayer(-1)}
        for _ in range(12):
            if b.generate_possible_moves(cur):
                mv = pl[cur].get_move(b)
                moves.append(tuple(int(v) for v in mv))
                b = b.make_move(*mv, cur)
            cur = -cur
        out.append((moves, np.asarray(b.board).tolist()))
    assert out[0] == out[1]
