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


def test_collect_game_data_reproduces_the_reference_generator(monkeypatch, tmp_path):
    """SL/generate_training_games.py end to end: the reference's collect_game_data loop with the REFERENCE's OptimalPlayer
    over the reference's board (its script cannot be imported -- tkinter -- so its 12-line loop body is driven here with
    its own classes) against betazero_amd.examples_io.collect_game_data with OUR OptimalPlayer, same `random` seed:
    identical states and actions, and an identical CSV, in the format of the reference's own tic_tac_toe_data.csv."""
    import importlib.util
    from betazero_amd.examples_io import collect_game_data, save_to_csv
    path = os.path.join(REF, "src/tic_tac_toe")
    monkeypatch.syspath_prepend(path)
    for k in ("players", "tic_tac_toe_board"):
        monkeypatch.delitem(sys.modules, k, raising=False)
    spec = importlib.util.spec_from_file_location("ref_ttt_players", os.path.join(path, "players.py"))
    rp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rp)
    ref_board = importlib.import_module("tic_tac_toe_board").TicTacToeBoard

    def ref_games(n):  # tic_tac_toe.py:13-34 + generate_training_games.py:12-23, over the reference's classes
        p1, p2 = rp.OptimalPlayer(1), rp.OptimalPlayer(-1)
        states, actions = [], []
        for _ in range(n):
            b, cur, positions = ref_board(), 1, []
            while True:
                positions.append(b.board)
                r, c = (p1 if cur == 1 else p2).get_move(b)
                b = b.make_move(r, c, cur)
                over, _w = b.is_game_over()
                if over:
                    positions.append(b.board)
                    break
                cur = -cur
            pos = np.stack(positions).astype(np.int64)
            pos = pos * ((-1) ** np.arange(len(pos)))[:, None, None]
            states.extend(pos[:-1]); actions.extend(-pos[1:] - pos[:-1])
        return np.array(states), np.array(actions)

    random.seed(2024)
    rs, ra = ref_games(12)
    random.seed(2024)
    s, a = collect_game_data(12, bz.OptimalPlayer(1), bz.OptimalPlayer(-1))
    assert np.array_equal(s, rs) and np.array_equal(a, ra) and len(s) == 12 * 9
    f1, f2 = tmp_path / "ours.csv", tmp_path / "ref.csv"
    save_to_csv(s, a, str(f1)); save_to_csv(rs, ra, str(f2))
    assert f1.read_bytes() == f2.read_bytes() and f1.read_text().startswith("State,Action\n")
