"""API-compatible TicTacToeBoard (reference: src/tic_tac_toe/tic_tac_toe_board.py:4-43),
TicTacToeHeadless (tic_tac_toe.py:6-34) and process_game_positions
(SL/generate_training_games.py:12-23).

Provenance note: this file is the API contract itself.  `__str__` follows the reference's text line for line on
purpose (tic_tac_toe_board.py:7-15: the printed board is observable output).  `TicTacToeHeadless.play` keeps the
reference's trajectory contract (tic_tac_toe.py:13-34: board.board before every move plus the final one, winner
from is_game_over, "Invalid move: ..." on an illegal move) in this build's own loop.  Everything else (bitboard
state, rule calls through the C ABI) is this build's own."""
import ctypes as C

import numpy as np

from . import _lib

_W3 = np.array([[1 << (3 * r + c) for c in range(3)] for r in range(3)], dtype=np.int64)


class TicTacToeBoard:
    def __init__(self, board=None):
        self.board = np.zeros((3, 3), dtype=int) if board is None else np.copy(board)

    def bits(self, player=1):
        x = int((_W3 * (self.board == 1)).sum())
        o = int((_W3 * (self.board == -1)).sum())
        return (x, o) if player == 1 else (o, x)

    def _other(self):
        """cells that are occupied by neither X nor O: make_move stores whatever `player` is (tic_tac_toe_board.py:28)
        and every non-zero cell is occupied afterwards (:21, :39, :43)"""
        return int((_W3 * ((self.board != 0) & (self.board != 1) & (self.board != -1))).sum())

    def __str__(self):  # tic_tac_toe_board.py:7-15
        out_str = ""
        for i, row in enumerate(self.board):
            if i != 0:
                out_str += '\n'
            out_str += ' ' + ' | '.join(['X' if cell == 1 else 'O' if cell == -1 else ' ' for cell in row]) + ' '
            if i != 2:
                out_str += '\n---+---+---'
        return out_str

    def __repr__(self):
        return f"{self.board}"

    def is_valid_move(self, row, col):
        if not (0 <= row < 3 and 0 <= col < 3):
            return False
        x, o = self.bits()
        out = C.c_uint32()
        _lib.check(_lib.lib().bz_ttt_legal(x, o, C.byref(out)))
        return bool((out.value & ~self._other()) >> (3 * row + col) & 1)

    def make_move(self, row, col, player):
        x, o = self.bits()
        out = C.c_uint32()
        if not self.is_valid_move(row, col):
            raise ValueError("Invalid move")
        _lib.check(_lib.lib().bz_ttt_apply(x, o, int(row), int(col), C.byref(out)))  # ValueError("Invalid move")
        new_board = TicTacToeBoard(self.board)
        new_board.board[row][col] = player  # the reference stores whatever `player` is
        return new_board

    def is_game_over(self):
        x, o = self.bits()
        over, w = C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().bz_ttt_game_over(x, o, C.byref(over), C.byref(w)))
        if not over.value and (x | o | self._other()) == 0x1FF:
            return True, 0  # full, counting the cells that hold something else (tic_tac_toe_board.py:38-39)
        return (True, w.value) if over.value else (False, None)

    def generate_possible_moves(self):
        x, o = self.bits()
        out = C.c_uint32()
        _lib.check(_lib.lib().bz_ttt_legal(x, o, C.byref(out)))
        m = out.value & ~self._other()
        return [(i, j) for i in range(3) for j in range(3) if m >> (3 * i + j) & 1]


class TicTacToeHeadless:
    """tic_tac_toe.py:6-34: records board.board before every move plus the final
    one and returns (game_positions, winner)."""

    def __init__(self, player1, player2):
        self.board = TicTacToeBoard()
        self.players = {1: player1, -1: player2}
        self.current_player = 1
        self.game_positions = []

    def play(self):
        """the trajectory contract of tic_tac_toe.py:13-34: `game_positions` receives board.board before every move and
        once more after the last one (len = plies + 1); an illegal move surfaces as ValueError("Invalid move: ...");
        returns (game_positions, winner) with winner as is_game_over reports it"""
        while True:
            self.game_positions.append(self.board.board)
            finished, winner = self.board.is_game_over()
            if finished:
                return self.game_positions, winner
            mover = self.current_player
            row, col = self.players[mover].get_move(self.board)
            try:
                self.board = self.board.make_move(row, col, mover)
            except ValueError as err:
                raise ValueError(f"Invalid move: {err}")
            self.current_player = -mover


def process_game_positions(positions):
    """generate_training_games.py:12-23 on numpy: side-to-move canonical states
    (position i times (-1)**i) and one-hot actions -p[i+1] - p[i]."""
    p = np.stack(positions).astype(np.int64)
    sign = (-1) ** np.arange(p.shape[0])
    p = p * sign.reshape((-1,) + (1,) * (p.ndim - 1))
    actions = -p[1:] - p[:-1]
    return p[:-1], actions
