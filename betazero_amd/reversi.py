"""API-compatible ReversiBoard (reference: src/reversi/game_logic/reversi_board.py:4-88)
backed by bitboards + libbz_hip.so's scalar rule entry points, and a headless
mirror of the reference's turn loop (reversi_terminal.py:16-38).

Provenance note: `__str__` and the two strings printed by `get_score(print_result=True)` reproduce the
reference's text (reversi_board.py:16-20, 78-83) verbatim, because that text is observable output of the
API; everything else in this file is this build's own."""
import ctypes as C

import numpy as np

from . import _lib

_W = {s: np.array([[1 << (8 * r + c) for c in range(s)] for r in range(s)], dtype=np.uint64) for s in (4, 6, 8)}


class ReversiBoard:
    """Same constructor, attributes, methods, return shapes, text and exception
    as the reference class.  `.board` is an int64 (size,size) ndarray with cells
    in {-1,0,+1}; it may be edited in place (edits are picked up on the next call)."""

    def __init__(self, board=None, size=8):
        if board is None:
            if size not in _W:
                raise ValueError("size must be 4, 6 or 8")
            self.size = size
            p = size // 2 - 1  # reversi_board.py:9-11
            self._x = (1 << (8 * p + p)) | (1 << (8 * (p + 1) + p + 1))
            self._o = (1 << (8 * p + p + 1)) | (1 << (8 * (p + 1) + p))
            self._arr = None
        else:  # copy-constructor from another board object (reads .board and .size)
            self.size = int(board.size)
            self._arr = np.copy(board.board)
            self._x = self._o = 0
            self._sync()

    # ---- bitboard <-> ndarray
    def _sync(self):
        if self._arr is not None:
            w = _W[self.size]
            self._x = int((w * (self._arr == 1)).sum())
            self._o = int((w * (self._arr == -1)).sum())

    @property
    def board(self):
        if self._arr is None:
            w = _W[self.size]
            x = (np.uint64(self._x) & w) != 0
            o = (np.uint64(self._o) & w) != 0
            self._arr = x.astype(int) - o.astype(int)
        return self._arr

    @board.setter
    def board(self, value):
        self._arr = np.array(value, dtype=int)
        self._sync()

    @classmethod
    def from_bits(cls, x, o, size=8):
        b = cls.__new__(cls)
        b.size, b._x, b._o, b._arr = size, int(x), int(o), None
        return b

    def bits(self, player=1):
        """(own, opp) bitboards seen by `player` (bit = 8*row+col)."""
        self._sync()
        return (self._x, self._o) if player == 1 else (self._o, self._x)

    # ---- reference API
    def __str__(self):  # reversi_board.py:16-20
        out_str = "  " + " ".join(map(str, range(self.size))) + "\n"
        for i, row in enumerate(self.board):
            out_str += str(i) + ' ' + ' '.join(['X' if cell == 1 else 'O' if cell == -1 else '.' for cell in row]) + "\n"
        return out_str

    def __repr__(self):
        return f"{self.board}"

    def _legal(self, player):
        own, opp = self.bits(player)
        out = C.c_uint64()
        _lib.check(_lib.lib().bz_reversi_legal(own, opp, self.size, C.byref(out)))
        return out.value

    def is_valid_move(self, row, col, player):
        if not (0 <= row < self.size and 0 <= col < self.size):
            return False
        return bool(self._legal(player) >> (8 * row + col) & 1)

    def make_move(self, row, col, player):
        own, opp = self.bits(player)
        a, b = C.c_uint64(), C.c_uint64()
        if row is None or col is None:
            raise ValueError("Invalid move")
        _lib.check(_lib.lib().bz_reversi_apply(own, opp, self.size, int(row), int(col), C.byref(a), C.byref(b), None))
        return ReversiBoard.from_bits(a.value, b.value, self.size) if player == 1 else \
            ReversiBoard.from_bits(b.value, a.value, self.size)

    def is_game_over(self):
        self._sync()
        out = C.c_int32()
        _lib.check(_lib.lib().bz_reversi_game_over(self._x, self._o, self.size, C.byref(out)))
        return bool(out.value)

    def get_score(self, print_result=False):
        self._sync()
        w, n1, n2 = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().bz_reversi_score(self._x, self._o, C.byref(w), C.byref(n1), C.byref(n2)))
        winner, count_player1, count_player2 = w.value, n1.value, n2.value
        if print_result:  # reversi_board.py:78-83
            if winner == 0:
                print(f"It's a tie! Player X: {count_player1}, Player O: {count_player2}")
            else:
                winner_symbol = 'X' if winner == 1 else 'O'
                print(f"Player {winner_symbol} wins! Score - Player X: {count_player1}, Player O: {count_player2}")
        return winner, (count_player1, count_player2)

    def generate_possible_moves(self, player):
        m = self._legal(player)
        return [(i, j) for i in range(self.size) for j in range(self.size) if m >> (8 * i + j) & 1]


class ReversiHeadless:
    """The reference's game loop (ReversiTerminal.play, reversi_terminal.py:16-38)
    without the prints: pass rule, terminal check after every turn, records the
    position before every move like TicTacToeHeadless does."""

    def __init__(self, player1, player2, size=8):
        self.board = ReversiBoard(size=size)
        self.players = {1: player1, -1: player2}
        self.current_player = 1
        self.game_positions = []
        self.movers = []

    def play(self):
        game_over = False
        while not game_over:
            moves = self.board.generate_possible_moves(self.current_player)
            if moves:
                row, col = self.players[self.current_player].get_move(self.board)
                try:
                    new_board = self.board.make_move(row, col, self.current_player)
                except ValueError as e:
                    raise ValueError(f"Invalid move: {e}")
                self.game_positions.append(self.board.board)
                self.movers.append(self.current_player)
                self.board = new_board
            game_over = self.board.is_game_over()
            self.current_player *= -1
        self.game_positions.append(self.board.board)
        winner, _ = self.board.get_score()
        return self.game_positions, winner
