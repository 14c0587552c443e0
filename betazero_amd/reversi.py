"""API-compatible ReversiBoard (reference: src/reversi/game_logic/reversi_board.py:4-88)
backed by bitboards + libbz_hip.so's scalar rule entry points, and a headless
mirror of the reference's turn loop (reversi_terminal.py:16-38).

Provenance note: `__str__` and the two strings printed by `get_score(print_result=True)` reproduce the
reference's text (reversi_board.py:16-20, 78-83) verbatim, because that text is observable output of the
API; everything else in this file is this build's own."""
import ctypes as C

import numpy as np

from . import _lib

_W = {s: np.array([[1 << (8 * r + c) for c in range(s)] for r in range(s)], dtype=np.uint64) for s in range(1, 9)}


def _to_bits(mask, w):
    return int(w[mask].sum()) if mask.any() else 0


class ReversiBoard:
    """Same constructor, attributes, methods, return shapes, text and exception
    as the reference class.  `.board` is an int64 (size,size) ndarray with cells
    in {-1,0,+1}; it may be edited in place (edits are picked up on the next call).

    Sizes: every size whose cells fit the 64-bit boards, 1..8 (the reference's constructor is generic,
    reversi_board.py:4-14; its own drivers use 4, 6 and 8); a larger size is refused with a ValueError.
    Cells and players outside {-1, 0, +1}: as in the reference, any non-zero cell is occupied
    (reversi_board.py:26), `player` plays against `-player` whatever the number is (:34-37), and make_move
    stores `player` (:48,:58): such cells are walls for the +-1 game -- rays stop there, nobody moves there."""

    def __init__(self, board=None, size=8):
        if board is None:
            if not (isinstance(size, (int, np.integer)) and 1 <= size <= 8):
                raise ValueError("size must be 1..8 (bit = 8*row+col boards; the reference's sizes beyond 8 do not fit)")
            self.size = int(size)
            p = size // 2 - 1  # reversi_board.py:9-11 (numpy indexing: -1 is the last cell, which matters at size 1)
            a, b = p % size, (p + 1) % size
            self._o = (1 << (8 * a + b)) | (1 << (8 * b + a))
            self._x = ((1 << (8 * a + a)) | (1 << (8 * b + b))) & ~self._o  # the -1 cells are assigned last
            self._blk = 0
            self._arr = None
        else:  # copy-constructor from another board object (reads .board and .size)
            self.size = int(board.size)
            if not 1 <= self.size <= 8:
                raise ValueError("size must be 1..8 (bit = 8*row+col boards; the reference's sizes beyond 8 do not fit)")
            self._arr = np.copy(board.board)
            self._x = self._o = self._blk = 0
            self._sync()

    # ---- bitboard <-> ndarray
    def _sync(self):
        if self._arr is not None:
            w = _W[self.size]
            self._x = _to_bits(self._arr == 1, w)
            self._o = _to_bits(self._arr == -1, w)
            self._blk = _to_bits(self._arr != 0, w) & ~(self._x | self._o)  # occupied, but nobody's stone

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
        b.size, b._x, b._o, b._blk, b._arr = size, int(x), int(o), 0, None
        return b

    def bits(self, player=1):
        """(own, opp) bitboards seen by `player` (bit = 8*row+col)."""
        self._sync()
        return (self._x, self._o) if player == 1 else (self._o, self._x)

    def _sides(self, player):
        """(own, opp, walls) for ANY `player` value: own = cells equal to player, opp = cells equal to -player,
        walls = every other non-zero cell (reversi_board.py:26, :34-37)"""
        self._sync()
        if player == 1:
            return self._x, self._o, self._blk
        if player == -1:
            return self._o, self._x, self._blk
        if player == 0 or self._arr is None:  # nobody's stones: no cell equals +-player (0 never closes a ray, :36)
            return 0, 0, self._x | self._o | self._blk
        w = _W[self.size]
        own, opp = _to_bits(self._arr == player, w), _to_bits(self._arr == -player, w)
        return own, opp, (self._x | self._o | self._blk) & ~(own | opp)

    # ---- reference API
    def __str__(self):  # reversi_board.py:16-20
        out_str = "  " + " ".join(map(str, range(self.size))) + "\n"
        for i, row in enumerate(self.board):
            out_str += str(i) + ' ' + ' '.join(['X' if cell == 1 else 'O' if cell == -1 else '.' for cell in row]) + "\n"
        return out_str

    def __repr__(self):
        return f"{self.board}"

    def _legal(self, player):
        own, opp, walls = self._sides(player)
        if own == 0 or opp == 0:
            return 0
        out = C.c_uint64()
        _lib.check(_lib.lib().bz_reversi_legal(own, opp, self.size, C.byref(out)))
        return out.value & ~walls

    def is_valid_move(self, row, col, player):
        if not (0 <= row < self.size and 0 <= col < self.size):
            return False
        return bool(self._legal(player) >> (8 * row + col) & 1)

    def make_move(self, row, col, player):
        if row is None or col is None:
            raise ValueError("Invalid move")
        if not self.is_valid_move(row, col, player):
            raise ValueError("Invalid move")
        own, opp, walls = self._sides(player)
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib().bz_reversi_apply(own, opp, self.size, int(row), int(col), C.byref(a), C.byref(b), None))
        if player in (1, -1) and not walls:
            return ReversiBoard.from_bits(a.value, b.value, self.size) if player == 1 else \
                ReversiBoard.from_bits(b.value, a.value, self.size)
        nb = ReversiBoard(self)  # off-domain cells / players: the array is the state (reversi_board.py:47-58)
        nb._arr[(np.uint64(a.value) & _W[self.size]) != 0] = player
        nb._sync()
        return nb

    def is_game_over(self):
        self._sync()
        if self._blk:
            return self._legal(1) == 0 and self._legal(-1) == 0
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
