"""Player plug-in point (reference: src/tic_tac_toe/players.py:6-9,
src/reversi/players/reversi_players.py:5-8) and the MCTS player that fills it."""
import random
from abc import ABC, abstractmethod

import numpy as np


class Player(ABC):
    @abstractmethod
    def get_move(self, board):
        pass


class ReversiPlayer(ABC):
    @abstractmethod
    def get_move(self, board):
        pass


class RandomPlayer(Player):  # players.py:25-27
    def get_move(self, board):
        return random.choice(board.generate_possible_moves())


class ReversiRandomPlayer(ReversiPlayer):  # reversi_players.py:26-32
    def __init__(self, symbol):
        self.symbol = symbol

    def get_move(self, board):
        moves = board.generate_possible_moves(self.symbol)
        return random.choice(moves) if moves else (None, None)


class OptimalPlayer(Player):
    """The reference's tic-tac-toe OptimalPlayer (src/tic_tac_toe/players.py:30-70) on the library's
    minimax (bz_ttt_minimax: the host build of the code the arena kernel runs): random opening move on an
    empty board, otherwise the first move with the best full-depth minimax score."""

    def __init__(self, symbol):
        self.symbol = symbol

    def minimax(self, board, is_maximizing=True):
        """(score, move) for `symbol` to move -- the reference's minimax(board, True)"""
        import ctypes as C
        from . import _lib
        assert is_maximizing, "only the root call minimax(board, True) is mirrored"
        own, opp = board.bits(self.symbol)
        mv, sc = C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().bz_ttt_minimax(own, opp, self.symbol, C.byref(mv), C.byref(sc)))
        return sc.value, (None if mv.value < 0 else (mv.value // 3, mv.value % 3))

    def get_move(self, board):
        moves = board.generate_possible_moves()
        if len(moves) == 9:
            return random.choice(moves)
        return self.minimax(board, True)[1]


class ReversiOptimalPlayer(ReversiPlayer):
    """The reference's Reversi OptimalPlayer (src/reversi/players/reversi_players.py:35-77): depth-limited
    minimax on the stone difference, no pass rule inside the search, random.choice when no move scores above
    -inf -- on bz_reversi_minimax.  Any max_depth >= 0, like the reference's recursion (the scalar entry point keeps a
    stack of 60 frames -- no line of play is longer; only the BATCHED kernel of the arena is limited to depth 8)."""

    def __init__(self, symbol, max_depth=4):
        if int(max_depth) < 0:
            raise ValueError(f"ReversiOptimalPlayer: max_depth must be >= 0 (got {max_depth})")
        self.symbol, self.max_depth = symbol, int(max_depth)

    def minimax(self, board, is_maximizing=True, depth=0):
        import ctypes as C
        from . import _lib
        assert is_maximizing and depth == 0, "only the root call minimax(board, True, 0) is mirrored"
        own, opp = board.bits(self.symbol)
        mv, sc = C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().bz_reversi_minimax(own, opp, board.size, self.max_depth, C.byref(mv), C.byref(sc)))
        score = float("inf") if sc.value >= 1000 else (float("-inf") if sc.value <= -1000 else sc.value)
        return score, (None if mv.value < 0 else (mv.value // 8, mv.value % 8))

    def get_move(self, board):
        _, best_move = self.minimax(board, True, 0)
        if best_move is None:
            return random.choice(board.generate_possible_moves(self.symbol))
        return best_move


class NetPlayer(ReversiPlayer):
    """The net-only player: AIPlayer.get_move (src/tic_tac_toe/players.py:84-98) for Reversi (8x8, or the reference's
    6x6 / 4x4 boards in the top-left corner of the 8x8 planes) and the conv
    policy/value net -- canonicalise the position for the side to move (:85), one forward (:86), play the best LEGAL
    move in descending-logit order (:92-98; ties -> lowest action, where torch.sort leaves them unspecified)."""

    def __init__(self, symbol, net, precision="bf16"):
        self.symbol, self.net, self.precision = symbol, net, precision
        self.last_logits = None

    def get_move(self, board):
        import torch
        own, opp = board.bits(self.symbol)
        o = torch.as_tensor(np.array([own], dtype=np.uint64).view(np.int64)).to(self.net.device)
        p = torch.as_tensor(np.array([opp], dtype=np.uint64).view(np.int64)).to(self.net.device)
        lg, _ = self.net.forward(o, p, bf16=self.precision != "f32", fp8=self.precision == "fp8")
        lg = lg[0, :64].cpu().numpy()
        self.last_logits = lg
        moves = board.generate_possible_moves(self.symbol)
        if not moves:
            return (None, None)
        idx = np.array([8 * r + c for r, c in moves])
        best = idx[np.argmax(lg[idx])]  # generate_possible_moves is row-major: the first maximum is the lowest action
        return int(best) // 8, int(best) % 8


class MCTSPlayer(Player):
    """get_move(board) -> (row, col) by one GPU search (PUCT, `sims` simulations)
    from `board` with `symbol` to move; plays argmax visit count (ties -> lowest
    action).  Works for TicTacToeBoard and 8x8 ReversiBoard; evaluator "net_bf16"
    / "net_f32" need a betazero_amd.net.DeviceNet (Reversi, any of the reference's board sizes)."""

    def __init__(self, symbol, sims=800, net=None, evaluator=None, c_puct=1.5, device="cuda:0"):
        from .engine import check_sims
        check_sims(sims)  # a ValueError naming the limit here, not a RuntimeError at the first get_move
        self.symbol, self.sims, self.net, self.c_puct, self.device = symbol, sims, net, c_puct, device
        # evaluator: "uniform" | "hash" | "net_bf16" | "net_f32" | "net_fp8", or a callable (own, opp, kind) -> (logits, value)
        # on CUDA tensors (SelfPlayEngine.search_external): any torch module, e.g. an MLP for tic-tac-toe
        self.eval_fn = evaluator if callable(evaluator) else None
        if self.eval_fn is not None:
            evaluator = "external"
        self.evaluator = evaluator or ("net_bf16" if net is not None else "uniform")
        self._eng = {}
        self.last_visits = None

    def _engine(self, game):
        from .engine import SelfPlayEngine
        if game not in self._eng:
            self._eng[game] = SelfPlayEngine(game, 1, self.sims, self.evaluator, self.net, self.c_puct,
                                             device=self.device)
        return self._eng[game]

    def get_move(self, board):
        if hasattr(board, "size") and board.size not in (8, 6, 4):
            raise ValueError(f"the search engine plays Reversi on 8x8, 6x6 and 4x4 boards, not {board.size}x{board.size} "
                             "(the rule calls of ReversiBoard cover every size)")
        game = {8: "reversi", 6: "reversi6", 4: "reversi4"}[board.size] if hasattr(board, "size") else "ttt"
        if game == "ttt" and self.evaluator.startswith("net"):
            raise ValueError("the conv net evaluators serve the Reversi boards; use evaluator='uniform' or 'hash' for tic-tac-toe")
        own, opp = board.bits(self.symbol)
        eng = self._engine(game)
        eng.set_roots([own], [opp], [self.symbol])
        if self.eval_fn is not None:
            eng.search_external(self.eval_fn)
        else:
            eng.search()
        N, _, _ = eng.root_stats()
        eng.status()  # raises on engine error flags (e.g. terminal root)
        self.last_visits = N[0]
        a = int(np.argmax(N[0]))  # first maximum = lowest action
        if N[0][a] == 0 or a == 64:
            return (None, None)  # mover has no move (reversi_players.py:32)
        n = 3 if game == "ttt" else 8
        return a // n, a % n
