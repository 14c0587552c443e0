"""betazero_amd -- MI355X-native self-play data generation behind BetaZero's
Python Game / Player API (reference: whaiproject/BetaZero).

Single-board rule calls go through the scalar entry points of libbz_hip.so;
everything batched (env step, MCTS, net, self-play) runs as gfx950 HIP kernels
and raises when no HIP device is present -- there is no CPU fallback."""
from .reversi import ReversiBoard, ReversiHeadless  # noqa: F401
from .tic_tac_toe import TicTacToeBoard, TicTacToeHeadless, process_game_positions  # noqa: F401
from .players import (Player, ReversiPlayer, RandomPlayer, ReversiRandomPlayer, MCTSPlayer,  # noqa: F401
                      OptimalPlayer, ReversiOptimalPlayer, NetPlayer)

__all__ = ["ReversiBoard", "ReversiHeadless", "TicTacToeBoard", "TicTacToeHeadless", "process_game_positions",
           "Player", "ReversiPlayer", "RandomPlayer", "ReversiRandomPlayer", "MCTSPlayer",
           "OptimalPlayer", "ReversiOptimalPlayer", "NetPlayer"]
