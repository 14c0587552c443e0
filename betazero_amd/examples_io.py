"""On-disk example format compatible with the reference's CSV
(src/tic_tac_toe/SL/generate_training_games.py:40-54 writes it, SL/train.py:16,38-40
reads it): columns State,Action with space-joined integers.  save_examples_csv adds
Pi and Z columns (ignored by the reference's reader, which only touches State/Action).

The CSV is the reference's format and fine at its size (180 rows); a self-play iteration at the headline configuration is
~237,000 rows x 65 floats -- ~150 MB of text.  save_examples_npz / load_examples_npz are the format for that (SURVEY 8(f) row
2 "or an .npz side-car"): the Examples arrays as they are (bitboards, pi, z, mover, act, game, ply), compressed, loaded
back bit for bit; `states()` of the loaded object gives the reference's State column whenever it is wanted."""
import csv

import numpy as np


def save_to_csv(states, actions, filename="tic_tac_toe_data.csv"):
    """Same file the reference writes: header State,Action; one row per position;
    states/actions are integer arrays [n, ...] flattened row-major."""
    states = np.asarray(states).reshape(len(states), -1)
    actions = np.asarray(actions).reshape(len(actions), -1)
    with open(filename, "w", newline="") as f:
        f.write("State,Action\n")
        for s, a in zip(states, actions):
            f.write(" ".join(str(int(v)) for v in s) + "," + " ".join(str(int(v)) for v in a) + "\n")


def save_examples_csv(ex, filename):
    """Examples -> CSV readable by the reference's TicTacToeDataset (State, Action = one-hot of
    the move played) plus Pi (visit-count policy) and Z (outcome for the mover)."""
    s = ex.states().reshape(len(ex), -1).astype(np.int64)
    onehot = np.zeros_like(s)
    on_board = np.asarray(ex.act) < s.shape[1]              # (Reversi's pass, action 64, has no cell)
    onehot[np.nonzero(on_board)[0], np.asarray(ex.act)[on_board]] = 1
    pi = np.asarray(ex.pi, dtype=np.float32)
    # one str() per ELEMENT, not per row of Python-level joins: repr of a float32 (shortest text that reads back to the same
    # float32) through numpy's own formatter, then one join per row
    pis = np.array([np.format_float_positional(v, unique=True, trim="0") if np.isfinite(v) else repr(float(v)) for v in pi.reshape(-1)],
                   dtype=object).reshape(pi.shape)
    with open(filename, "w", newline="") as f:
        f.write("State,Action,Pi,Z\n")
        f.writelines(" ".join(map(str, s[i])) + "," + " ".join(map(str, onehot[i])) + "," + " ".join(pis[i]) + "," +
                     str(int(ex.z[i])) + "\n" for i in range(len(ex)))


_NPZ_FIELDS = ("own", "opp", "pi", "z", "mover", "act", "game", "ply")


def save_examples_npz(ex, filename):
    """Examples (host) or DeviceExamples -> one compressed .npz: the arrays as they are, plus the board size"""
    if hasattr(ex, "cpu") and not isinstance(ex.own, np.ndarray):
        ex = ex.cpu()
    np.savez_compressed(filename, size=np.int64(ex.size), **{k: np.asarray(getattr(ex, k)) for k in _NPZ_FIELDS})


def load_examples_npz(filename):
    """-> Examples, bit for bit what save_examples_npz was given (no pickle involved)"""
    from .engine import Examples
    d = np.load(filename, allow_pickle=False)
    return Examples(**{k: d[k] for k in _NPZ_FIELDS}, size=int(d["size"]))


def load_csv(filename):
    """-> dict of arrays: State [n, k] int64, Action [n, k] int64 and, when present, Pi, Z"""
    cols = {}
    with open(filename, newline="") as f:
        rd = csv.reader(f)
        header = next(rd)
        rows = list(rd)
    for j, name in enumerate(header):
        if name in ("State", "Action"):
            cols[name] = np.array([[int(v) for v in r[j].split()] for r in rows], dtype=np.int64)
        elif name == "Pi":
            cols[name] = np.array([[float(v) for v in r[j].split()] for r in rows], dtype=np.float32)
        elif name == "Z":
            cols[name] = np.array([int(r[j]) for r in rows], dtype=np.int8)
    return cols


def collect_game_data(num_games, player1, player2):
    """The reference's data generator (SL/generate_training_games.py:25-38): num_games head-less tic-tac-toe games between
    two players, every position canonicalised for its mover, the move as a one-hot "action".  Returns (states, actions)
    as int64 arrays [n, 3, 3]; save_to_csv() then writes the reference's file.  With the reference's default players --
    OptimalPlayer(1) vs OptimalPlayer(-1), here betazero_amd.OptimalPlayer on the library's minimax -- and the same
    `random` seed this reproduces the reference's games move for move (the only randomness is the opening move)."""
    from .tic_tac_toe import TicTacToeHeadless, process_game_positions
    all_states, all_actions = [], []
    for _ in range(num_games):
        raw_positions, _ = TicTacToeHeadless(player1, player2).play()
        states, actions = process_game_positions(raw_positions)
        all_states.extend(np.asarray(states))
        all_actions.extend(np.asarray(actions))
    return np.array(all_states, dtype=np.int64), np.array(all_actions, dtype=np.int64)
