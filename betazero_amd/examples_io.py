"""On-disk example format compatible with the reference's CSV
(src/tic_tac_toe/SL/generate_training_games.py:40-54 writes it, SL/train.py:16,38-40
reads it): columns State,Action with space-joined integers.  save_examples_csv adds
Pi and Z columns (ignored by the reference's reader, which only touches State/Action)."""
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
    s = ex.states().reshape(len(ex), -1)
    na = ex.pi.shape[1]
    with open(filename, "w", newline="") as f:
        f.write("State,Action,Pi,Z\n")
        for i in range(len(ex)):
            onehot = np.zeros(s.shape[1], dtype=np.int64)
            if ex.act[i] < s.shape[1]:
                onehot[ex.act[i]] = 1
            f.write(" ".join(str(int(v)) for v in s[i]) + "," + " ".join(str(int(v)) for v in onehot) + "," +
                    " ".join(repr(float(v)) for v in ex.pi[i][:na]) + "," + str(int(ex.z[i])) + "\n")


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
