"""SURVEY 8(f) rows, CPU side: the CSV example format and the D4 index maps."""
import hashlib
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), "golden")


def test_csv_writer_reproduces_the_reference_csv_byte_for_byte(tmp_path):
    from betazero_amd.examples_io import load_csv, save_to_csv
    d = np.load(os.path.join(G, "ttt_csv.npz"))
    meta = json.load(open(os.path.join(G, "csv_meta.json")))
    p = tmp_path / "tic_tac_toe_data.csv"
    save_to_csv(d["states"], d["actions"], str(p))
    raw = open(p, "rb").read()
    assert len(raw) == meta["bytes"] and hashlib.sha256(raw).hexdigest() == meta["sha256"]
    back = load_csv(str(p))
    assert np.array_equal(back["State"], d["states"]) and np.array_equal(back["Action"], d["actions"])


def test_examples_csv_roundtrip(tmp_path):
    from betazero_amd.engine import Examples
    from betazero_amd.examples_io import load_csv, save_examples_csv
    rng = np.random.default_rng(0)
    n = 12
    own = rng.integers(0, 512, n).astype(np.uint64)
    opp = (rng.integers(0, 512, n).astype(np.uint64)) & ~own
    pi = rng.random((n, 9)).astype(np.float32)
    ex = Examples(own, opp, pi, rng.integers(-1, 2, n).astype(np.int8), np.ones(n, np.int8),
                  rng.integers(0, 9, n).astype(np.uint8), np.arange(n), np.zeros(n, np.int32), 3)
    p = tmp_path / "ex.csv"
    save_examples_csv(ex, str(p))
    back = load_csv(str(p))
    assert np.array_equal(back["State"], ex.states().reshape(n, 9))
    assert np.array_equal(back["Pi"], pi) and np.array_equal(back["Z"], ex.z)
    assert np.array_equal(back["Action"].argmax(1), ex.act)
