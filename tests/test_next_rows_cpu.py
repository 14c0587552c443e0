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


def test_e4m3_quantiser_matches_the_oracle_and_is_idempotent():
    """betazero_amd.quant.e4m3_round (numpy) == the oracle's C rounding on a dense sweep; per-channel
    scales are powers of two that put max|w| in (224, 448]; fake-quantisation is idempotent."""
    import torch
    from betazero_amd.net import PolicyValueNet
    from betazero_amd.quant import channel_scale, e4m3_round, fake_quantize_fp8_
    from oracle import oracle as orc
    r0, r1 = np.random.default_rng(0), np.random.default_rng(1)
    xs = np.concatenate([np.linspace(0, 500, 5001), 2.0 ** np.arange(-14, 9).astype(np.float64),
                         r0.random(5000) * 2.0 ** r1.integers(-12, 9, 5000).astype(np.float64)]).astype(np.float32)
    xs = np.concatenate([xs, -xs])
    a = e4m3_round(xs)
    b = np.array([orc.lib().orc_e4m3_round(float(x)) for x in xs], dtype=np.float32)
    assert np.array_equal(a, b) and len(np.unique(np.abs(a))) == 127 and np.abs(a).max() == 448
    w = r0.normal(0, 0.03, (16, 9 * 128)).astype(np.float32)
    s = channel_scale(w)
    m = np.abs(w).max(1) * s
    assert np.all(np.log2(s) == np.round(np.log2(s))) and np.all(m > 224) and np.all(m <= 448)
    torch.manual_seed(1)
    net = fake_quantize_fp8_(PolicyValueNet(32, 1, 64))
    p1 = net.flat_params().copy()
    p2 = fake_quantize_fp8_(net).flat_params()
    assert np.array_equal(p1, p2)
