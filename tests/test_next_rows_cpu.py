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


def test_dedupe_is_exact_under_key_collisions():
    """SL/train.py:45-50 keeps the first of every EXACTLY equal pair.  The device dedupe groups rows by a 64-bit
    content key and then compares every row with the head of its group on the full content, so two different rows that
    share a key are both kept -- also when a duplicate of the first sits behind the second (the case a neighbour
    compare gets wrong)."""
    import torch
    from betazero_amd.augment import _first_occurrences
    own = torch.tensor([5, 9, 5, 7, 9, 5], dtype=torch.int64)
    opp = torch.tensor([1, 2, 1, 3, 2, 1], dtype=torch.int64)
    pi = torch.tensor([[.5, .5], [.1, .9], [.5, .5], [1., 0.], [.1, .9], [.25, .75]], dtype=torch.float32)
    # rows 0, 2 equal; rows 1, 4 equal; row 5 has row 0's stones but another pi.  Keys: everything collides on 42
    # except row 3 -- the worst case for a hash-only dedupe
    key = torch.tensor([42, 42, 42, 7, 42, 42], dtype=torch.int64)
    keep = _first_occurrences(key, own, opp, pi).tolist()
    assert keep == [0, 1, 3, 5]   # exact: no distinct row dropped, no duplicate kept (row 4 repeats row 1 behind a colliding key)
    # with honest keys (equal content <=> equal key) it is exactly "first occurrence, insertion order"
    key2 = torch.tensor([10, 11, 10, 12, 11, 13], dtype=torch.int64)
    assert _first_occurrences(key2, own, opp, pi).tolist() == [0, 1, 3, 5]


def test_device_examples_unpack_equals_host_unpack_on_cpu_tensors():
    """unpack_example_block_device (views from the engine geometry, game ids from the header as tensor scalars, row
    selection by nonzero) against unpack_example_block (header parsed on the host) on a block of the engine's exact
    layout -- the function is device-agnostic, so the CPU suite can pin it"""
    import torch
    from betazero_amd.engine import (DeviceExamples, build_example_block, concat_device_examples, unpack_example_block,
                                     unpack_example_block_device, _EX_FIELDS, _GAMES, _SIZES)
    g = torch.Generator().manual_seed(3)
    R, B, T, na = 2, 7, 64, 65
    ln = torch.randint(1, T, (R, B), generator=g, dtype=torch.int32)
    ln[1, ::2] = -1  # unfinished games
    arr = {"own": torch.randint(-2**62, 2**62, (R, B, T), generator=g, dtype=torch.int64),
           "opp": torch.randint(0, 2**62, (R, B, T), generator=g, dtype=torch.int64), "pi": torch.rand((R, B, T, na), generator=g),
           "z": torch.randint(-1, 2, (R, B, T), generator=g).to(torch.int8), "mover": torch.ones((R, B, T), dtype=torch.int8),
           "act": torch.randint(0, 64, (R, B, T), generator=g).to(torch.uint8), "len": ln,
           "winner": torch.zeros((R, B), dtype=torch.int8)}
    block = build_example_block(arr, 1000, 50, "reversi")
    offs, off = [], 0
    for name, dt, esz in _EX_FIELDS:
        offs.append(off)
        off += (arr[name].numel() * esz + 255) & ~255
    geom = {"B": B, "rounds": R, "t_max": T, "na": na, "game": _GAMES["reversi"], "size": 8, "offs": offs,
            "ex_bytes": int(block.numel())}
    dev = unpack_example_block_device(block, geom)
    host = unpack_example_block(block)
    got = dev.cpu()
    for f in ("own", "opp", "z", "mover", "act", "game", "ply"):
        assert np.array_equal(getattr(got, f), getattr(host, f)), f
    assert np.array_equal(got.pi.view(np.uint32), host.pi.view(np.uint32)) and got.size == host.size == 8
    assert len(dev) == int(ln.clamp(min=0).sum())
    rt = DeviceExamples.from_host(host, "cpu")   # host -> device form -> host is the identity; concat keeps the order
    both = concat_device_examples([rt, dev]).cpu()
    assert np.array_equal(both.own, np.concatenate([host.own, host.own])) and np.array_equal(both.states()[:len(host)], host.states())
    assert np.array_equal(dev.states().numpy(), host.states())


def test_refresh_device_net_refuses_non_finite_weights():
    """the losses of a training step are computed before its optimiser update, so a last step that overflows is invisible
    in them; refresh_device_net must not push such weights into the engine (round 3's channels-last runs did: every later
    search returned garbage) -- it raises, names the parameter and leaves the engine's net alone"""
    import pytest
    import torch
    from betazero_amd.net import PolicyValueNet
    from betazero_amd.train import refresh_device_net

    class Net:
        updated = False

        def update(self, params):
            self.updated = True
    for fused in (False, True):
        m = PolicyValueNet(32, 1, 8, fused_tower=fused)
        dn = Net()
        refresh_device_net(dn, m)
        assert dn.updated
        with torch.no_grad():
            m.polfc.bias[3] = float("inf")
        dn = Net()
        with pytest.raises(FloatingPointError, match="polfc.bias"):
            refresh_device_net(dn, m)
        assert not dn.updated


def test_examples_npz_roundtrip_bit_for_bit_and_csv_floats_read_back_exactly(tmp_path):
    """the .npz format for headline-sized example sets (SURVEY 8(f) row 2's side-car): every array bit for bit; and the CSV
    writer's Pi text (shortest representation of each float32) reads back to the same float32"""
    from betazero_amd.engine import Examples
    from betazero_amd.examples_io import load_csv, load_examples_npz, save_examples_csv, save_examples_npz
    rng = np.random.default_rng(3)
    n = 200
    own = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64) | (np.uint64(1) << np.uint64(63))
    opp = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64) & ~own
    pi = (rng.random((n, 65)) ** 8).astype(np.float32)
    pi[0] = 0.0; pi[0, 64] = 1.0; pi[1, 3] = np.float32(1e-30); pi[2, 5] = np.float32(1 / 3)
    ex = Examples(own, opp, pi, rng.integers(-1, 2, n).astype(np.int8), rng.choice([-1, 1], n).astype(np.int8),
                  rng.integers(0, 65, n).astype(np.uint8), rng.integers(0, 10**12, n), rng.integers(0, 60, n).astype(np.int32), 8)
    p = tmp_path / "ex.npz"
    save_examples_npz(ex, str(p))
    back = load_examples_npz(str(p))
    assert back.size == 8 and len(back) == n
    for k in ("own", "opp", "z", "mover", "act", "game", "ply"):
        assert np.array_equal(getattr(back, k), getattr(ex, k)) and getattr(back, k).dtype == getattr(ex, k).dtype, k
    assert np.array_equal(back.pi.view(np.uint32), ex.pi.view(np.uint32))
    c = tmp_path / "ex.csv"
    save_examples_csv(ex, str(c))
    rows = load_csv(str(c))
    assert np.array_equal(rows["Pi"].view(np.uint32), ex.pi.view(np.uint32)) and np.array_equal(rows["Z"], ex.z)
    assert np.array_equal(rows["State"], ex.states().reshape(n, 64))
    passes = ex.act == 64
    assert (rows["Action"][passes].sum(1) == 0).all() and np.array_equal(rows["Action"][~passes].argmax(1), ex.act[~passes])


def test_holdout_split_is_the_reference_split():
    """SL/train.py:66-76: val_size = int(total_size * validation_split), a random partition of the rows"""
    import torch
    from betazero_amd.engine import DeviceExamples, Examples
    from betazero_amd.train import holdout_split, select_rows
    g = torch.Generator().manual_seed(0)
    for n, frac in ((180, 0.2), (7, 0.2), (1000, 0.25), (5, 0.0)):
        tr, va = holdout_split(n, frac, g, "cpu")
        assert len(va) == int(n * frac) and len(tr) == n - len(va)
        assert sorted(torch.cat([tr, va]).tolist()) == list(range(n))
    tr2, va2 = holdout_split(180, 0.2, torch.Generator().manual_seed(0), "cpu")
    tr3, va3 = holdout_split(180, 0.2, torch.Generator().manual_seed(0), "cpu")
    assert torch.equal(va2, va3) and not torch.equal(va2, torch.arange(36))       # seeded, and a real shuffle
    n = 30
    ex = DeviceExamples.from_host(Examples(np.arange(n, dtype=np.uint64), np.zeros(n, np.uint64), np.zeros((n, 9), np.float32),
                                           np.zeros(n, np.int8), np.ones(n, np.int8), np.zeros(n, np.uint8), np.arange(n),
                                           np.zeros(n, np.int32), 3), "cpu")
    sub = select_rows(ex, va2[va2 < n])
    assert sub.own.tolist() == va2[va2 < n].tolist() and sub.size == 3
