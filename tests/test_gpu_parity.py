"""GPU parity tests proper: HIP path (through the C ABI) vs the CPU oracle on the
same seeded inputs, vs the committed golden fixtures, and size-independent
properties at BASELINE.json's full sizes.  Integer work bit-exact; tree floats
(W, P, pi) bit-exact; fp32 net bit-exact; bf16 net within a stated tolerance."""
import json
import os

import numpy as np
import pytest
import torch

from betazero_amd import _lib
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _dev_u64(a):
    return torch.as_tensor(np.asarray(a, dtype=np.uint64).view(np.int64)).to(DEV)


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------- env step
def _reversi_step(own, opp, act):
    n = len(own)
    o, p, a = _dev_u64(own), _dev_u64(opp), torch.as_tensor(np.asarray(act, dtype=np.uint8)).to(DEV)
    on, pn, lg = (torch.empty(n, dtype=torch.int64, device=DEV) for _ in range(3))
    st = torch.empty(n, dtype=torch.uint8, device=DEV)
    w = torch.empty(n, dtype=torch.int8, device=DEV)
    _lib.check(_lib.lib().bz_reversi_step_batch(o.data_ptr(), p.data_ptr(), a.data_ptr(), n, on.data_ptr(),
                                                pn.data_ptr(), lg.data_ptr(), st.data_ptr(), w.data_ptr(), _stream()))
    torch.cuda.synchronize()
    return (on.cpu().numpy().view(np.uint64), pn.cpu().numpy().view(np.uint64), lg.cpu().numpy().view(np.uint64),
            st.cpu().numpy(), w.cpu().numpy())


def test_reversi_step_batch_on_golden_games():
    d = np.load(os.path.join(G, "reversi_random_games.npz"))
    rows = d["rows"][d["rows"][:, 1] == 8]
    cur = rows[:, 3].astype(np.int64) - 1
    x, o = rows[:, 4], rows[:, 5]
    own = np.where(cur == 1, x, o)
    opp = np.where(cur == 1, o, x)
    act = np.where(rows[:, 7] == 255, 64, rows[:, 7]).astype(np.uint8)
    on, pn, lg, st, w = _reversi_step(own, opp, act)
    # golden: legal mask of the NEXT row of the same game is the legal mask of the next mover
    for i in range(len(rows)):
        exp = orc.reversi_apply(int(own[i]), int(opp[i]), 8, int(act[i]) >> 3, int(act[i]) & 7) if act[i] != 64 \
            else (int(own[i]), int(opp[i]), 0)
        assert (int(pn[i]), int(on[i])) == (exp[0], exp[1])
        assert int(lg[i]) == orc.reversi_legal(exp[1], exp[0], 8)
        over = bool(rows[i, 9])
        assert (st[i] == _lib.ST_TERMINAL) == over
        if over:
            ws, _ = orc.reversi_score(exp[0], exp[1])
            assert w[i] == ws
        elif int(lg[i]) == 0:
            assert st[i] == _lib.ST_MUST_PASS
        else:
            assert st[i] == _lib.ST_RUNNING
    # next-row consistency with the fixture itself
    same = (rows[1:, 0] == rows[:-1, 0])
    nxt_legal = rows[1:, 6]
    assert np.array_equal(lg[:-1][same], nxt_legal[same])


def test_reversi_step_batch_arbitrary_positions_and_illegal_actions():
    d = np.load(os.path.join(G, "reversi_positions.npz"))
    pos = d["pos"][d["pos"][:, 0] == 8]
    own, opp, legal = pos[:, 1], pos[:, 2], pos[:, 3]
    rng = np.random.default_rng(0)
    act = rng.integers(0, 65, size=len(pos)).astype(np.uint8)
    on, pn, lg, st, w = _reversi_step(own, opp, act)
    for i in range(len(pos)):
        a = int(act[i])
        ok = (int(legal[i]) == 0) if a == 64 else bool(int(legal[i]) >> a & 1)
        if not ok:
            assert st[i] == _lib.ST_ILLEGAL and (int(on[i]), int(pn[i])) == (int(own[i]), int(opp[i]))
        else:
            exp = (int(own[i]), int(opp[i]), 0) if a == 64 else orc.reversi_apply(int(own[i]), int(opp[i]), 8, a >> 3, a & 7)
            assert (int(pn[i]), int(on[i])) == (exp[0], exp[1])
    # legal_batch == fixture masks
    o, p = _dev_u64(own), _dev_u64(opp)
    out = torch.empty(len(pos), dtype=torch.int64, device=DEV)
    _lib.check(_lib.lib().bz_reversi_legal_batch(o.data_ptr(), p.data_ptr(), len(pos), out.data_ptr(), _stream()))
    assert np.array_equal(out.cpu().numpy().view(np.uint64), legal)


def test_reversi_step_batch_100k_random_positions_every_output_vs_oracle():
    """the batched env step on 100,003 random positions of every density (sparse openings to nearly full boards, where
    passes, must-pass and terminal positions are common) with random actions -- legal moves, passes, occupied cells, moves
    that flip nothing -- against the oracle's rules, all five outputs: stones after the move, next legal mask, status
    (RUNNING / TERMINAL / ILLEGAL / MUST_PASS), winner.  (Round 3 rewrote this kernel: flips and east / west moves by carry
    propagation, the rare second legal mask once per lane: the fixtures alone hold 1200 positions.)"""
    rng = np.random.default_rng(2026)
    n = 100003  # ragged: not a multiple of 4 games per lane
    a = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, n).astype(np.uint64)
    b = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, n).astype(np.uint64)
    keep = np.full(n, ~np.uint64(0))
    for _ in range(3):  # thin out a share of the boards: densities from ~6 % to 50 % per colour
        m = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, n).astype(np.uint64)
        keep = np.where(rng.random(n) < 0.4, keep & m, keep)
    fill = rng.random(n) < 0.25  # and fill others up: nearly full boards
    own, opp = a & ~b & keep, b & ~a & keep
    extra = np.where(fill, ~(own | opp) & (rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1)), np.uint64(0))
    own = own | (extra & a)
    opp = opp | (extra & ~a)
    legal = np.array([orc.reversi_legal(int(o), int(p)) for o, p in zip(own, opp)], dtype=np.uint64)
    act = rng.integers(0, 65, n).astype(np.uint8)
    pick_legal = (rng.random(n) < 0.6) & (legal != 0)   # most actions legal, the rest anything (incl. pass)
    low = np.array([(int(l) & -int(l)).bit_length() - 1 if l else 64 for l in legal], dtype=np.uint8)
    act = np.where(pick_legal, low, act).astype(np.uint8)
    act = np.where((legal == 0) & (rng.random(n) < 0.7), 64, act).astype(np.uint8)
    on, pn, lg, st, w = _reversi_step(own, opp, act)
    seen = {0: 0, 1: 0, 2: 0, 3: 0}
    for i in range(n):
        o, p, ac, l = int(own[i]), int(opp[i]), int(act[i]), int(legal[i])
        ok = (l == 0) if ac == 64 else bool(l >> ac & 1)
        if not ok:
            assert st[i] == _lib.ST_ILLEGAL and (int(on[i]), int(pn[i]), int(lg[i]), int(w[i])) == (o, p, l, 0), i
            seen[2] += 1
            continue
        no, np_ = (p, o) if ac == 64 else orc.reversi_apply(o, p, 8, ac >> 3, ac & 7)[:2][::-1]
        # reversi_apply returns (mover's stones, opponent's stones) after the move; the kernel returns the NEXT mover's view
        assert (int(on[i]), int(pn[i])) == (no, np_), i
        nl = orc.reversi_legal(no, np_)
        assert int(lg[i]) == nl, i
        if nl == 0 and orc.reversi_legal(np_, no) == 0:
            d = bin(np_).count("1") - bin(no).count("1")
            assert st[i] == _lib.ST_TERMINAL and int(w[i]) == (d > 0) - (d < 0), i
            seen[1] += 1
        elif nl == 0:
            assert st[i] == _lib.ST_MUST_PASS and w[i] == 0, i
            seen[3] += 1
        else:
            assert st[i] == _lib.ST_RUNNING and w[i] == 0, i
            seen[0] += 1
    assert min(seen.values()) > 200, seen  # every status really occurs, many times


def test_ttt_step_batch_exhaustive():
    d = np.load(os.path.join(G, "ttt_exhaustive.npz"))
    pos = np.concatenate([d["pos"], d["extra"]])
    rows = []
    for x, o, cur, legal, over, w1 in pos.tolist():
        tm = 1 if cur == 1 else -1
        own, opp = (x, o) if tm == 1 else (o, x)
        for a in range(9):
            rows.append((own, opp, a, tm))
    rows = np.array(rows, dtype=np.int64)
    n = len(rows)
    t16 = lambda a: torch.as_tensor(a.astype(np.int16)).to(DEV)  # noqa: E731
    own, opp = t16(rows[:, 0]), t16(rows[:, 1])
    act = torch.as_tensor(rows[:, 2].astype(np.uint8)).to(DEV)
    tm = torch.as_tensor(rows[:, 3].astype(np.int8)).to(DEV)
    on, pn, lg = (torch.empty(n, dtype=torch.int16, device=DEV) for _ in range(3))
    st = torch.empty(n, dtype=torch.uint8, device=DEV)
    w = torch.empty(n, dtype=torch.int8, device=DEV)
    _lib.check(_lib.lib().bz_ttt_step_batch(own.data_ptr(), opp.data_ptr(), act.data_ptr(), tm.data_ptr(), n,
                                            on.data_ptr(), pn.data_ptr(), lg.data_ptr(), st.data_ptr(), w.data_ptr(),
                                            _stream()))
    on, pn, lg, st, w = (t.cpu().numpy() for t in (on, pn, lg, st, w))
    for i, (o_, p_, a, t) in enumerate(rows.tolist()):
        if (o_ | p_) >> a & 1:
            assert st[i] == _lib.ST_ILLEGAL
            continue
        me = o_ | 1 << a
        x, o = (me, p_) if t == 1 else (p_, me)
        over, win = orc.ttt_game_over(x, o)
        assert (int(pn[i]), int(on[i])) == (me, p_) and int(lg[i]) == orc.ttt_legal(x, o)
        assert (st[i] == _lib.ST_TERMINAL) == over and (not over or w[i] == win)


# ---------------------------------------------------------------- MCTS search
def _engine(game, n, sims, ev, **kw):
    from betazero_amd.engine import SelfPlayEngine
    return SelfPlayEngine(game, n, sims, ev, **kw)


def _cases(kind):
    return [c for c in json.load(open(os.path.join(G, "mcts_twin.json")))["cases"] if c["kind"] == kind]


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "stepwise"])
def test_search_matches_golden_twin_and_oracle(fused):
    d = np.load(os.path.join(G, "mcts_twin.npz"))
    for case in _cases("search"):
        ev = case["eval"]
        eng = _engine(case["game"], 3, case["sims"], ev if fused else "external")
        eng.set_roots([case["own"]] * 3, [case["opp"]] * 3, [case["to_move"]] * 3)
        if fused:
            eng.search()
        else:  # drive the step API with caller-filled logits / value (BZ_EVAL_EXTERNAL)
            lb = eng.leaf_buffers()
            na = eng.na

            def fill():
                torch.cuda.synchronize()
                own = lb["own"].cpu().numpy().view(np.uint64)
                opp = lb["opp"].cpu().numpy().view(np.uint64)
                lg = np.zeros((3, na), np.float32)
                v = np.zeros(3, np.float32)
                if ev == "hash":
                    for g in range(3):
                        lg[g], v[g] = orc.eval_hash(int(own[g]), int(opp[g]), na)
                lb["logits"].copy_(torch.from_numpy(lg))
                lb["value"].copy_(torch.from_numpy(v))
            eng.root_begin(); fill(); eng.expand_backup()
            for s in range(case["sims"]):
                eng.select(s); fill(); eng.expand_backup()
        N, W, P = eng.root_stats()
        eng.status()
        i = case["id"]
        for g in range(3):
            assert np.array_equal(N[g], d[f"s{i}_N"]), case
            assert np.array_equal(W[g].view(np.uint32), d[f"s{i}_W"].view(np.uint32)), case
            assert np.array_equal(P[g].view(np.uint32), d[f"s{i}_P"].view(np.uint32)), case
        _, _, _, cnt = orc.mcts_search({"ttt": orc.GAME_TTT, "reversi": orc.GAME_REVERSI, "reversi6": orc.GAME_REVERSI6,
                                        "reversi4": orc.GAME_REVERSI4}[case["game"]], case["own"],
                                       case["opp"], case["to_move"], case["sims"],
                                       orc.EVAL_UNIFORM if ev == "uniform" else orc.EVAL_HASH)
        got = eng.counters()
        for k in ("n_sims", "n_path_nodes", "n_child_scored", "n_edges_backed", "n_expanded", "n_child_written",
                  "n_env_steps"):
            assert got[k] == 3 * cnt[k], (k, got, cnt)
        eng.reset_counters()


def test_search_random_reversi_positions_vs_oracle():
    d = np.load(os.path.join(G, "reversi_random_games.npz"))
    rows = d["rows"][(d["rows"][:, 1] == 8) & (d["rows"][:, 7] != 255)]
    rng = np.random.default_rng(5)
    sel = rows[rng.choice(len(rows), 96, replace=False)]
    cur = sel[:, 3].astype(np.int64) - 1
    own = np.where(cur == 1, sel[:, 4], sel[:, 5])
    opp = np.where(cur == 1, sel[:, 5], sel[:, 4])
    eng = _engine("reversi", 96, 200, "hash")
    eng.set_roots(own, opp, cur.astype(np.int8))
    eng.search()
    N, W, P = eng.root_stats()
    eng.status()
    for g in range(96):
        n, w, p, _ = orc.mcts_search(orc.GAME_REVERSI, int(own[g]), int(opp[g]), int(cur[g]), 200, orc.EVAL_HASH)
        assert np.array_equal(N[g], n)
        assert np.array_equal(W[g].view(np.uint32), w.view(np.uint32))
        assert np.array_equal(P[g].view(np.uint32), p.view(np.uint32))


# ---------------------------------------------------------------- self-play
def _check_selfplay(game, n, sims, ev, temp_moves, openings, seed, base=0, net=None, onet=None, **kw):
    eng = _engine(game, n, sims, ev, net=net, temp_moves=temp_moves, openings=openings, seed=seed, game_id_base=base, **kw)
    eng.run_iteration()
    ex = eng.examples()
    winners, lens = eng.winners()
    ogame = orc.GAME_TTT if game == "ttt" else orc.GAME_REVERSI
    oev = {"uniform": orc.EVAL_UNIFORM, "hash": orc.EVAL_HASH, "net_f32": orc.EVAL_NET_F32}[ev]
    for g in range(n):
        r = orc.selfplay_game(ogame, base + g, sims, oev, temp_moves, openings, seed, net=onet)
        m = ex.game == base + g
        assert lens[0, g] == len(r["own"]) and winners[0, g] == r["winner"]
        assert np.array_equal(ex.own[m], r["own"]) and np.array_equal(ex.opp[m], r["opp"])
        assert np.array_equal(ex.act[m], r["act"]) and np.array_equal(ex.mover[m], r["mover"])
        assert np.array_equal(ex.pi[m].view(np.uint32), r["pi"].view(np.uint32))
        assert np.array_equal(ex.z[m], r["z"])
    return ex


def test_selfplay_ttt_and_reversi_vs_oracle_bitexact():
    _check_selfplay("ttt", 64, 25, "uniform", 0, 0, 0)          # BASELINE cfg 1 setting
    _check_selfplay("ttt", 64, 40, "hash", 4, 0, 7, base=100)
    _check_selfplay("reversi", 48, 24, "hash", 8, 1, 0)          # cfg 3 diversification rules
    _check_selfplay("reversi", 24, 16, "uniform", 0, 0, 3, base=5)


def test_selfplay_matches_golden_twin_fixture():
    d = np.load(os.path.join(G, "mcts_twin.npz"))
    for case in _cases("selfplay"):
        eng = _engine(case["game"], 1, case["sims"], case["eval"], temp_moves=case["temp_moves"],
                      openings=case["openings"], seed=case["seed"], game_id_base=case["gid"])
        eng.run_iteration()
        ex = eng.examples()
        i = case["id"]
        assert np.array_equal(ex.own, d[f"g{i}_own"]) and np.array_equal(ex.act, d[f"g{i}_act"])
        assert np.array_equal(ex.pi.view(np.uint32), d[f"g{i}_pi"].view(np.uint32))
        assert eng.winners()[0][0, 0] == case["winner"]
        assert np.array_equal(ex.z, (case["winner"] * d[f"g{i}_mover"]).astype(np.int8))


def test_examples_are_canonical_like_the_reference_csv():
    """s_{k+1} == -(s_k + onehot(a_k)) inside a TTT game (the CSV invariant)."""
    eng = _engine("ttt", 32, 30, "hash", temp_moves=9, seed=11)
    eng.run_iteration()
    ex = eng.examples()
    s = ex.states().reshape(len(ex), 9).astype(np.int64)
    for g in np.unique(ex.game):
        idx = np.nonzero(ex.game == g)[0]
        for a, b in zip(idx[:-1], idx[1:]):
            onehot = np.zeros(9, np.int64); onehot[ex.act[a]] = 1
            assert np.array_equal(s[b], -(s[a] + onehot))
    assert np.allclose(ex.pi.sum(1), 1.0, atol=1e-6) and set(np.unique(ex.z)) <= {-1, 0, 1}


# ---------------------------------------------------------------- net
def _net(C, NB, seed=0, bf16=False):
    from betazero_amd.net import PolicyValueNet
    torch.manual_seed(seed)
    m = PolicyValueNet(C, NB, 64)
    if bf16:
        m.round_to_bf16_()
    return m


def _positions(n, seed=1):
    d = np.load(os.path.join(G, "reversi_random_games.npz"))
    rows = d["rows"][d["rows"][:, 1] == 8]
    idx = np.random.default_rng(seed).choice(len(rows), n, replace=n > len(rows))
    return rows[idx, 4].copy(), rows[idx, 5].copy()


@pytest.mark.parametrize("C,NB", [(32, 2), (64, 1), (128, 6), (256, 1)])
def test_net_f32_bitexact_vs_oracle(C, NB):
    from betazero_amd.net import DeviceNet
    m = _net(C, NB)
    n = 24 if C == 128 else 64
    own, opp = _positions(n)
    dn = DeviceNet.from_module(m, n)
    lg, v = dn.forward(_dev_u64(own), _dev_u64(opp), bf16=False)
    on = orc.Net(C, NB, 64, m.flat_params())
    olg, ov = on.forward(own, opp)
    assert np.array_equal(lg.cpu().numpy().view(np.uint32), olg.view(np.uint32))
    assert np.array_equal(v.cpu().numpy().view(np.uint32), ov.view(np.uint32))
    # and the oracle's net agrees with the plain torch fp32 module (tolerance: summation order)
    from betazero_amd.net import bits_to_planes
    with torch.no_grad():
        tl, tv = m(bits_to_planes(own, opp))
    assert np.allclose(olg, tl.numpy(), atol=2e-4) and np.allclose(ov, tv.numpy(), atol=2e-5)


def test_net_bf16_mfma_vs_oracle_bf16_emulation():
    """bf16 tower (MFMA, fp32 accumulate) vs the oracle rounding activations to bf16
    at the same layer boundaries.  Tolerance 2e-3 abs on logits (std 0.15) / 1e-3 on value: about 3x the
    measured error (5.8e-4 / 1.1e-4; accumulation order differs, one bf16 ulp is 2^-8 relative), plus a
    per-row bound relative to the row's own logit range."""
    from betazero_amd.net import DeviceNet
    m = _net(128, 6, bf16=True)
    n = 37  # ragged: not a multiple of the 4-position workgroup tile
    own, opp = _positions(n, seed=3)
    dn = DeviceNet.from_module(m, 64)
    lg, v = dn.forward(_dev_u64(own), _dev_u64(opp), bf16=True)
    on = orc.Net(128, 6, 64, m.flat_params())
    olg, ov = on.forward(own, opp, bf16=True)
    err_l = np.abs(lg.cpu().numpy() - olg).max()
    err_v = np.abs(v.cpu().numpy() - ov).max()
    print("bf16 net max |dlogit|", err_l, "max |dv|", err_v)
    assert err_l < 2e-3 and err_v < 1e-3
    rel = np.abs(lg.cpu().numpy() - olg).max(1) / (olg.max(1) - olg.min(1))
    assert rel.max() < 5e-3, rel.max()  # per row: error under 0.5 % of that row's logit range
    flg, fv = on.forward(own, opp, bf16=False)  # report the bf16-vs-fp32 gap too
    print("bf16-vs-fp32 gap: logits", np.abs(lg.cpu().numpy() - flg).max(), "value", np.abs(v.cpu().numpy() - fv).max())


def test_selfplay_with_f32_net_bitexact_vs_oracle():
    from betazero_amd.net import DeviceNet
    m = _net(32, 2, seed=4)
    dn = DeviceNet.from_module(m, 8)
    on = orc.Net(32, 2, 64, m.flat_params())
    _check_selfplay("reversi", 8, 12, "net_f32", 8, 1, 0, net=dn, onet=on)


def test_selfplay_with_bf16_net_is_legal_and_terminates():
    from betazero_amd.net import DeviceNet
    m = _net(128, 6, bf16=True)
    dn = DeviceNet.from_module(m, 16)
    eng = _engine("reversi", 16, 8, "net_bf16", net=dn, temp_moves=8, openings=1)
    eng.run_iteration()
    ex = eng.examples()
    winners, lens = eng.winners()
    assert (lens[0] > 40).all()
    for g in range(16):  # replay every game through the oracle's rules
        m_ = ex.game == g
        own, opp, act = ex.own[m_], ex.opp[m_], ex.act[m_]
        for k in range(len(own)):
            assert orc.reversi_legal(int(own[k]), int(opp[k])) >> int(act[k]) & 1
        r = orc.reversi_apply(int(own[-1]), int(opp[-1]), 8, int(act[-1]) >> 3, int(act[-1]) & 7)
        assert orc.reversi_game_over(r[0], r[1])
        x, o = (r[0], r[1]) if ex.mover[m_][-1] == 1 else (r[1], r[0])
        assert orc.reversi_score(x, o)[0] == winners[0, g]


# ---------------------------------------------------------------- players / full size
def test_mcts_player_plugs_into_the_reference_style_loop():
    import betazero_amd as bz
    p1, p2 = bz.MCTSPlayer(1, sims=200, evaluator="uniform"), bz.RandomPlayer()
    import random
    random.seed(0)
    for _ in range(5):
        positions, winner = bz.TicTacToeHeadless(p1, p2).play()
        assert winner in (1, 0)  # 200-sim MCTS as X does not lose to a random player
    g = bz.ReversiHeadless(bz.MCTSPlayer(1, sims=50, evaluator="hash"), bz.ReversiRandomPlayer(-1))
    positions, winner = g.play()
    assert g.board.is_game_over()
    N, _, _, _ = orc.mcts_search(orc.GAME_REVERSI, *bz.ReversiBoard().bits(1), 1, 50, orc.EVAL_HASH)
    pl = bz.MCTSPlayer(1, sims=50, evaluator="hash")
    assert pl.get_move(bz.ReversiBoard()) == divmod(int(np.argmax(N)), 8)
    assert np.array_equal(pl.last_visits, N)


def test_cfg2_full_size_ttt_properties():
    """BASELINE cfg 2 at full size: 65,536 concurrent TTT games, 50 sims, uniform
    priors.  All games are identical by construction, so one oracle game pins
    all of them; plus checksum-style invariants."""
    eng = _engine("ttt", 65536, 50, "uniform")
    eng.run_iteration()
    ex = eng.examples()
    r = orc.selfplay_game(orc.GAME_TTT, 0, 50, orc.EVAL_UNIFORM)
    winners, lens = eng.winners()
    assert (lens[0] == len(r["own"])).all() and (winners[0] == r["winner"]).all()
    T = len(r["own"])
    assert np.array_equal(ex.own.reshape(65536, T), np.broadcast_to(r["own"], (65536, T)))
    assert np.array_equal(ex.pi.reshape(65536, T, 9).view(np.uint32),
                          np.broadcast_to(r["pi"].view(np.uint32), (65536, T, 9)))
    cnt = eng.counters()
    assert cnt["n_sims"] == 65536 * 50 * T and cnt["n_edges_backed"] > cnt["n_sims"]


def test_cfg3_size_reversi_batch_properties():
    """4096 concurrent Reversi games (cfg 3 batch, hash evaluator, 32 sims):
    per-game oracle spot checks + invariants over the whole batch."""
    eng = _engine("reversi", 4096, 32, "hash", temp_moves=8, openings=1)
    eng.run_iteration()
    ex = eng.examples()
    winners, lens = eng.winners()
    assert (lens[0] >= 7).all() and (lens[0] <= 58).all()  # 9 plies is the shortest Reversi game (2 are openings)
    assert np.allclose(ex.pi.sum(1), 1.0, atol=1e-6)
    assert np.array_equal(ex.z, (winners[0][ex.game] * ex.mover).astype(np.int8))
    assert (ex.own & ex.opp == 0).all()
    for g in (0, 1, 777, 4095):
        r = orc.selfplay_game(orc.GAME_REVERSI, g, 32, orc.EVAL_HASH, 8, 1, 0)
        m = ex.game == g
        assert np.array_equal(ex.act[m], r["act"]) and np.array_equal(ex.pi[m].view(np.uint32), r["pi"].view(np.uint32))
    assert len(set(ex.act[ex.ply == 0].tolist())) > 1  # openings + tau=1 diversify the games


# ---------------------------------------------------------------- SURVEY 8(f) rows
def test_d4_augmentation_and_dedupe_match_the_reference_transforms():
    """bz_augment_d4_batch + dedupe vs the reference's 8 torch transforms applied to its own CSV
    (fixture augment.npz): same rows, same insertion order."""
    from betazero_amd.augment import augment_examples
    from betazero_amd.engine import Examples
    d = np.load(os.path.join(G, "ttt_csv.npz"))
    a = np.load(os.path.join(G, "augment.npz"))
    st, ac = d["states"], d["actions"]
    w = (1 << np.arange(9)).astype(np.uint64)
    own = ((st == 1) * w).sum(1).astype(np.uint64)
    opp = ((st == -1) * w).sum(1).astype(np.uint64)
    ex = Examples(own, opp, ac.astype(np.float32), np.zeros(180, np.int8), np.ones(180, np.int8),
                  ac.argmax(1).astype(np.uint8), np.arange(180) // 9, np.arange(180) % 9, 3)
    full = augment_examples(ex, dedupe=False)
    assert len(full) == 1440
    for t in range(8):  # every transform equals the reference's index map
        m = a["maps3"][t]
        assert np.array_equal(full.states()[t::8].reshape(180, 9), st[:, m])
        assert np.array_equal(full.pi[t::8], ac[:, m].astype(np.float32))
    out = augment_examples(ex, dedupe=True)
    assert len(out) == len(a["states"]) == 329
    assert np.array_equal(out.states().reshape(-1, 9), a["states"])
    assert np.array_equal(out.pi.astype(np.int64), a["actions"])
    assert np.array_equal(out.act, a["actions"].argmax(1))
    # 8x8 with the pass column: maps of the 8x8 grid, pass probability stays in column 64
    rng = np.random.default_rng(2)
    n = 50
    o8 = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64)
    p8 = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64) & ~o8
    pi8 = rng.random((n, 65)).astype(np.float32)
    ex8 = Examples(o8, p8, pi8, np.zeros(n, np.int8), np.ones(n, np.int8), pi8[:, :64].argmax(1).astype(np.uint8),
                   np.arange(n), np.zeros(n, np.int32), 8)
    f8 = augment_examples(ex8, dedupe=False)
    for t in range(8):
        m = a["maps8"][t]
        assert np.array_equal(f8.states()[t::8].reshape(n, 64), ex8.states().reshape(n, 64)[:, m])
        assert np.array_equal(f8.pi[t::8, :64], pi8[:, m]) and np.array_equal(f8.pi[t::8, 64], pi8[:, 64])


def test_dedupe_terminates_on_nan_rows_and_keeps_first_occurrences():
    """a pi row holding a NaN (a diverged net's targets) is not equal to itself under float comparison: the dedupe compares
    bit patterns and never hands a group's head to the next pass, so it ends, keeps such a row once and drops its exact
    copies like any other row"""
    from betazero_amd.augment import augment_examples
    from betazero_amd.engine import Examples
    rng = np.random.default_rng(4)
    n = 12
    own = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64)
    opp = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64) & ~own
    pi = rng.random((n, 65)).astype(np.float32)
    pi[3, 10] = np.nan
    own[7], opp[7], pi[7] = own[3], opp[3], pi[3]          # an exact copy of the NaN row
    own[9], opp[9], pi[9] = own[2], opp[2], pi[2]          # and of an ordinary one
    ex = Examples(own, opp, pi, np.zeros(n, np.int8), np.ones(n, np.int8), np.zeros(n, np.uint8), np.arange(n), np.zeros(n, np.int32), 8)
    full = augment_examples(ex, dedupe=False)
    out = augment_examples(ex, dedupe=True)
    keys = [(int(a), int(b), p.tobytes()) for a, b, p in zip(full.own, full.opp, full.pi)]
    first = [i for i, k in enumerate(keys) if k not in keys[:i]]
    assert len(out) == len(first) < len(full)
    assert np.array_equal(out.own, full.own[first]) and np.array_equal(out.pi.view(np.uint32), full.pi[first].view(np.uint32))
    assert int(np.isnan(out.pi).any(1).sum()) == int(np.isnan(full.pi[first]).any(1).sum()) > 0


def test_holdout_split_and_validation_line_on_the_engine_net():
    """the reference's 80 / 20 split and per-epoch validation line (SL/train.py:66-78, :121-146) for (s, pi, z) rows:
    holdout_split = a random partition with int(0.2 n) validation rows; validate() = policy CE, value MSE and top-1
    agreement on the ENGINE's net (DeviceNet.forward, bf16 MFMA) against the same quantities from the torch module's
    fp32 forward on the same rows (bf16 tolerance), and an index subset equals the selected rows"""
    import torch.nn.functional as F
    from betazero_amd.engine import DeviceExamples, Examples
    from betazero_amd.net import DeviceNet
    from betazero_amd.train import holdout_split, planes_from_bits, select_rows, validate
    rng = np.random.default_rng(8)
    n = 700
    own, opp = _positions(n, seed=17)
    pi = rng.random((n, 65)).astype(np.float32) ** 6
    pi /= pi.sum(1, keepdims=True)
    z = rng.integers(-1, 2, n).astype(np.int8)
    ex = DeviceExamples.from_host(Examples(own, opp, pi, z, np.ones(n, np.int8), np.zeros(n, np.uint8), np.arange(n), np.zeros(n, np.int32), 8))
    gen = torch.Generator(device=DEV).manual_seed(3)
    tr, va = holdout_split(n, 0.2, gen, DEV)
    assert len(va) == int(n * 0.2) and len(tr) == n - len(va)
    assert sorted(torch.cat([tr, va]).cpu().tolist()) == list(range(n))
    m = _net(64, 2, bf16=True)
    dn = DeviceNet.from_module(m, 256)                     # smaller than the validation set: validate() walks it in chunks
    got = validate(dn, ex)
    with torch.no_grad():
        lg, v = m.cuda()(planes_from_bits(ex.own, ex.opp))
    ce = float(-(ex.pi * F.log_softmax(lg.float(), dim=1)).sum(1).mean())
    mse = float(((v.float() - ex.z.float()) ** 2).mean())
    top1 = float((lg.argmax(1) == ex.pi.argmax(1)).float().mean())
    assert got["rows"] == n and abs(got["policy_ce"] - ce) < 2e-3 and abs(got["value_mse"] - mse) < 2e-3
    assert abs(got["top1"] - top1) <= 3 / n and abs(got["loss"] - (ce + mse)) < 4e-3
    sub = validate(dn, ex, va)
    sel = validate(dn, select_rows(ex, va))
    assert sub == sel and sub["rows"] == len(va)
    assert validate(dn, select_rows(ex, va[:0]))["policy_ce"] is None


def test_device_resident_pipeline_gather_augment_train_never_visits_the_host(monkeypatch):
    """pack kernels -> all_gather_packed -> unpack -> augment_examples -> train_step up to loss.backward() with every
    device-to-host door bolted (Tensor.cpu / numpy / tolist / item raise; only a packed block's 256-byte header -- the
    row count -- may cross): the rows stay on the GPU the whole way
    (the pipeline of SL/train.py:24-52 then :85-136).  The device rows equal the host path's rows, before and after
    augmentation + dedupe."""
    import torch.distributed as dist
    from betazero_amd import distributed as bd
    from betazero_amd.augment import augment_examples
    from betazero_amd.net import DeviceNet
    from betazero_amd.train import make_optimizer, train_step
    m = _net(64, 2, bf16=True)
    dn = DeviceNet.from_module(m, 64)
    engs = [_engine("reversi", 24, 8, "net_bf16", net=dn, temp_moves=8, openings=1, game_id_base=24 * i, game_id_stride=48)
            for i in range(2)]
    for e in engs:
        e.run_iteration()
    host = [e.examples() for e in engs]
    host_aug = augment_examples(host[0], dedupe=True)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    own_pg = not dist.is_initialized()
    if own_pg:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        opt = make_optimizer(m.cuda(), lr=1e-3)

        with monkeypatch.context() as mp:
            for name in ("cpu", "numpy", "tolist", "item"):
                def raiser(self, *a, _n=name, _orig=getattr(torch.Tensor, name), **k):
                    # (Adam's step counter is a host tensor: its .item() moves nothing off the GPU.  The one device read
                    # that IS part of the path: a packed block's 256-byte header -- the row count -- per gathered rank)
                    if self.is_cuda and not (_n == "cpu" and self.dtype == torch.uint8 and self.numel() == 256):
                        raise AssertionError(f"Tensor.{_n}() on a device tensor inside the device-resident pipeline")
                    return _orig(self, *a, **k)
                mp.setattr(torch.Tensor, name, raiser)
            dex = bd.gather_examples_device(engs)           # ONE all-gather (RCCL), unpacked where it lands
            daug = augment_examples(dex, dedupe=True)
            idx = torch.arange(0, len(daug), 3, device="cuda:0")
            loss, ce, mse = train_step(m, opt, daug, idx)   # forward, backward, optimiser step
        assert dex.own.is_cuda and daug.pi.is_cuda and loss.is_cuda
    finally:
        if own_pg:
            dist.destroy_process_group()
    assert np.isfinite(float(loss))
    got = dex.cpu()
    for i, h in enumerate(host):  # the device rows of engine i == its host rows (same order)
        sel = (got.game >= 24 * i) & (got.game < 24 * (i + 1))
        assert np.array_equal(got.own[sel], h.own) and np.array_equal(got.pi[sel].view(np.uint32), h.pi.view(np.uint32))
        assert np.array_equal(got.z[sel], h.z) and np.array_equal(got.act[sel], h.act) and np.array_equal(got.ply[sel], h.ply)
        assert np.array_equal(got.game[sel], h.game)
    d0 = augment_examples(engs[0].device_examples(), dedupe=True).cpu()
    assert np.array_equal(d0.own, host_aug.own) and np.array_equal(d0.opp, host_aug.opp) and np.array_equal(d0.act, host_aug.act)
    assert np.array_equal(d0.pi.view(np.uint32), host_aug.pi.view(np.uint32)) and np.array_equal(d0.z, host_aug.z)


def test_graphed_train_step_equals_the_eager_step():
    """GraphedTrainStep (the training step captured into a HIP graph and replayed) against train_step (eager) from
    the same weights on the same batches: same losses step by step (same kernels, same order), and the weights move."""
    import copy
    from betazero_amd.engine import DeviceExamples
    from betazero_amd.train import GraphedTrainStep, make_optimizer, train_step
    rng = np.random.default_rng(11)
    n = 512
    own, opp = _positions(n, seed=91)
    pi = rng.random((n, 65)).astype(np.float32); pi /= pi.sum(1, keepdims=True)
    z = rng.integers(-1, 2, n).astype(np.int8)
    from betazero_amd.engine import Examples
    ex = DeviceExamples.from_host(Examples(own, opp, pi, z, np.ones(n, np.int8), np.zeros(n, np.uint8), np.arange(n),
                                           np.zeros(n, np.int32), 8))
    m1 = _net(64, 2, seed=5).cuda()
    m2 = copy.deepcopy(m1)
    opt = make_optimizer(m1, lr=1e-3)
    g = GraphedTrainStep(m2, lr=1e-3, batch=128)
    for s in range(5):
        idx = torch.arange(128 * (s % 4), 128 * (s % 4) + 128, device="cuda:0")
        a = torch.stack(train_step(m1, opt, ex, idx)).cpu().numpy()
        b = g(ex, idx).cpu().numpy()
        assert np.allclose(a, b, rtol=2e-3, atol=2e-3), (s, a, b)
    w0 = _net(64, 2, seed=5).stem.weight
    assert (m2.stem.weight.detach().cpu() - w0).abs().max() > 1e-4


def test_training_step_closes_the_loop():
    from betazero_amd.net import DeviceNet
    from betazero_amd.train import make_optimizer, refresh_device_net, train_step
    m = _net(128, 6, bf16=True)
    dn = DeviceNet.from_module(m, 64)
    eng = _engine("reversi", 32, 8, "net_bf16", net=dn, temp_moves=8, openings=1)
    eng.run_iteration()
    ex = eng.examples()
    opt = make_optimizer(m, lr=1e-3)
    losses = [float(train_step(m, opt, ex)[0]) for _ in range(6)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    own, opp = _dev_u64(ex.own[:16]), _dev_u64(ex.opp[:16])
    before = dn.forward(own, opp)[0].cpu().numpy()
    refresh_device_net(dn, m)
    after = dn.forward(own, opp)[0].cpu().numpy()
    assert np.abs(after - before).max() > 1e-4  # the engine's net now holds the trained weights
    import copy
    mm = copy.deepcopy(m).cpu().round_to_bf16_()
    olg, _ = orc.Net(128, 6, 64, mm.flat_params()).forward(ex.own[:16], ex.opp[:16], bf16=True)
    assert np.abs(after - olg).max() < 2e-3


def test_minimax_kernels_match_the_reference_fixture():
    """the arena's opponent kernels (one game per lane) against fixture F9 = decisions of the reference's
    OptimalPlayer classes (reversi_players.py:35-77, players.py:30-70), generated by importing them"""
    d = np.load(os.path.join(G, "minimax_players.npz"))
    rows = d["reversi"]
    L = _lib.lib()
    for size in (4, 6, 8):
        for depth in sorted(set(rows[rows[:, 0] == size][:, 1].tolist())):
            r = rows[(rows[:, 0] == size) & (rows[:, 1] == depth)]
            sym = r[:, 2].astype(np.int64) - 1
            own = np.where(sym == 1, r[:, 3], r[:, 4]).astype(np.uint64)
            opp = np.where(sym == 1, r[:, 4], r[:, 3]).astype(np.uint64)
            n = len(r)
            od, pd_ = _dev_u64(own), _dev_u64(opp)
            mv = torch.empty(n, dtype=torch.int8, device=DEV)
            sc = torch.empty(n, dtype=torch.int16, device=DEV)
            _lib.check(L.bz_reversi_minimax_batch(od.data_ptr(), pd_.data_ptr(), None, n, size, depth, mv.data_ptr(),
                                                  sc.data_ptr(), _stream()))
            torch.cuda.synchronize()
            exp_mv = np.where(r[:, 5] == 255, -1, r[:, 5].astype(np.int64))
            assert np.array_equal(mv.cpu().numpy().astype(np.int64), exp_mv), (size, depth)
            assert np.array_equal(sc.cpu().numpy().astype(np.int64), r[:, 6].astype(np.int64) - 2000), (size, depth)
    t = d["ttt"]
    cur = t[:, 2] - 1
    own = torch.as_tensor(np.where(cur == 1, t[:, 0], t[:, 1]).astype(np.int16)).to(DEV)
    opp = torch.as_tensor(np.where(cur == 1, t[:, 1], t[:, 0]).astype(np.int16)).to(DEV)
    symd = torch.as_tensor(cur.astype(np.int8)).to(DEV)
    act = torch.ones(len(t), dtype=torch.uint8, device=DEV)
    act[::7] = 0  # masked-out games report "no move"
    mv = torch.empty(len(t), dtype=torch.int8, device=DEV)
    sc = torch.empty(len(t), dtype=torch.int16, device=DEV)
    _lib.check(L.bz_ttt_minimax_batch(own.data_ptr(), opp.data_ptr(), symd.data_ptr(), act.data_ptr(), len(t),
                                      mv.data_ptr(), sc.data_ptr(), _stream()))
    torch.cuda.synchronize()
    on = act.cpu().numpy() == 1
    assert np.array_equal(mv.cpu().numpy()[on].astype(np.int64), t[on, 3]) and (mv.cpu().numpy()[~on] == -1).all()
    assert np.array_equal(sc.cpu().numpy()[on].astype(np.int64), t[on, 4] - 1)


def test_arena_batched_mcts_never_loses_to_minimax_at_tictactoe():
    """SURVEY 8(f) row 3, batched: 64 concurrent games of MCTS (5000 sims, uniform priors, v = 0 at non-terminal
    leaves, so it learns from terminal nodes only) against the reference's full-depth minimax (kernel), MCTS as X in
    half of them and as O in the other half: tic-tac-toe is a draw under best play, so MCTS must never lose.
    (At 800 simulations the averaging backup still loses about 1 game in 12 as O.)"""
    from betazero_amd.arena import play_arena
    res = play_arena("ttt", 64, 5000, evaluator="uniform", seed=1)
    s = res.summary()
    print("ttt arena:", s)
    assert s["losses"] == 0 and s["games"] == 64 and (res.plies >= 5).all() and (res.plies <= 9).all()
    assert set(res.mcts_colour.tolist()) == {1, -1}


def test_arena_random_openings_make_the_games_different_and_reproducible():
    """both arena players are deterministic (two distinct games without help): `opening_plies` seeded random legal
    moves diversify them; the same seed replays the same games, every opening move is legal (the env step would
    raise), and the players take over afterwards"""
    from betazero_amd.arena import play_arena
    a = play_arena("reversi", 64, 32, opponent_depth=1, evaluator="uniform", seed=3, opening_plies=4)
    b = play_arena("reversi", 64, 32, opponent_depth=1, evaluator="uniform", seed=3, opening_plies=4)
    seq = lambda r: [tuple(int(act[g]) for act, _ in r.moves) for g in range(64)]  # noqa: E731
    assert seq(a) == seq(b) and np.array_equal(a.winner, b.winner)
    assert len(set(seq(a))) >= 32, len(set(seq(a)))
    assert len({s[:4] for s in seq(a)}) >= 16  # the openings themselves differ
    c = play_arena("reversi", 64, 32, opponent_depth=1, evaluator="uniform", seed=3)
    assert len(set(seq(c))) <= 2
    t = play_arena("ttt", 32, 200, evaluator="uniform", seed=2, opening_plies=2)
    assert t.summary()["games"] == 32 and len({tuple(int(act[g]) for act, _ in t.moves[:2]) for g in range(32)}) >= 8


def test_arena_batched_reversi_mcts_vs_depth_limited_minimax():
    """64 concurrent 8x8 games, MCTS (300 sims, hash evaluator -- there is no trained net in this repo, so no
    strength is asserted) vs the reference's OptimalPlayer at depth 2 (kernel): every game must be a legal game of
    Reversi by the oracle's rules with the winner the oracle computes, and every minimax move must equal the
    scalar library call on that position."""
    import ctypes as C
    from betazero_amd.arena import play_arena
    B = 64
    res = play_arena("reversi", B, 300, opponent_depth=2, evaluator="hash", seed=5)
    print("reversi arena (untrained evaluator):", res.summary())
    L = _lib.lib()
    for g in range(0, B, 5):
        x, o, cur = 0x0000001008000000, 0x0000000810000000, 1  # reversi_board.py:9-11 (+1 on the main diagonal)
        for act, mover in res.moves:
            if mover[g] == 0:
                continue
            assert mover[g] == cur
            own, opp = (x, o) if cur == 1 else (o, x)
            a = int(act[g])
            assert orc.reversi_legal(own, opp) >> a & 1
            if cur != res.mcts_colour[g]:  # the minimax side: same decision as the scalar entry point
                mv, sc = C.c_int32(), C.c_int32()
                _lib.check(L.bz_reversi_minimax(own, opp, 8, 2, C.byref(mv), C.byref(sc)))
                assert mv.value == a or mv.value == -1
            no, np_, _ = orc.reversi_apply(own, opp, 8, a >> 3, a & 7)
            x, o = (no, np_) if cur == 1 else (np_, no)
            if orc.reversi_game_over(x, o):
                break
            cur = -cur
            own, opp = (x, o) if cur == 1 else (o, x)
            if orc.reversi_legal(own, opp) == 0:  # pass rule, reversi_terminal.py:31-35
                cur = -cur
        assert orc.reversi_game_over(x, o) and orc.reversi_score(x, o)[0] == res.winner[g]


def test_net_fp8_mfma_vs_oracle_fp8_emulation():
    """fp8 tower (e4m3 weights with per-channel power-of-two scales, e4m3(x*16) activations,
    MX-scaled 32x32x64 MFMA, fp32 accumulate) vs the oracle quantising at the same points.
    Accumulation order differs, and one e4m3 ulp is 2^-3 relative, so a handful of activations
    land on the neighbouring code: tolerance 0.02 abs on logits / 0.01 on value, about 2x / 4x the measured
    0.009 / 0.002 (fp8 run is BASELINE cfg 5, not the headline path); the fp8-vs-fp32 gap is printed."""
    from betazero_amd.net import DeviceNet
    from betazero_amd.quant import fake_quantize_fp8_
    m = fake_quantize_fp8_(_net(128, 6))
    n = 41
    own, opp = _positions(n, seed=5)
    dn = DeviceNet.from_module(m, 64)
    lg, v = dn.forward(_dev_u64(own), _dev_u64(opp), fp8=True)
    on = orc.Net(128, 6, 64, m.flat_params())
    olg, ov = on.forward(own, opp, bf16=2)
    err_l = np.abs(lg.cpu().numpy() - olg).max()
    err_v = np.abs(v.cpu().numpy() - ov).max()
    flg, fv = on.forward(own, opp, bf16=0)
    print("fp8 net max |dlogit|", err_l, "max |dv|", err_v, "| fp8-vs-fp32 (same fake-quantised weights): logits",
          np.abs(lg.cpu().numpy() - flg).max(), "value", np.abs(v.cpu().numpy() - fv).max(),
          "| logit std", flg.std())
    assert err_l < 0.02 and err_v < 0.01
    # self-play with the fp8 net in the loop is legal and terminates
    eng = _engine("reversi", 8, 8, "net_fp8", net=dn, temp_moves=8, openings=1)
    eng.run_iteration()
    assert (eng.winners()[1][0] > 20).all()


def test_step_batch_empty_and_ragged_sizes():
    """n = 0 is a no-op; n not a multiple of the 4-games-per-lane vector width takes the tail path"""
    d = np.load(os.path.join(G, "reversi_positions.npz"))
    pos = d["pos"][d["pos"][:, 0] == 8]
    L = _lib.lib()
    assert L.bz_reversi_step_batch(1, 1, 1, 0, 1, 1, 1, 1, 1, 0) == 0
    assert L.bz_reversi_step_batch(None, None, None, 5, None, None, None, None, None, 0) == _lib.BZ_EINVAL
    for n in (1, 2, 3, 5, 7, 130):
        own, opp, legal = pos[:n, 1].copy(), pos[:n, 2].copy(), pos[:n, 3]
        act = np.array([(int(l) & -int(l)).bit_length() - 1 if l else 64 for l in legal], dtype=np.uint8)
        on, pn, lg, st, w = _reversi_step(own, opp, act)
        for i in range(n):
            a = int(act[i])
            exp = (int(own[i]), int(opp[i]), 0) if a == 64 else orc.reversi_apply(int(own[i]), int(opp[i]), 8, a >> 3, a & 7)
            assert (int(pn[i]), int(on[i])) == (exp[0], exp[1]) and st[i] != _lib.ST_ILLEGAL
            assert int(lg[i]) == orc.reversi_legal(exp[1], exp[0], 8)


def test_engine_ragged_batch_sizes_match_oracle():
    """B = 1, 3, 5 games: partial lane groups / partial waves / partial net tiles"""
    from betazero_amd.net import DeviceNet
    m = _net(128, 6, bf16=True)
    dn = DeviceNet.from_module(m, 8)
    for B in (1, 3, 5):
        _check_selfplay("reversi", B, 10, "hash", 4, 1, 9, base=B)
        _check_selfplay("ttt", B, 12, "hash", 2, 0, 1)
        eng = _engine("reversi", B, 4, "net_bf16", net=dn, openings=1)
        eng.run_iteration()
        assert (eng.winners()[1][0] > 8).all()


def test_search_wide_nodes_and_cpuct_variants_vs_oracle():
    """positions with 17..28 legal moves (the >16-children chunk path of the 16-lane walk, arbitrary
    unreachable stones included) and other exploration constants: root N/W/P bit-exact vs the oracle"""
    rng = np.random.default_rng(3)
    rows = []
    while len(rows) < 40:  # seeded random stones, filtered by the oracle: 17..21 legal moves at the root
        u = rng.random(64)
        own_ = sum(1 << i for i in range(64) if u[i] < 0.12)
        opp_ = sum(1 << i for i in range(64) if 0.12 <= u[i] < 0.52)
        c = bin(orc.reversi_legal(own_, opp_)).count("1")
        if c >= 17:
            rows.append((own_, opp_, 1 if len(rows) % 2 == 0 else -1, c))
    assert max(r[3] for r in rows) >= 19
    own = np.array([r[0] for r in rows], dtype=np.uint64)
    opp = np.array([r[1] for r in rows], dtype=np.uint64)
    tm = np.array([r[2] for r in rows], dtype=np.int8)
    for c_puct, sims in ((1.5, 300), (0.25, 150), (6.0, 150)):
        eng = _engine("reversi", len(rows), sims, "hash", c_puct=c_puct)
        eng.set_roots(own, opp, tm)
        eng.search()
        N, W, P = eng.root_stats()
        eng.status()
        for g in range(len(rows)):
            n, w, p, _ = orc.mcts_search(orc.GAME_REVERSI, int(own[g]), int(opp[g]), int(tm[g]), sims, orc.EVAL_HASH,
                                         c_puct=c_puct)
            assert np.array_equal(N[g], n), (g, c_puct)
            assert np.array_equal(W[g].view(np.uint32), w.view(np.uint32))
            assert np.array_equal(P[g].view(np.uint32), p.view(np.uint32))


def test_selfplay_longer_searches_vs_oracle_bitexact():
    _check_selfplay("reversi", 64, 100, "hash", 8, 1, 21, base=1000)
    _check_selfplay("ttt", 256, 200, "hash", 9, 0, 4)      # 200 sims: the tree saturates into terminals
    _check_selfplay("ttt", 64, 60, "uniform", 3, 0, 2, base=7)


@pytest.mark.parametrize("size", [4, 6, 1, 2, 3, 5, 7])
def test_reversi_step_batch_small_boards_on_golden_games(size):
    """the reference's 4x4 / 6x6 boards (reversi_board.py:93, reversi_terminal.py:46) through the batched kernel, and the
    other sizes its generic constructor accepts that fit the 64-bit boards (fixture F11: 1, 2, 3, 5, 7)"""
    d = np.load(os.path.join(G, "reversi_random_games.npz" if size in (4, 6) else "reversi_other_sizes.npz"))
    rows = d["rows"][d["rows"][:, 1] == size]
    assert len(rows) > 0
    cur = rows[:, 3].astype(np.int64) - 1
    own = np.where(cur == 1, rows[:, 4], rows[:, 5])
    opp = np.where(cur == 1, rows[:, 5], rows[:, 4])
    act = np.where(rows[:, 7] == 255, 64, rows[:, 7]).astype(np.uint8)
    n = len(rows)
    o, p, a = _dev_u64(own), _dev_u64(opp), torch.as_tensor(act).to(DEV)
    on, pn, lg = (torch.empty(n, dtype=torch.int64, device=DEV) for _ in range(3))
    st = torch.empty(n, dtype=torch.uint8, device=DEV)
    w = torch.empty(n, dtype=torch.int8, device=DEV)
    _lib.check(_lib.lib().bz_reversi_step_batch_sized(o.data_ptr(), p.data_ptr(), a.data_ptr(), n, size, on.data_ptr(),
                                                      pn.data_ptr(), lg.data_ptr(), st.data_ptr(), w.data_ptr(), _stream()))
    on, pn, lg, st = on.cpu().numpy().view(np.uint64), pn.cpu().numpy().view(np.uint64), lg.cpu().numpy().view(np.uint64), st.cpu().numpy()
    for i in range(n):
        exp = (int(own[i]), int(opp[i]), 0) if act[i] == 64 else orc.reversi_apply(int(own[i]), int(opp[i]), size, int(act[i]) >> 3, int(act[i]) & 7)
        assert (int(pn[i]), int(on[i])) == (exp[0], exp[1])
        assert int(lg[i]) == orc.reversi_legal(exp[1], exp[0], size)
        assert (st[i] == _lib.ST_TERMINAL) == bool(rows[i, 9])
    same = rows[1:, 0] == rows[:-1, 0]
    assert np.array_equal(lg[:-1][same], rows[1:, 6][same])  # next row's legal mask in the fixture


def test_small_reversi_boards_search_and_selfplay_vs_twin_and_oracle():
    """6x6 / 4x4 Reversi (the reference's demo sizes) through the same tree kernels: root statistics and
    whole self-play games vs the Python twin over the reference's boards (fixture) and vs the oracle"""
    d = np.load(os.path.join(G, "mcts_twin.npz"))
    games = {"reversi6": orc.GAME_REVERSI6, "reversi4": orc.GAME_REVERSI4}
    n_checked = 0
    for case in _cases("search"):
        if case["game"] not in games:
            continue
        eng = _engine(case["game"], 2, case["sims"], case["eval"])
        eng.set_roots([case["own"]] * 2, [case["opp"]] * 2, [case["to_move"]] * 2)
        eng.search()
        N, W, P = eng.root_stats()
        eng.status()
        i = case["id"]
        assert np.array_equal(N[1], d[f"s{i}_N"]) and np.array_equal(W[1].view(np.uint32), d[f"s{i}_W"].view(np.uint32))
        assert np.array_equal(P[0].view(np.uint32), d[f"s{i}_P"].view(np.uint32))
        n_checked += 1
    for case in _cases("selfplay"):
        if case["game"] not in games:
            continue
        eng = _engine(case["game"], 1, case["sims"], case["eval"], temp_moves=case["temp_moves"], seed=case["seed"],
                      game_id_base=case["gid"])
        eng.run_iteration()
        ex = eng.examples()
        i = case["id"]
        assert np.array_equal(ex.own, d[f"g{i}_own"]) and np.array_equal(ex.act, d[f"g{i}_act"])
        assert np.array_equal(ex.pi.view(np.uint32), d[f"g{i}_pi"].view(np.uint32))
        assert eng.winners()[0][0, 0] == case["winner"]
        n_checked += 1
    assert n_checked == 6
    for name, og in games.items():  # a batch of games vs the oracle
        eng = _engine(name, 40, 30, "hash", temp_moves=4, seed=3, game_id_base=50)
        eng.run_iteration()
        ex = eng.examples()
        winners, lens = eng.winners()
        for g in range(40):
            r = orc.selfplay_game(og, 50 + g, 30, orc.EVAL_HASH, 4, 0, 3)
            m = ex.game == 50 + g
            assert lens[0, g] == len(r["own"]) and winners[0, g] == r["winner"]
            assert np.array_equal(ex.act[m], r["act"]) and np.array_equal(ex.pi[m].view(np.uint32), r["pi"].view(np.uint32))
        assert ex.states().shape[1:] == ({"reversi6": 6, "reversi4": 4}[name],) * 2


def test_mcts_player_on_the_reference_demo_board_sizes():
    import random
    import betazero_amd as bz
    random.seed(2)
    for size in (4, 6):  # reversi_terminal.py:42-47 plays on 4x4, reversi_gui.py:105 on 6x6
        g = bz.ReversiHeadless(bz.MCTSPlayer(1, sims=60, evaluator="hash"), bz.ReversiRandomPlayer(-1), size=size)
        positions, winner = g.play()
        assert g.board.is_game_over() and winner in (-1, 0, 1)


# ---------------------------------------------------------------- BASELINE cfg 3 / cfg 5 at their real settings
def _ex_rows(eng, n_rows):
    """first n_rows example rows of every slot's round-0 game (also for unfinished games)"""
    t = eng.example_tensors()
    return {k: t[k][0, :, :n_rows].cpu().numpy() for k in ("own", "opp", "pi", "mover", "act")}


def test_cfg3_800_sims_tree_geometry_vs_oracle_bitexact():
    """cfg 3's search depth (800 simulations: 802-node / 27,268-edge trees, long paths, u32 edge offsets)
    with the synthetic hash evaluator, 64 games from the cfg-3 openings: root N/W/P of the first two searched
    moves and the two played moves (tau = 1 sampling), all bit-exact vs the oracle."""
    B, sims = 64, 800
    eng = _engine("reversi", B, sims, "hash", temp_moves=8, openings=1, seed=0)
    eng.reset_games()
    for mv in range(2):
        own, opp, tm, st = eng.positions()
        assert (st == 0).all()
        eng.search()
        N, W, P = eng.root_stats()
        eng.status()
        for g in range(B):
            n, w, p, _ = orc.mcts_search(orc.GAME_REVERSI, int(own[g]), int(opp[g]), int(tm[g]), sims, orc.EVAL_HASH)
            assert N[g].sum() == sims and np.array_equal(N[g], n), (mv, g)
            assert np.array_equal(W[g].view(np.uint32), w.view(np.uint32)), (mv, g)
            assert np.array_equal(P[g].view(np.uint32), p.view(np.uint32)), (mv, g)
        eng.play(False)
    rows = _ex_rows(eng, 2)
    for g in range(B):
        r = orc.selfplay_game(orc.GAME_REVERSI, g, sims, orc.EVAL_HASH, 8, 1, 0, max_moves=2)
        assert np.array_equal(rows["own"][g].view(np.uint64), r["own"]) and np.array_equal(rows["act"][g], r["act"])
        assert np.array_equal(rows["pi"][g].view(np.uint32), r["pi"].view(np.uint32))
    cnt = eng.counters()
    assert cnt["n_sims"] == 2 * B * sims


@pytest.mark.parametrize("stepwise", [True, False], ids=["step_kernels", "fused"])
def test_deep_narrow_trees_vs_oracle_bitexact(stepwise):
    """walks deeper than the lane group is wide: with c_puct = 0.05 the search exploits (mean 12 nodes per walk, many
    walks beyond 16 levels), so the path's entries past the 16 that live in registers, the backup's loop over them and the
    second and third chunk of wide nodes are all on the road.  Root N / W / P and every work counter vs the oracle, for
    the step kernels (k_tree_step, hash evaluator through the step API) and the fused search."""
    own, opp = _positions(40, seed=31)
    own[:8], opp[:8] = 0x0000000810000000, 0x0000001008000000   # the start position: the deepest trees (mean 12 nodes per walk)
    tm = np.where(np.arange(40) % 2 == 0, 1, -1).astype(np.int8)
    sims, c = 400, 0.05
    eng = _engine("reversi", 40, sims, "hash", c_puct=c)
    eng.set_roots(own, opp, tm)
    eng.reset_counters()
    if stepwise:
        eng.root_begin(); eng.evaluate(); eng.expand_backup()
        for s_ in range(sims):
            eng.select(s_); eng.evaluate(); eng.expand_backup()
    else:
        eng.search()
    N, W, P = eng.root_stats()
    eng.status()
    tot, deepest = {}, 0.0
    for g in range(40):
        n, w, p, cnt = orc.mcts_search(orc.GAME_REVERSI, int(own[g]), int(opp[g]), int(tm[g]), sims, orc.EVAL_HASH, c_puct=c)
        deepest = max(deepest, cnt["n_path_nodes"] / cnt["n_sims"])
        assert np.array_equal(N[g], n), g
        assert np.array_equal(W[g].view(np.uint32), w.view(np.uint32)) and np.array_equal(P[g].view(np.uint32), p.view(np.uint32)), g
        for k, v in cnt.items():
            tot[k] = tot.get(k, 0) + v
    got = eng.counters()
    for k in ("n_sims", "n_path_nodes", "n_child_scored", "n_edges_backed", "n_expanded", "n_child_written", "n_env_steps"):
        assert got[k] == tot[k], (k, got[k], tot[k])
    assert deepest > 11  # the trees really are deep: a MEAN of 12 nodes per walk puts many walks beyond 16 levels


@pytest.mark.parametrize("stepwise", [True, False], ids=["step_kernels", "fused"])
def test_search_at_the_packed_edge_limit_vs_oracle_bitexact(stepwise):
    """sims = BZ_ENGINE_MAX_SIMS = 8189: the tree then holds 8190 nodes -- every one of the 13 bits of a child id, visit
    counts up to 8189 in the 14-bit field, edge offsets up to ~70 k -- and must still equal the oracle's tree bit for bit
    (root N / W / P and the work counters); sims = 8190 is refused (tests/test_abi.py)."""
    sims = 8189
    own = np.array([0x0000000810000000, 0x0000001008000000], np.uint64)
    opp = np.array([0x0000001008000000, 0x0000000810000000], np.uint64)
    tm = np.array([1, -1], np.int8)
    eng = _engine("reversi", 2, sims, "hash")
    eng.set_roots(own, opp, tm)
    eng.reset_counters()
    if stepwise:
        eng.root_begin(); eng.evaluate(); eng.expand_backup()
        for s_ in range(sims):
            eng.select(s_); eng.evaluate(); eng.expand_backup()
    else:
        eng.search()
    N, W, P = eng.root_stats()
    eng.status()
    tot = {}
    for g in range(2):
        n, w, p, cnt = orc.mcts_search(orc.GAME_REVERSI, int(own[g]), int(opp[g]), int(tm[g]), sims, orc.EVAL_HASH)
        assert N[g].sum() == sims and np.array_equal(N[g], n), g
        assert np.array_equal(W[g].view(np.uint32), w.view(np.uint32)) and np.array_equal(P[g].view(np.uint32), p.view(np.uint32)), g
        for k, v in cnt.items():
            tot[k] = tot.get(k, 0) + v
    got = eng.counters()
    for k in ("n_sims", "n_path_nodes", "n_child_scored", "n_edges_backed", "n_expanded", "n_child_written", "n_env_steps"):
        assert got[k] == tot[k], (k, got[k], tot[k])


def _legal_per_oracle(own, opp, act):
    return all(orc.reversi_legal(int(o), int(p)) >> int(a) & 1 for o, p, a in zip(own, opp, act))


@pytest.mark.parametrize("ev,B,stagger", [("net_bf16", 4096, 0), ("net_bf16", 4096, 58), ("net_fp8", 8192, 58)],
                         ids=["openings", "all_game_phases", "cfg5_selfplay_fp8_8192_games"])
def test_cfg3_full_size_4096_games_800_sims_bf16_net_invariants(ev, B, stagger):
    """BASELINE cfg 3 exactly as bench.py runs it (4096 concurrent games, 800 sims/move, bf16 MFMA net in
    the loop, packed leaves) for two moves: no error flag, every root's visits sum to 800, pi == N/800 bit
    for bit and sums to 1, every played action legal per the oracle's rules, and the work counters add up:
    one net evaluation per expanded node, one env step per created node, and (from the openings, where no
    search reaches a terminal position) exactly sims + 1 net evaluations per search."""
    from betazero_amd.net import DeviceNet
    sims = 800
    mod = _net(128, 6, bf16=True)
    if ev == "net_fp8":  # bench.py's secondary.cfg5_selfplay: the fp8 MFMA tower as the in-loop evaluator of 8192 games
        from betazero_amd.quant import fake_quantize_fp8_
        fake_quantize_fp8_(mod)
    dn = DeviceNet.from_module(mod, B)
    eng = _engine("reversi", B, sims, ev, net=dn, temp_moves=8, openings=1, seed=0, stagger=stagger, rounds=2)
    eng.reset_games()
    eng.reset_counters()
    roots = 0
    for mv in range(2):
        own, opp, tm, st = eng.positions()
        active = st == 0
        roots += int(active.sum())
        eng.search()
        N, W, P = eng.root_stats()
        a_, f_ = eng.status()  # raises on any engine error flag (edge / depth / example overflow, terminal root)
        assert (N[active].sum(1) == sims).all()
        legal = np.array([orc.reversi_legal(int(o), int(p)) for o, p in zip(own, opp)], dtype=np.uint64)
        onb = (legal[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)
        assert (N[:, :64][onb == 0] == 0).all()  # visits only on legal actions
        assert (np.abs(P[active].sum(1) - 1.0) < 1e-5).all() and (np.abs(W) <= N + 1e-3).all()
        nex_before = eng.example_tensors()["len"].cpu().numpy().copy()
        eng.play(True)
        t = eng.example_tensors()
        # the row just written: pi == N / 800 (same single float division as the oracle's spec)
        pi_exp = (N.astype(np.float32) / np.float32(sims)).astype(np.float32)
        if stagger == 0:
            pi = t["pi"][0, :, mv].cpu().numpy()
            act = t["act"][0, :, mv].cpu().numpy()
            assert np.array_equal(pi.view(np.uint32), pi_exp.view(np.uint32))
            assert np.abs(pi.sum(1) - 1.0).max() < 1e-6
            assert _legal_per_oracle(own, opp, act)
            assert (N[np.arange(B), act] > 0).all()
        del nex_before
    cnt = eng.counters()
    assert cnt["n_sims"] == roots * sims
    # every expansion consumed one evaluation: computed by the net, or shared with an earlier node of the same search that
    # holds the same position (the evaluation cache, on by default with a net evaluator)
    evals = cnt["n_net_leaves"] + cnt["n_cache_hits"]
    assert evals == cnt["n_expanded"] and 0 < cnt["n_cache_hits"] < 0.3 * evals
    assert cnt["n_env_steps"] <= cnt["n_sims"] and evals - roots <= cnt["n_env_steps"]
    assert cnt["n_child_written"] >= cnt["n_expanded"]
    if stagger == 0:  # no terminal inside an 800-sim tree from the openings: every simulation ends in an evaluation
        assert evals == cnt["n_sims"] + roots and cnt["n_env_steps"] == cnt["n_sims"]
    else:             # late-game slots do reach terminals
        assert evals < cnt["n_sims"] + roots


@pytest.mark.parametrize("ev,game,B,sims", [("net_bf16", "reversi", 384, 800), ("net_f32", "reversi6", 40, 300), ("net_fp8", "reversi", 300, 200),
                                            ("net_f32", "reversi4", 64, 300)],
                         ids=["bf16_800", "f32_6x6_300", "fp8_200", "f32_4x4_300"])
def test_evaluation_cache_changes_no_result_and_saves_evaluations(ev, game, B, sims):
    """BZ_ENGINE_EVAL_CACHE (+ _CARRY): a leaf whose position was evaluated earlier in the same search -- or, with carry-over,
    in the slot's previous search -- takes that node's priors and value instead of an evaluator row.  Three engines on the
    same games (cache with carry-over = the default, cache inside a search only, no cache), from all game phases (staggered
    pool) and over three moves: root N / W / P of every game bit for bit, the played moves and the example rows bit for bit,
    every work counter equal -- except that the evaluator computed fewer rows: n_net_leaves + n_cache_hits is the same in
    all three, with in-search hits > 3 % at 800 simulations (the CPU measurement on the oracle,
    profiles/r05_leaf_duplication.json: 6 - 12 %) and carried-over hits on top from the second move on (about the played
    move's share of the previous search's visits).  (That the cache-ON engine equals the ORACLE, which has no cache, is what
    every net-evaluator parity test checks.)"""
    from betazero_amd.net import DeviceNet
    mod = _net(128, 6, bf16=True) if ev != "net_f32" else _net(32, 2, seed=3)
    if ev == "net_fp8":
        from betazero_amd.quant import fake_quantize_fp8_
        fake_quantize_fp8_(mod)
    dn = DeviceNet.from_module(mod, B)
    # (4x4: games of at most 12 plies, most of a 300-simulation tree is terminal nodes and repeats -- the table's worst case)
    kw = dict(net=dn, temp_moves=8, openings=1, seed=2, rounds=2, stagger={"reversi": 58, "reversi6": 20, "reversi4": 6}[game])
    engs = [_engine(game, B, sims, ev, eval_cache=mode, **kw) for mode in (True, "search", False)]
    for e in engs:
        e.reset_games(); e.reset_counters()
    for mv in range(3):
        for e in engs:
            e.search()
        N0, W0, P0 = engs[2].root_stats()
        for e in engs[:2]:
            N1, W1, P1 = e.root_stats()
            assert np.array_equal(N1, N0) and np.array_equal(W1.view(np.uint32), W0.view(np.uint32)) and np.array_equal(P1.view(np.uint32), P0.view(np.uint32)), mv
        for e in engs:
            e.play(True)
    assert engs[0].status() == engs[1].status() == engs[2].status()
    b = engs[2].example_tensors()
    for e in engs[:2]:
        a = e.example_tensors()
        for f in ("own", "opp", "act", "mover", "len"):
            assert torch.equal(a[f], b[f]), f
        assert torch.equal(a["pi"].view(torch.int32), b["pi"].view(torch.int32))
    cc, cs, c0 = (e.counters() for e in engs)
    for k in c0:
        if k not in ("n_net_leaves", "n_cache_hits", "n_cache_hits_prev"):
            assert cc[k] == cs[k] == c0[k], (k, cc[k], cs[k], c0[k])
    assert c0["n_cache_hits"] == 0 and cs["n_cache_hits_prev"] == 0
    assert cc["n_net_leaves"] + cc["n_cache_hits"] == cs["n_net_leaves"] + cs["n_cache_hits"] == c0["n_net_leaves"]
    frac_s, frac_c = cs["n_cache_hits"] / c0["n_net_leaves"], cc["n_cache_hits"] / c0["n_net_leaves"]
    print(f"evaluation cache, {ev} {game} {B} games x {sims} sims x 3 moves: inside a search {cs['n_cache_hits']} of {c0['n_net_leaves']} "
          f"evaluations shared = {frac_s:.3f}; with carry-over {cc['n_cache_hits']} ({cc['n_cache_hits_prev']} from the previous search) = {frac_c:.3f}")
    assert frac_s > (0.03 if sims >= 300 else 0.01)   # (a 200-simulation tree meets fewer transpositions than an 800-simulation one)
    assert cc["n_cache_hits_prev"] > 0.05 * c0["n_net_leaves"] and frac_c > frac_s + 0.05


def test_mcts_player_with_the_evaluation_cache_searches_like_an_engine_without():
    """the plug-in's path (bz_engine_set_roots per move, one game, a search every second ply of the game): two MCTSPlayers --
    cache with carry-over, the default -- play 16 plies; before every move a cache-less engine searches the same position:
    the player's visit counts are the engine's, bit for bit, and from each player's second move on part of its evaluations
    come from its previous search (the opponent's reply was one of the children it had searched)."""
    import betazero_amd as bz
    from betazero_amd.net import DeviceNet
    sims = 200
    dn = DeviceNet.from_module(_net(128, 6, bf16=True), 1)
    players = {1: bz.MCTSPlayer(1, sims=sims, net=dn), -1: bz.MCTSPlayer(-1, sims=sims, net=dn)}
    ref = _engine("reversi", 1, sims, "net_bf16", net=dn, eval_cache=False)
    b, side, carried = bz.ReversiBoard(), 1, 0
    for ply in range(16):
        if not b.generate_possible_moves(side):
            side = -side
            continue
        mv = players[side].get_move(b)
        own, opp = b.bits(side)
        ref.set_roots([own], [opp], [side]); ref.search()
        N, _, _ = ref.root_stats()
        assert np.array_equal(players[side].last_visits, N[0]), ply
        b = b.make_move(mv[0], mv[1], side)
        side = -side
    for pl in players.values():
        c = pl._engine("reversi").counters()
        assert c["n_net_leaves"] + c["n_cache_hits"] == c["n_expanded"]
        carried += c["n_cache_hits_prev"]
    assert carried > 0


def test_evaluation_cache_with_a_degenerate_hash_changes_nothing_either():
    """"a tag match is confirmed against the stored node's position; a full bucket just means no insertion" -- exercised: a
    test build of the library (-DBZ_EXP_TT_WEAK_HASH: every position has tag 0 and one of two buckets, so nearly every
    lookup meets a tag match that is NOT its position, and the buckets are full after 32 nodes) runs the cache-on engine
    against the cache-off engine in a child process: root statistics, moves and example rows bit for bit the same over
    three moves."""
    import subprocess
    import sys
    from betazero_amd import build
    so = build.build_variant("ttweak", ["-DBZ_EXP_TT_WEAK_HASH"])
    script = """
import numpy as np, torch, sys
sys.path.insert(0, %r)
from betazero_amd.engine import SelfPlayEngine
from betazero_amd.net import DeviceNet, PolicyValueNet
from betazero_amd import _lib
assert b"TT_WEAK_HASH" in _lib.lib().bz_build_info(), _lib.lib().bz_build_info()
torch.manual_seed(0)
dn = DeviceNet.from_module(PolicyValueNet(128, 6, 64).round_to_bf16_(), 64)
kw = dict(net=dn, temp_moves=8, openings=1, seed=3, rounds=2, stagger=50)
on = SelfPlayEngine("reversi", 64, 300, "net_bf16", eval_cache=True, **kw)
off = SelfPlayEngine("reversi", 64, 300, "net_bf16", eval_cache=False, **kw)
for e in (on, off):
    e.reset_games(); e.reset_counters()
for mv in range(3):
    for e in (on, off):
        e.search()
    (N1, W1, P1), (N0, W0, P0) = on.root_stats(), off.root_stats()
    assert np.array_equal(N1, N0) and np.array_equal(W1.view(np.uint32), W0.view(np.uint32)) and np.array_equal(P1.view(np.uint32), P0.view(np.uint32)), mv
    for e in (on, off):
        e.play(True)
assert on.status() == off.status()
a, b = on.example_tensors(), off.example_tensors()
for f in ("own", "opp", "act", "mover", "len"):
    assert torch.equal(a[f], b[f]), f
assert torch.equal(a["pi"].view(torch.int32), b["pi"].view(torch.int32))
c1, c0 = on.counters(), off.counters()
assert c1["n_net_leaves"] + c1["n_cache_hits"] == c0["n_net_leaves"]
print("hits", c1["n_cache_hits"], "of", c0["n_net_leaves"])
""" % ROOT
    r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, BZ_HIP_SO=so, BZ_ALLOW_EXPERIMENT="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.startswith("hits "), r.stdout


def test_evaluation_cache_generation_stamp_wraps_cleanly():
    """the cache stamps its entries with the search count modulo 2^19 - 2 (19 bits beside the 13-bit node id).  An engine
    started three searches below the wrap (test hook bz_engine_debug_set_search_seq) plays six moves across it: same root
    statistics as an engine without the cache after every search, and carried-over hits in every search but the first --
    the search with stamp 1 takes the tree of stamp 2^19 - 2 as its previous one."""
    from betazero_amd import _lib
    from betazero_amd.net import DeviceNet
    B, sims = 32, 200
    dn = DeviceNet.from_module(_net(128, 6, bf16=True), B)
    kw = dict(net=dn, temp_moves=8, openings=1, seed=6, rounds=2, stagger=40)
    on, off = _engine("reversi", B, sims, "net_bf16", eval_cache=True, **kw), _engine("reversi", B, sims, "net_bf16", eval_cache=False, **kw)
    _lib.check(_lib.lib().bz_engine_debug_set_search_seq(on.h, (1 << 19) - 2 - 3))
    for e in (on, off):
        e.reset_games(); e.reset_counters()
    prev_hits = []
    for mv in range(6):
        before = on.counters()["n_cache_hits_prev"]
        for e in (on, off):
            e.search()
        prev_hits.append(on.counters()["n_cache_hits_prev"] - before)
        (N1, W1, P1), (N0, W0, P0) = on.root_stats(), off.root_stats()
        assert np.array_equal(N1, N0), mv
        assert np.array_equal(W1.view(np.uint32), W0.view(np.uint32)) and np.array_equal(P1.view(np.uint32), P0.view(np.uint32)), mv
        for e in (on, off):
            e.play(True)
    assert on.status() == off.status()
    assert prev_hits[0] == 0 and all(h > 0 for h in prev_hits[1:]), prev_hits


def test_evaluation_cache_carries_nothing_across_a_weight_update():
    """the carried-over part of the evaluation cache holds evaluations of the PREVIOUS search's weights: after
    DeviceNet.update (bz_net_update: refresh_device_net between two iterations on a live engine) the next search must not
    take them.  Two engines on one net (cache with carry-over / no cache): search, move, new weights, search, move, search --
    root statistics bit for bit the same after every search; the search right after the update has no carried-over hit,
    the one after that has them again."""
    from betazero_amd.net import DeviceNet
    B, sims = 32, 200
    m1, m2 = _net(128, 6, bf16=True), _net(128, 6, seed=5, bf16=True)
    dn = DeviceNet.from_module(m1, B)
    kw = dict(net=dn, temp_moves=8, openings=1, seed=4, rounds=2, stagger=40)
    on, off = _engine("reversi", B, sims, "net_bf16", eval_cache=True, **kw), _engine("reversi", B, sims, "net_bf16", eval_cache=False, **kw)
    for e in (on, off):
        e.reset_games(); e.reset_counters()
    prev_hits = []
    for mv in range(4):
        if mv == 2:
            torch.cuda.synchronize()
            dn.update(m2.flat_params())
        before = on.counters()["n_cache_hits_prev"]
        for e in (on, off):
            e.search()
        prev_hits.append(on.counters()["n_cache_hits_prev"] - before)
        (N1, W1, P1), (N0, W0, P0) = on.root_stats(), off.root_stats()
        assert np.array_equal(N1, N0), mv
        assert np.array_equal(W1.view(np.uint32), W0.view(np.uint32)) and np.array_equal(P1.view(np.uint32), P0.view(np.uint32)), mv
        for e in (on, off):
            e.play(True)
    assert on.status() == off.status()
    assert prev_hits[0] == 0 and prev_hits[1] > 0 and prev_hits[2] == 0 and prev_hits[3] > 0, prev_hits


def test_cfg3_800_sims_bf16_net_search_close_to_oracle_bf16_emulation():
    """one 800-simulation search with the bf16 MFMA net in the loop vs the oracle running the same
    search with its bf16-emulating net (CPU threads, 8 games).  The two nets differ by ~6e-4 on the logits
    (accumulation order), so the trees may part ways late; the visit distributions must stay close:
    total-variation distance of pi < 0.01 and the same most-visited move (measured on MI355X: identical
    visit counts in all 8 games, TV = 0)."""
    import threading
    from betazero_amd.net import DeviceNet
    m = _net(128, 6, bf16=True)
    B, sims = 8, 800
    dn = DeviceNet.from_module(m, B)
    eng = _engine("reversi", B, sims, "net_bf16", net=dn, temp_moves=8, openings=1, seed=0)
    eng.reset_games()
    own, opp, tm, _ = eng.positions()
    eng.search()
    N, _, P = eng.root_stats()
    eng.status()
    on = orc.Net(128, 6, 64, m.flat_params())
    res = [None] * B

    def work(g):
        res[g] = orc.mcts_search(orc.GAME_REVERSI, int(own[g]), int(opp[g]), int(tm[g]), sims, orc.EVAL_NET_BF16, net=on)
    th = [threading.Thread(target=work, args=(g,)) for g in range(B)]
    [t.start() for t in th]
    [t.join() for t in th]
    tv, same = [], 0
    for g in range(B):
        n, _, p, _ = res[g]
        assert np.abs(P[g] - p).max() < 2e-3  # root priors: softmax of logits that agree to ~1e-3
        tv.append(0.5 * np.abs(N[g].astype(np.float64) - n.astype(np.float64)).sum() / sims)
        same += int(np.argmax(N[g]) == np.argmax(n))
    print("bf16 search vs oracle emulation: TV distance of pi per game", np.round(tv, 4), "same argmax", same, "of", B)
    assert max(tv) < 0.01 and same == B


def test_cfg5_800_sims_fp8_net_search_close_to_oracle_fp8_emulation():
    """BASELINE cfg 5 as SURVEY 7 reads it -- the fp8 net as the evaluator INSIDE the search: one 800-simulation search
    of 8 games with the fp8 MFMA tower in the loop vs the oracle running the same search with its fp8-emulating net.
    The two nets differ by ~1e-2 on the logits (accumulation order over e4m3 products), so the trees may part ways
    sooner than with bf16; the visit distributions must stay close: total-variation distance of pi < 0.02 and the same
    most-visited move."""
    import threading
    from betazero_amd.net import DeviceNet
    from betazero_amd.quant import fake_quantize_fp8_
    m = fake_quantize_fp8_(_net(128, 6))
    B, sims = 8, 800
    dn = DeviceNet.from_module(m, B)
    eng = _engine("reversi", B, sims, "net_fp8", net=dn, temp_moves=8, openings=1, seed=0)
    eng.reset_games()
    own, opp, tm, _ = eng.positions()
    eng.search()
    N, _, P = eng.root_stats()
    eng.status()
    on = orc.Net(128, 6, 64, m.flat_params())
    res = [None] * B

    def work(g):
        res[g] = orc.mcts_search(orc.GAME_REVERSI, int(own[g]), int(opp[g]), int(tm[g]), sims, orc.EVAL_NET_FP8, net=on)
    th = [threading.Thread(target=work, args=(g,)) for g in range(B)]
    [t.start() for t in th]
    [t.join() for t in th]
    tv, same = [], 0
    for g in range(B):
        n, _, p, _ = res[g]
        assert N[g].sum() == sims and np.abs(P[g] - p).max() < 1e-2  # root priors: softmax of logits that agree to ~1e-2
        tv.append(0.5 * np.abs(N[g].astype(np.float64) - n.astype(np.float64)).sum() / sims)
        same += int(np.argmax(N[g]) == np.argmax(n))
    print("fp8 search vs oracle emulation: TV distance of pi per game", np.round(tv, 4), "same argmax", same, "of", B)
    assert max(tv) < 0.02 and same == B


def _stage_isolating_params(m, stage):
    """copy of module m in which conv stage `stage` (0 = stem, 2k+1 / 2k+2 = conv1 / conv2 of block k) is the
    LAST one that changes the activations: later blocks are zeroed (a zero block is the identity on the
    non-negative residual stream); to expose conv1 of block k, its conv2 becomes the identity kernel."""
    import copy
    mm = copy.deepcopy(m)
    with torch.no_grad():
        for k in range(mm.NB):
            s1, s2 = 2 * k + 1, 2 * k + 2
            if s1 > stage:       # whole block after the stage: zero = identity block
                for c in (mm.c1[k], mm.c2[k]):
                    c.weight.zero_(); c.bias.zero_()
            elif s2 > stage:     # stage is this block's conv1: conv2 := identity (centre tap, exact in bf16)
                mm.c2[k].weight.zero_(); mm.c2[k].bias.zero_()
                for c in range(mm.C):
                    mm.c2[k].weight[c, c, 1, 1] = 1.0
    return mm


def test_net_bf16_every_conv_stage_on_its_own_vs_oracle():
    """13 nets, one per conv stage (stem + 12 conv3x3): everything after the stage is an identity, so the
    heads see that stage's output directly and a dropped tap / mis-swizzled chunk in ONE layer cannot hide
    behind the later layers.  Tolerance 2e-3 on logits / 1e-3 on value (about 3x the whole-net error)."""
    from betazero_amd.net import DeviceNet
    m = _net(128, 6, bf16=True)
    n = 37
    own, opp = _positions(n, seed=3)
    dn = DeviceNet.from_module(m, 64)
    worst = (0.0, 0.0)
    outs = []
    for stage in range(13):
        mm = _stage_isolating_params(m, stage)
        dn.update(mm.flat_params())
        lg, v = dn.forward(_dev_u64(own), _dev_u64(opp), bf16=True)
        olg, ov = orc.Net(128, 6, 64, mm.flat_params()).forward(own, opp, bf16=True)
        el, ev = np.abs(lg.cpu().numpy() - olg).max(), np.abs(v.cpu().numpy() - ov).max()
        worst = (max(worst[0], el), max(worst[1], ev))
        assert el < 2e-3 and ev < 1e-3, (stage, el, ev)
        outs.append(olg)
    for a, b in zip(outs[:-1], outs[1:]):  # the construction really isolates stages: consecutive outputs differ
        assert np.abs(a - b).max() > 1e-3
    print("per-stage bf16 net: worst |dlogit| %.2e  |dv| %.2e" % worst)


@pytest.mark.parametrize("n,stride", [(301, 8), (1848, 41)], ids=["301_positions", "bench_launch_1848_positions"])
def test_net_bf16_every_conv_stage_on_its_own_throughput_shape_vs_oracle(n, stride):
    """the same 13 stage-isolating nets through the THROUGHPUT shape of the fused kernel (batches > 256 positions:
    Tw<128, 4, true>, four positions per workgroup, row-tile units that skip the padding-row MFMAs, the K-loop on
    v_mfma_f32_16x16x32_bf16 since round 4 -- the headline kernel): 301 positions (a ragged last workgroup) and 1848 (what
    one launch of bench.py's two pipelines evaluates: 462 workgroups, nearly two rounds of the chip), a strided sample of
    rows and the last rows compared with the oracle per stage, so the row-tile skip logic and the new lane map are pinned
    layer by layer and not only through the whole net."""
    from betazero_amd.net import DeviceNet
    m = _net(128, 6, bf16=True)
    own, opp = _positions(n, seed=13)
    pick = np.unique(np.concatenate([np.arange(0, n, stride), np.arange(n - 5, n)]))
    dn = DeviceNet.from_module(m, 2048)
    worst = (0.0, 0.0)
    for stage in range(13):
        mm = _stage_isolating_params(m, stage)
        dn.update(mm.flat_params())
        lg, v = dn.forward(_dev_u64(own), _dev_u64(opp), bf16=True)
        olg, ov = orc.Net(128, 6, 64, mm.flat_params()).forward(own[pick], opp[pick], bf16=True)
        el, ev = np.abs(lg.cpu().numpy()[pick] - olg).max(), np.abs(v.cpu().numpy()[pick] - ov).max()
        worst = (max(worst[0], el), max(worst[1], ev))
        assert el < 2e-3 and ev < 1e-3, (stage, el, ev)
        # and the latency shape (<= 256 positions) gives the same bits for the same positions
        l2, v2 = dn.forward(_dev_u64(own[:64]), _dev_u64(opp[:64]), bf16=True)
        assert np.array_equal(l2.cpu().numpy().view(np.uint32), lg.cpu().numpy()[:64].view(np.uint32)), stage
    print("per-stage bf16 net, throughput shape: worst |dlogit| %.2e  |dv| %.2e" % worst)


@pytest.mark.parametrize("prec", ["bf16", "fp8"])
def test_cfg5_batch_8192_net_parity_on_sampled_rows(prec):
    """BASELINE cfg 5 size: one forward over 8192 distinct positions (2048 workgroups = two per CU for the fp8
    kernel, eight waves of workgroups for bf16).  Every row must equal, bit for bit, the same position
    evaluated in small ragged batches (rows are independent of their neighbours), and a strided sample of
    rows is compared with the oracle's emulation at the tolerances of the small-batch tests."""
    from betazero_amd.net import DeviceNet
    from betazero_amd.quant import fake_quantize_fp8_
    fp8 = prec == "fp8"
    m = _net(128, 6, bf16=True)
    if fp8:
        m = fake_quantize_fp8_(_net(128, 6))
    d = np.load(os.path.join(G, "reversi_random_games.npz"))
    rows = d["rows"][d["rows"][:, 1] == 8]
    assert len(rows) >= 8192
    idx = np.random.default_rng(11).choice(len(rows), 8192, replace=False)
    own, opp = rows[idx, 4].copy(), rows[idx, 5].copy()
    dn = DeviceNet.from_module(m, 8192)
    lg, v = dn.forward(_dev_u64(own), _dev_u64(opp), bf16=True, fp8=fp8)
    lg, v = lg.cpu().numpy(), v.cpu().numpy()
    assert np.isfinite(lg).all() and np.isfinite(v).all() and (np.abs(v) <= 1).all()
    for lo in range(0, 8192, 2048):  # chunks of 41 positions (ragged workgroup tiles) over a quarter of the rows
        for s in range(lo, lo + 41 * 5, 41):
            l2, v2 = dn.forward(_dev_u64(own[s:s + 41]), _dev_u64(opp[s:s + 41]), bf16=True, fp8=fp8)
            assert np.array_equal(l2.cpu().numpy().view(np.uint32), lg[s:s + 41].view(np.uint32))
            assert np.array_equal(v2.cpu().numpy().view(np.uint32), v[s:s + 41].view(np.uint32))
    sample = np.arange(5, 8192, 89)  # 92 rows spread over all workgroups
    olg, ov = orc.Net(128, 6, 64, m.flat_params()).forward(own[sample], opp[sample], bf16=2 if fp8 else 1)
    el, ev = np.abs(lg[sample] - olg).max(), np.abs(v[sample] - ov).max()
    print(f"batch-8192 {prec} net vs oracle on {len(sample)} sampled rows: |dlogit| {el:.2e} |dv| {ev:.2e}")
    assert (el < 0.02 and ev < 0.01) if fp8 else (el < 2e-3 and ev < 1e-3)


def test_reversi_score_batch_counts_vs_oracle():
    """a6 batched: winner and counts[B][2] of get_score (reversi_board.py:67-85) on the fixture's final boards"""
    d = np.load(os.path.join(G, "reversi_random_games.npz"))
    fin = d["finals"]  # game, size, winner+1, n_plus, n_minus, passes, x_final, o_final
    x, o = fin[:, 6].copy(), fin[:, 7].copy()
    n = len(fin)
    w = torch.empty(n, dtype=torch.int8, device=DEV)
    c = torch.empty((n, 2), dtype=torch.uint8, device=DEV)
    xd, od = _dev_u64(x), _dev_u64(o)  # keep the device copies alive across the asynchronous call
    _lib.check(_lib.lib().bz_reversi_score_batch(xd.data_ptr(), od.data_ptr(), n, w.data_ptr(), c.data_ptr(), _stream()))
    torch.cuda.synchronize()
    assert np.array_equal(w.cpu().numpy().astype(np.int64), fin[:, 2].astype(np.int64) - 1)
    assert np.array_equal(c.cpu().numpy().astype(np.int64), fin[:, 3:5].astype(np.int64))
    for i in range(0, n, 17):
        ws, (nx, no) = orc.reversi_score(int(x[i]), int(o[i]))
        assert (ws, nx, no) == (int(w[i]), int(c[i, 0]), int(c[i, 1]))


@pytest.mark.parametrize("gw", [1, 2, 4, 8])
def test_ttt_specialised_fused_search_equals_generic_kernel_and_oracle(gw):
    """the TTT-specialised fused search (root edges in registers, child header packed into the edge word, no
    node loads below the root; lanes per game = cfg.ttt_lanes = 1, 2, 4 or 8) against the generic fused kernel
    (ttt_lanes = -1) and the oracle: root N/W/P, every work counter and complete games, bit for bit, from random
    reachable positions"""
    d = np.load(os.path.join(G, "ttt_exhaustive.npz"))
    pos = d["pos"]
    live = pos[pos[:, 4] == 0]  # not game-over
    rng = np.random.default_rng(gw)
    sel = live[rng.choice(len(live), 500, replace=False)]
    tm = np.where(sel[:, 2] == 1, 1, -1).astype(np.int8)
    own = np.where(tm == 1, sel[:, 0], sel[:, 1]).astype(np.uint64)
    opp = np.where(tm == 1, sel[:, 1], sel[:, 0]).astype(np.uint64)
    out = {}
    for mode in (gw, -1):
        for ev, sims in (("hash", 120), ("uniform", 50)):
            eng = _engine("ttt", len(sel), sims, ev, ttt_lanes=mode)
            eng.set_roots(own, opp, tm)
            eng.reset_counters()
            eng.search()
            N, W, P = eng.root_stats()
            eng.status()
            out[(mode, ev)] = (N, W.view(np.uint32), P.view(np.uint32), eng.counters())
    for ev, sims in (("hash", 120), ("uniform", 50)):
        a, b = out[(gw, ev)], out[(-1, ev)]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert a[3] == b[3], (a[3], b[3])
        for g in range(0, len(sel), 25):
            n, w, p, _ = orc.mcts_search(orc.GAME_TTT, int(own[g]), int(opp[g]), int(tm[g]), sims,
                                         orc.EVAL_HASH if ev == "hash" else orc.EVAL_UNIFORM)
            assert np.array_equal(a[0][g], n) and np.array_equal(a[1][g], w.view(np.uint32))
    _check_selfplay("ttt", 96, 40, "hash", 4, 0, 7, base=300, ttt_lanes=gw)


@pytest.mark.parametrize("C,NB", [(64, 3), (256, 2)])
def test_net_bf16_mfma_other_widths_vs_oracle_bf16_emulation(C, NB):
    """the fused bf16 MFMA kernel at 64 channels (8 positions per workgroup, 128-byte cells, 3-bit swizzle) and at
    256 channels (2 positions per workgroup, two M-tiles per wave, half-tap weight chunks) vs the oracle's bf16
    emulation, at ragged batch sizes (not multiples of the workgroup's positions), plus self-play with the net in the
    loop (legal, terminates) and equality of every row with the same position evaluated in another batch."""
    from betazero_amd.net import DeviceNet
    m = _net(C, NB, seed=7, bf16=True)
    on = orc.Net(C, NB, 64, m.flat_params())
    dn = DeviceNet.from_module(m, 320)
    worst = (0.0, 0.0)
    for n in (1, 7, 37, 291):  # 291 > 256: the throughput shape (row-tile units at C = 64), ragged last workgroup
        own, opp = _positions(n, seed=20 + n)
        lg, v = dn.forward(_dev_u64(own), _dev_u64(opp), bf16=True)
        olg, ov = on.forward(own, opp, bf16=True)
        el, ev = np.abs(lg.cpu().numpy() - olg).max(), np.abs(v.cpu().numpy() - ov).max()
        worst = (max(worst[0], el), max(worst[1], ev))
        assert el < 2e-3 and ev < 1e-3, (C, n, el, ev)
    print(f"bf16 MFMA net C={C}: worst |dlogit| {worst[0]:.2e} |dv| {worst[1]:.2e}; logit std {olg.std():.3f}")
    own, opp = _positions(37, seed=57)
    a = dn.forward(_dev_u64(own), _dev_u64(opp), bf16=True)[0].cpu().numpy()
    b = dn.forward(_dev_u64(own[5:30]), _dev_u64(opp[5:30]), bf16=True)[0].cpu().numpy()
    assert np.array_equal(a[5:30].view(np.uint32), b.view(np.uint32))  # rows do not depend on their neighbours
    own, opp = _positions(291, seed=58)   # throughput shape vs latency shape: the same bits for every row
    big = dn.forward(_dev_u64(own), _dev_u64(opp), bf16=True)
    small = dn.forward(_dev_u64(own[100:140]), _dev_u64(opp[100:140]), bf16=True)
    assert np.array_equal(big[0].cpu().numpy()[100:140].view(np.uint32), small[0].cpu().numpy().view(np.uint32))
    assert np.array_equal(big[1].cpu().numpy()[100:140].view(np.uint32), small[1].cpu().numpy().view(np.uint32))
    eng = _engine("reversi", 12, 6, "net_bf16", net=dn, temp_moves=8, openings=1)
    eng.run_iteration()
    ex = eng.examples()
    assert (eng.winners()[1][0] > 20).all() and _legal_per_oracle(ex.own, ex.opp, ex.act)


# ---------------------------------------------------------------- opt-in search features
@pytest.mark.parametrize("gw", [2, 4, -1])
def test_dirichlet_root_noise_search_and_selfplay_vs_oracle_bitexact(gw):
    """Dirichlet noise on the root priors (float32 log / Gamma sampler spec'd to the bit): root
    N / W / P of noisy searches and complete noisy self-play games equal the oracle's, for the TTT-specialised
    fused kernel (2, 4 lanes), the generic fused kernel (ttt_lanes = -1, and Reversi) and the step kernels (f32 net)."""
    from betazero_amd.net import DeviceNet
    d = np.load(os.path.join(G, "ttt_exhaustive.npz"))
    live = d["pos"][d["pos"][:, 4] == 0]
    sel = live[np.random.default_rng(1).choice(len(live), 120, replace=False)]
    tm = np.where(sel[:, 2] == 1, 1, -1).astype(np.int8)
    own = np.where(tm == 1, sel[:, 0], sel[:, 1]).astype(np.uint64)
    opp = np.where(tm == 1, sel[:, 1], sel[:, 0]).astype(np.uint64)
    for alpha in (0.3, 1.0):
        eng = _engine("ttt", len(sel), 60, "hash", seed=5, game_id_base=1000, dirichlet_alpha=alpha, dirichlet_eps=0.25,
                      ttt_lanes=gw)
        eng.set_roots(own, opp, tm)
        eng.search()
        N, W, P = eng.root_stats()
        eng.status()
        for g in range(len(sel)):
            n, w, p, _ = orc.mcts_search(orc.GAME_TTT, int(own[g]), int(opp[g]), int(tm[g]), 60, orc.EVAL_HASH,
                                         dir_alpha=alpha, dir_eps=0.25, seed=5, gid=1000 + g, ply=0)
            assert np.array_equal(N[g], n) and np.array_equal(W[g].view(np.uint32), w.view(np.uint32)), (alpha, g)
            assert np.array_equal(P[g].view(np.uint32), p.view(np.uint32)), (alpha, g)
    if gw != 4:
        return
    # Reversi: generic fused kernel (hash) and the step kernels (f32 net), whole games with a fresh draw per move
    for ev, oev, net, onet, n_games, sims in (("hash", orc.EVAL_HASH, None, None, 24, 40),):
        eng = _engine("reversi", n_games, sims, ev, temp_moves=6, openings=1, seed=3, game_id_base=50,
                      dirichlet_alpha=0.5, dirichlet_eps=0.25)
        eng.run_iteration()
        ex = eng.examples()
        for g in range(n_games):
            r = orc.selfplay_game(orc.GAME_REVERSI, 50 + g, sims, oev, 6, 1, 3, dir_alpha=0.5, dir_eps=0.25)
            m = ex.game == 50 + g
            assert np.array_equal(ex.act[m], r["act"]) and np.array_equal(ex.pi[m].view(np.uint32), r["pi"].view(np.uint32))
    m = _net(32, 2, seed=4)
    dn, on = DeviceNet.from_module(m, 8), orc.Net(32, 2, 64, m.flat_params())
    eng = _engine("reversi", 6, 10, "net_f32", net=dn, temp_moves=4, openings=1, seed=8, dirichlet_alpha=0.3, dirichlet_eps=0.5)
    eng.run_iteration()
    ex = eng.examples()
    for g in range(6):
        r = orc.selfplay_game(orc.GAME_REVERSI, g, 10, orc.EVAL_NET_F32, 4, 1, 8, net=on, dir_alpha=0.3, dir_eps=0.5)
        mk = ex.game == g
        assert np.array_equal(ex.act[mk], r["act"]) and np.array_equal(ex.pi[mk].view(np.uint32), r["pi"].view(np.uint32))


def test_external_evaluator_with_root_noise_vs_oracle():
    """Dirichlet root noise used to be drawn only inside bz_engine_search, so step-API callers
    (search_external, BZ_EVAL_EXTERNAL) silently searched on clean priors although the config asked for noise.
    bz_engine_root_noise is now an entry point and search_external calls it where bz_engine_search does: root N / W / P of
    a noisy search driven through the step API with caller-filled (hash) logits equal the oracle's noisy search."""
    own, opp = _positions(6, seed=77)
    tm = np.array([1, -1, 1, -1, 1, -1], np.int8)
    sims = 48
    eng = _engine("reversi", 6, sims, "external", seed=9, game_id_base=200, dirichlet_alpha=0.4, dirichlet_eps=0.25)
    eng.set_roots(own, opp, tm)

    def hash_eval(o, p, kind):
        torch.cuda.synchronize()
        oc, pc = o.cpu().numpy().view(np.uint64), p.cpu().numpy().view(np.uint64)
        lg, v = np.zeros((6, eng.na), np.float32), np.zeros(6, np.float32)
        for g in range(6):
            lg[g], v[g] = orc.eval_hash(int(oc[g]), int(pc[g]), eng.na)
        return torch.from_numpy(lg).cuda(), torch.from_numpy(v).cuda()
    eng.search_external(hash_eval)
    N, W, P = eng.root_stats()
    eng.status()
    clean = 0
    for g in range(6):
        n, w, p, _ = orc.mcts_search(orc.GAME_REVERSI, int(own[g]), int(opp[g]), int(tm[g]), sims, orc.EVAL_HASH,
                                     dir_alpha=0.4, dir_eps=0.25, seed=9, gid=200 + g, ply=0)
        assert np.array_equal(N[g], n) and np.array_equal(W[g].view(np.uint32), w.view(np.uint32)), g
        assert np.array_equal(P[g].view(np.uint32), p.view(np.uint32)), g
        _, _, p0, _ = orc.mcts_search(orc.GAME_REVERSI, int(own[g]), int(opp[g]), int(tm[g]), sims, orc.EVAL_HASH)
        clean += int(np.array_equal(p, p0))
    assert clean <= 1  # the noise really moved the priors (a root with a single move keeps P = 1)


def test_non_finite_evaluator_output_raises_instead_of_walking_garbage():
    """a NaN / infinite logit or value makes every PUCT comparison false; the search would quietly pick edge 0 with a
    zeroed header, overwrite root edges and report visits on action 0 (seen once with a diverged net: the arena then
    played an illegal move).  The expansion now raises an engine error flag, which status() turns into an exception."""
    own, opp = _positions(4, seed=5)
    tm = np.ones(4, np.int8)
    for what in ("nan_logit", "inf_logit", "nan_value"):
        eng = _engine("reversi", 4, 8, "external")
        eng.set_roots(own, opp, tm)

        def bad(o, p, kind, what=what):
            lg = torch.zeros((4, eng.na), device="cuda:0")
            v = torch.zeros(4, device="cuda:0")
            if what == "nan_logit":
                lg[2, :] = float("nan")
            elif what == "inf_logit":
                lg[1, :] = float("inf")
            else:
                v[3] = float("nan")
            return lg, v
        eng.search_external(bad)
        with pytest.raises(RuntimeError, match="non-finite"):
            eng.status()
    eng = _engine("reversi", 4, 8, "external")   # and a clean evaluator raises nothing
    eng.set_roots(own, opp, tm)
    eng.search_external(lambda o, p, k: (torch.zeros((4, eng.na), device="cuda:0"), torch.zeros(4, device="cuda:0")))
    eng.status()


def test_arena_both_players_only_ever_play_legal_moves_stress():
    """4096 concurrent arena games from seeded random openings, MCTS (hash / uniform evaluators) against the minimax
    kernel: every move of either side goes through the batched env step (legality from the carry-propagation flips),
    which must never flag one -- three rule implementations (tree kernels, minimax kernel, env step) agreeing on
    positions none of the fixtures holds."""
    from betazero_amd.arena import play_arena
    for seed, ev, depth, sims in ((0, "uniform", 2, 16), (1, "hash", 3, 24)):
        r = play_arena("reversi", 2048, sims, opponent_depth=depth, evaluator=ev, seed=seed, opening_plies=6 + seed)
        s = r.summary()
        assert s["games"] == 2048 and s["wins"] + s["draws"] + s["losses"] == 2048


def test_subtree_reuse_at_the_packed_edge_limit_vs_oracle_bitexact():
    """sims = BZ_ENGINE_MAX_SIMS_REUSE = 2045 with subtree reuse: the arena holds 4 x 2047 = 8188 nodes, kept subtrees of
    up to ~2000 nodes are copied and renumbered (child ids and first-edge offsets rewritten inside the packed edge words),
    visit counts of kept roots exceed the fresh-search range.  The first three moves of four games, rows and counters
    against the oracle."""
    sims, n, moves = 2045, 4, 3
    eng = _engine("reversi", n, sims, "hash", temp_moves=8, openings=1, seed=5, reuse_subtree=True)
    eng.reset_games()
    eng.reset_counters()
    for _ in range(moves):
        eng.search()
        eng.play(False)
    eng.status()
    rows = _ex_rows(eng, moves)
    exp = {}
    for g in range(n):
        r = orc.selfplay_game(orc.GAME_REVERSI, g, sims, orc.EVAL_HASH, 8, 1, 5, reuse=True, max_moves=moves)
        assert np.array_equal(rows["own"][g].view(np.uint64), r["own"]) and np.array_equal(rows["act"][g], r["act"]), g
        assert np.array_equal(rows["pi"][g].view(np.uint32), r["pi"].view(np.uint32)), g
        for k, v in r["counters"].items():
            exp[k] = exp.get(k, 0) + v
    cnt = eng.counters()
    for k in ("n_sims", "n_path_nodes", "n_child_scored", "n_edges_backed", "n_expanded", "n_child_written", "n_env_steps"):
        assert cnt[k] == exp[k], (k, cnt[k], exp[k])
    assert cnt["n_sims"] == n * moves * sims


def test_subtree_reuse_selfplay_vs_oracle_bitexact():
    """subtree reuse: the chosen child's subtree is copied to the front of the other arena and searched on
    (`sims` new simulations on top of the retained statistics), through passes, with the arena-capacity rule, with and
    without root noise, over two rounds of games.  Whole games and the work counters equal the oracle's."""
    from betazero_amd.net import DeviceNet
    cases = (("ttt", orc.GAME_TTT, 48, 30, 3, 0, 0.0, 0.0), ("reversi", orc.GAME_REVERSI, 32, 40, 6, 1, 0.0, 0.0),
             ("reversi6", orc.GAME_REVERSI6, 24, 25, 4, 0, 0.5, 0.25), ("reversi4", orc.GAME_REVERSI4, 24, 40, 4, 0, 0.3, 0.25))
    total_pass = 0
    for game, og, n, sims, tmv, openings, alpha, eps in cases:
        eng = _engine(game, n, sims, "hash", temp_moves=tmv, openings=openings, seed=7, game_id_base=10,
                      dirichlet_alpha=alpha, dirichlet_eps=eps, reuse_subtree=True)
        eng.reset_counters()
        eng.run_iteration()
        ex = eng.examples()
        winners, lens = eng.winners()
        cnt = eng.counters()
        exp = {k: 0 for k in cnt}
        for g in range(n):
            r = orc.selfplay_game(og, 10 + g, sims, orc.EVAL_HASH, tmv, openings, 7, reuse=True, dir_alpha=alpha, dir_eps=eps)
            m = ex.game == 10 + g
            assert lens[0, g] == len(r["own"]) and winners[0, g] == r["winner"], (game, g)
            assert np.array_equal(ex.act[m], r["act"]) and np.array_equal(ex.own[m], r["own"]), (game, g)
            assert np.array_equal(ex.pi[m].view(np.uint32), r["pi"].view(np.uint32)), (game, g)
            for k in exp:
                exp[k] += r["counters"][k]
            total_pass += r["passes"]
        for k in ("n_sims", "n_path_nodes", "n_child_scored", "n_edges_backed", "n_expanded", "n_child_written", "n_env_steps"):
            assert cnt[k] == exp[k], (game, k, cnt[k], exp[k])
    assert total_pass > 0
    # f32 net in the loop (step kernels with packed leaves), two rounds with restart
    m = _net(32, 2, seed=4)
    dn, on = DeviceNet.from_module(m, 8), orc.Net(32, 2, 64, m.flat_params())
    eng = _engine("reversi", 6, 12, "net_f32", net=dn, temp_moves=8, openings=1, seed=2, reuse_subtree=True)
    eng.run_iteration()
    ex = eng.examples()
    for g in range(6):
        r = orc.selfplay_game(orc.GAME_REVERSI, g, 12, orc.EVAL_NET_F32, 8, 1, 2, net=on, reuse=True)
        mk = ex.game == g
        assert np.array_equal(ex.act[mk], r["act"]) and np.array_equal(ex.pi[mk].view(np.uint32), r["pi"].view(np.uint32))
    # bf16 MFMA net, 800 simulations on top of the kept subtree: capacity rule, legality, visit sums >= sims
    dn = DeviceNet.from_module(_net(128, 6, bf16=True), 64)
    eng = _engine("reversi", 64, 200, "net_bf16", net=dn, temp_moves=8, openings=1, reuse_subtree=True, rounds=2)
    eng.reset_games()
    for mv in range(6):
        own, opp, tm_, st = eng.positions()
        eng.search()
        N, _, _ = eng.root_stats()
        eng.status()
        assert (N.sum(1) >= 200).all() and (mv == 0 or (N.sum(1) > 200).any())  # retained visits show up from move 2 on
        eng.play(True)
    t = eng.example_tensors()
    assert _legal_per_oracle(t["own"][0, :, :6].cpu().numpy().view(np.uint64).ravel(), t["opp"][0, :, :6].cpu().numpy().view(np.uint64).ravel(),
                             t["act"][0, :, :6].cpu().numpy().ravel())


def test_search_features_match_the_twin_over_reference_boards_fixture():
    """fixture F10 (twin over the REFERENCE's boards, root noise / subtree reuse on): the engine's games equal it"""
    d = np.load(os.path.join(G, "mcts_twin_features.npz"))
    for c in json.load(open(os.path.join(G, "mcts_twin_features.json")))["cases"]:
        eng = _engine(c["game"], 1, c["sims"], c["eval"], temp_moves=c["temp_moves"], openings=c["openings"], seed=c["seed"],
                      game_id_base=c["gid"], dirichlet_alpha=c["alpha"], dirichlet_eps=c["eps"], reuse_subtree=c["reuse"])
        eng.run_iteration()
        ex = eng.examples()
        i = c["id"]
        assert np.array_equal(ex.own, d[f"g{i}_own"]) and np.array_equal(ex.act, d[f"g{i}_act"]), c
        assert np.array_equal(ex.pi.view(np.uint32), d[f"g{i}_pi"].view(np.uint32)), c
        assert eng.winners()[0][0, 0] == c["winner"]


def test_net_player_follows_the_reference_aiplayer_rule():
    """NetPlayer = AIPlayer.get_move (players.py:84-98) on the conv net: canonical input for the side to move, best LEGAL
    move in descending-logit order.  With the exact-fp32 path the choice must be the oracle's, for both colours,
    over a whole game against a random opponent driven by the reference-style loop."""
    import random
    import betazero_amd as bz
    from betazero_amd.net import DeviceNet
    m = _net(32, 2, seed=9)
    dn, on = DeviceNet.from_module(m, 4), orc.Net(32, 2, 64, m.flat_params())
    random.seed(4)
    for sym in (1, -1):
        pl = bz.NetPlayer(sym, dn, precision="f32")
        other = bz.ReversiRandomPlayer(-sym)
        g = bz.ReversiHeadless(pl if sym == 1 else other, other if sym == 1 else pl)
        b, cur, n_checked = bz.ReversiBoard(), 1, 0
        while not b.is_game_over():
            mv = b.generate_possible_moves(cur)
            if mv:
                if cur == sym:
                    r, c = pl.get_move(b)
                    own, opp = b.bits(sym)
                    olg, _ = on.forward(np.array([own], np.uint64), np.array([opp], np.uint64))
                    assert np.array_equal(pl.last_logits.view(np.uint32), olg[0, :64].view(np.uint32))
                    idx = [8 * rr + cc for rr, cc in mv]
                    assert 8 * r + c == idx[int(np.argmax(olg[0][idx]))]
                    n_checked += 1
                else:
                    r, c = other.get_move(b)
                b = b.make_move(r, c, cur)
            cur = -cur
        assert n_checked > 20
        del g
    pl = bz.NetPlayer(1, DeviceNet.from_module(_net(128, 6, bf16=True), 4))  # the bf16 MFMA path (latency shape, B = 1)
    assert pl.get_move(bz.ReversiBoard()) in bz.ReversiBoard().generate_possible_moves(1)


def test_gather_examples_over_rccl_world_size_1():
    """the one collective of the multi-GPU path on the real backend ("nccl" = RCCL), as far as one GPU allows: a
    world-size-1 process group, ONE all_gather_into_tensor of two engines' example blocks straight from the workspaces,
    and the device-side unpack -- the pooled examples equal the engines' own."""
    import socket
    import torch.distributed as dist
    from betazero_amd import distributed as bd
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        engs = [_engine("ttt", 32, 20, "hash", temp_moves=4, seed=3, game_id_base=100 * i, game_id_stride=32) for i in range(2)]
        for e in engs:
            e.run_iteration()
        calls = []
        real = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        pooled = bd.gather_examples(engs)
        dist.all_gather_into_tensor = real
        assert len(calls) == 1
        own = np.concatenate([e.examples().own for e in engs])
        pi = np.concatenate([e.examples().pi for e in engs])
        gid = np.concatenate([e.examples().game for e in engs])
        assert np.array_equal(pooled.own, own) and np.array_equal(pooled.pi.view(np.uint32), pi.view(np.uint32))
        assert np.array_equal(pooled.game, gid) and len(pooled) > 300
    finally:
        dist.destroy_process_group()
        for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            os.environ.pop(k, None)


def test_examples_do_not_depend_on_how_the_games_are_sharded():
    """rank r plays global ids r*B + i and the RNG is keyed by the global id, so the pooled examples of G ranks
    are the examples one GPU would have produced for the same ids.  Two engines configured as the two ranks of a 2-GPU run
    (game_id_base 0 / B, stride 2B) against ONE engine playing all 2B games (openings by id, tau = 1 sampling by id):
    identical rows game by game -- with the synthetic evaluator and with the bf16 net in the loop."""
    from betazero_amd.net import DeviceNet
    B, sims = 24, 12
    dn = DeviceNet.from_module(_net(64, 2, bf16=True), 2 * B)
    for ev, net in (("hash", None), ("net_bf16", dn)):
        whole = _engine("reversi", 2 * B, sims, ev, net=net, temp_moves=8, openings=1, seed=11, game_id_base=0, game_id_stride=2 * B)
        halves = [_engine("reversi", B, sims, ev, net=net, temp_moves=8, openings=1, seed=11, game_id_base=r * B, game_id_stride=2 * B)
                  for r in range(2)]
        for e in [whole] + halves:
            e.run_iteration()
        a = whole.examples()
        for r, h in enumerate(halves):
            b = h.examples()
            assert sorted(set(b.game)) == list(range(r * B, (r + 1) * B))
            for gid in range(r * B, (r + 1) * B):
                ma, mb = a.game == gid, b.game == gid
                assert np.array_equal(a.own[ma], b.own[mb]) and np.array_equal(a.act[ma], b.act[mb]), (ev, gid)
                assert np.array_equal(a.pi[ma].view(np.uint32), b.pi[mb].view(np.uint32)) and np.array_equal(a.z[ma], b.z[mb]), (ev, gid)


def _rows_equal(a, b, what):
    assert np.array_equal(a.game, b.game) and np.array_equal(a.ply, b.ply), what
    assert np.array_equal(a.own, b.own) and np.array_equal(a.opp, b.opp) and np.array_equal(a.act, b.act), what
    assert np.array_equal(a.pi.view(np.uint32), b.pi.view(np.uint32)), what
    assert np.array_equal(a.z, b.z) and np.array_equal(a.mover, b.mover), what


@pytest.mark.parametrize("ev,B", [("net_bf16", 48), ("net_bf16", 61), ("net_fp8", 32)], ids=["bf16_48", "bf16_ragged_61", "fp8_32"])
def test_pipelined_self_play_two_streams_equals_the_engines_run_alone(ev, B):
    """The shape the headline is measured on -- two engines sharing ONE DeviceNet, interleaved on two HIP streams
    (betazero_amd.engine.PipelinedSelfPlay, what bench.py and self_play() run) -- for a whole iteration, against the
    SAME two engines run alone, one after the other, on the default stream: identical examples bit for bit (game ids,
    positions, actions, pi bits, z, movers), i.e. sharing the net's workspace and interleaving on the chip changes no
    result.  Also through the packed block (pack kernels) vs the raw-block unpack, and one engine of all B games."""
    from betazero_amd.engine import PipelinedSelfPlay, concat_examples, packed_block_header
    from betazero_amd.net import DeviceNet
    sims = 24
    mod = _net(128, 6, bf16=True) if ev == "net_fp8" else _net(64, 2, bf16=True)
    if ev == "net_fp8":
        from betazero_amd.quant import fake_quantize_fp8_
        fake_quantize_fp8_(mod)
    dn = DeviceNet.from_module(mod, B)
    kw = dict(temp_moves=8, openings=1, seed=3)
    sp = PipelinedSelfPlay("reversi", B, sims, ev, dn, pipelines=2, game_id_base=100, game_id_stride=1000, **kw)
    assert len({s.cuda_stream for s in sp.streams}) == 2 and 0 not in {s.cuda_stream for s in sp.streams}
    plies = sp.run_iteration()
    assert plies >= 50 and sp.status() == (0, B)
    blk = sp.pack_examples()
    got = sp.examples()
    h = packed_block_header(blk)
    assert h["n_games"] == B and h["n_rows"] == len(got) and h["dropped_rows"] == 0
    raw = concat_examples([e.examples() for e in sp.engines])  # the raw fixed-capacity blocks, unpacked the old way
    _rows_equal(got, raw, "packed vs raw unpack")
    alone = []
    for i, n in enumerate(sp.sizes):  # the same two engines, alone, default stream, one after the other
        e = _engine("reversi", n, sims, ev, net=dn, game_id_base=100 + sum(sp.sizes[:i]), game_id_stride=1000, **kw)
        e.run_iteration()
        alone.append(e.examples())
    _rows_equal(got, concat_examples(alone), "two streams vs alone")
    whole = _engine("reversi", B, sims, ev, net=dn, game_id_base=100, game_id_stride=1000, **kw)
    whole.run_iteration()
    _rows_equal(got, whole.examples(), "two pipelines vs one engine of all games")
    assert sorted(set(got.game)) == list(range(100, 100 + B))


def test_pipelined_self_play_fp32_net_two_streams_vs_oracle_games():
    """the same interleaved shape with the exact-fp32 net: every game equals the oracle's game (actions, pi bits, z)"""
    from betazero_amd.engine import PipelinedSelfPlay
    from betazero_amd.net import DeviceNet
    m = _net(32, 2, seed=4)
    B, sims = 10, 20
    dn, on = DeviceNet.from_module(m, B), orc.Net(32, 2, 64, m.flat_params())
    sp = PipelinedSelfPlay("reversi", B, sims, "net_f32", dn, pipelines=2, game_id_base=40, game_id_stride=64, temp_moves=6,
                           openings=1, seed=9)
    sp.run_iteration()
    ex = sp.examples()
    winners, lens = sp.winners()
    for g in range(B):
        r = orc.selfplay_game(orc.GAME_REVERSI, 40 + g, sims, orc.EVAL_NET_F32, 6, 1, 9, net=on)
        mk = ex.game == 40 + g
        assert lens[0, g] == len(r["own"]) and winners[0, g] == r["winner"], g
        assert np.array_equal(ex.own[mk], r["own"]) and np.array_equal(ex.act[mk], r["act"]), g
        assert np.array_equal(ex.pi[mk].view(np.uint32), r["pi"].view(np.uint32)) and np.array_equal(ex.z[mk], r["z"]), g


def test_self_play_entry_runs_the_pipelined_shape_and_pack_overflow_is_loud():
    """self_play() -- the batched collect_game_data (generate_training_games.py:25-38) -- runs two pipelines with a net
    evaluator and returns the rows a single engine returns; a packed block that is too small says so instead of
    dropping rows silently; the stream pair was chosen by the overlap probe."""
    from betazero_amd.engine import (PipelinedSelfPlay, packed_block_header, pipeline_stream_info, pipeline_streams,
                                     self_play, stream_overlap_ratio)
    from betazero_amd.net import DeviceNet
    dn = DeviceNet.from_module(_net(64, 2, bf16=True), 16)
    s, pi, z, ex = self_play("reversi", 16, 10, net=dn, seed=5, temp_moves=4, openings=1)
    s1, pi1, z1, ex1 = self_play("reversi", 16, 10, net=dn, seed=5, temp_moves=4, openings=1, pipelines=1)
    _rows_equal(ex, ex1, "self_play with 2 pipelines vs 1")
    assert s.shape == (len(ex), 8, 8) and pi.shape == (len(ex), 65) and np.array_equal(z, ex.z)
    st = pipeline_streams(DEV, 2)
    info = pipeline_stream_info(DEV, 2)
    assert info is not None and stream_overlap_ratio(st[0], st[1]) < 1.5, info  # the chosen pair really overlaps
    sp = PipelinedSelfPlay("reversi", 8, 6, "hash", pipelines=2, openings=1)
    sp.run_iteration()
    small = sp.pack_examples(cap_rows=100)  # 8 games x ~58 rows do not fit
    with pytest.raises(RuntimeError, match="did not fit"):
        packed_block_header(small)
    with pytest.raises(ValueError, match="sims must be in 1..8189"):
        _engine("reversi", 4, 9000, "hash")


def test_pipeline_streams_probe_rejects_a_serialising_pair_in_a_fresh_process():
    """profiles/r04_stream_pair_probe.txt: the third and fourth stream torch hands out in a process serialise with each
    other (probe ratio 2.05 against 1.03 for every other pair; 158 against 171 games/s).  A fresh process that has
    already taken two streams from the pool gets exactly that pair as its first candidate: pipeline_streams() must see
    it (bz_stream_overlap_probe) and move on, whatever the first candidate measured."""
    import subprocess
    import sys
    code = ("import sys, json, torch; sys.path.insert(0, %r)\n"
            "from betazero_amd.engine import pipeline_streams, pipeline_stream_info, stream_overlap_ratio\n"
            "taken = [torch.cuda.Stream(device='cuda:0') for _ in range(2)]\n"
            "st = pipeline_streams('cuda:0', 2)\n"
            "print(json.dumps({'info': pipeline_stream_info('cuda:0', 2), 'picked_ratio': stream_overlap_ratio(st[0], st[1])}))\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print("stream probe in a process whose pool had handed out two streams:", out)
    cand = out["info"]["overlap_ratio_of_candidates"]
    assert out["picked_ratio"] < 1.5 and cand[out["info"]["picked"]] < 1.5
    assert all(c >= 1.5 for c in cand[:out["info"]["picked"]])  # everything before the pick was rejected for a reason


def test_cfg3_full_size_two_pipelines_equal_one_engine_bit_for_bit():
    """the headline's exact configuration -- PipelinedSelfPlay: 2 x 2048 games on two streams, 800 simulations, bf16 MFMA
    net, openings + tau = 1 -- for two moves against ONE engine holding all 4096 games on the default stream: the same
    example rows bit for bit (positions, actions, pi bits, movers), the same work counters, no error flag.  Rows do not
    depend on their batch neighbours, on the stream, or on the host thread's bounded run-ahead."""
    from betazero_amd.engine import PipelinedSelfPlay
    from betazero_amd.net import DeviceNet
    B, sims = 4096, 800
    dn = DeviceNet.from_module(_net(128, 6, bf16=True), B)
    kw = dict(temp_moves=8, openings=1, seed=0, rounds=1)
    sp = PipelinedSelfPlay("reversi", B, sims, "net_bf16", dn, pipelines=2, **kw)
    one = _engine("reversi", B, sims, "net_bf16", net=dn, **kw)
    sp.reset_games(); sp.reset_counters()
    one.reset_games(); one.reset_counters()
    for _ in range(2):
        sp.step(False)
        one.search(); one.play(False)
    assert sp.status() == (B, 0) and one.status() == (B, 0)   # status() raises on any engine error flag
    a = [e.example_tensors() for e in sp.engines]
    b = one.example_tensors()
    for f in ("own", "opp", "act", "mover"):
        got = torch.cat([t[f][0, :, :2] for t in a]).cpu().numpy()
        assert np.array_equal(got, b[f][0, :, :2].cpu().numpy()), f
    got = torch.cat([t["pi"][0, :, :2] for t in a]).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), b["pi"][0, :, :2].cpu().numpy().view(np.uint32))
    ca, cb = sp.counters(), one.counters()
    assert ca == cb and ca["n_sims"] == 2 * B * sims and ca["n_net_leaves"] + ca["n_cache_hits"] == ca["n_sims"] + 2 * B, (ca, cb)


def test_bench_two_ranks_end_to_end_on_one_gpu_gloo():
    """`python bench.py --gpus 2` for real -- two rank processes with real engines, the timed region, the ONE all-gather
    of the example blocks, max-over-ranks timing and the per-rank proof -- as far as one GPU allows: the ranks share the
    card over gloo (BZ_DIST_BACKEND=gloo; under nccl the same command is refused for want of a second device, which is
    asserted too).  Small settings: 256 games per rank, 32 simulations."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--games", "256", "--sims", "32", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline", "--no-secondary"]
    r = subprocess.run(cmd, env=dict(env, BZ_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    rk = out["ranks"]
    assert rk["backend"] == "gloo" and rk["world_size"] == 2 and rk["distinct_devices"] == 1
    assert [p["rank"] for p in rk["per_rank"]] == [0, 1] and len({p["pid"] for p in rk["per_rank"]}) == 2
    assert all(p["device_uuid"] and p["games_finished"] >= 0 and p["seconds"] > 0 for p in rk["per_rank"])
    assert abs(sum(p["games_finished"] for p in rk["per_rank"]) / max(p["seconds"] for p in rk["per_rank"]) - out["value"]) \
        <= 0.02 * out["value"] + 1e-9  # value = all ranks' games / the slowest rank's time
    assert "bytes received per rank" in out["config"]["parallelism"]
    # the exchange: finished games only, buffers allocated before the clock, every rank's count arrived in its header
    assert out["pooled"]["collectives_in_timed_region"] == 1 and out["pooled"]["games"] >= sum(p["games_finished"] for p in rk["per_rank"])
    assert 0 < out["host"]["host_launch_cpu_frac"] and out["pooled"]["bytes_received_per_rank"] == 2 * out["pooled"]["block_bytes_per_rank"]
    if torch.cuda.device_count() < 2:  # RCCL needs one device per rank: a clear refusal, not a crash inside init
        r2 = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r2.returncode != 0 and "GPU(s) visible" in (r2.stderr + r2.stdout)


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_multi_gpu_branch_runs_under_rccl_at_world_size_1(launcher):
    """bench.py's N > 1 branch on the REAL backend ("nccl" = RCCL), as far as one GPU allows: `--force-collective` makes a
    world of one rank create the process group with `init_process_group("nccl", device_id=...)`, allocate GatherBuffers on
    the device, run the pack kernels, the ONE all_gather_into_tensor, the device-side all_reduces of [dt, games] and the
    per-rank proof -- every `ctx.dist` branch the 2/4/8-GPU driver run takes, so that run is not the code's first.  Started
    as a child process (under torchrun, and by bench.py itself) before anything here touches the GPU in THAT process.
    What is pooled: complete games only (the reference's collect_game_data, SL/generate_training_games.py:30-36)."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BZ_DIST_BACKEND", "BZ_BENCH_REHEARSAL"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    args = ["--gpus", "1", "--force-collective", "--games", "512", "--sims", "32", "--steps", "70", "--warmup", "2",
            "--no-cpu-baseline", "--no-secondary"]
    if launcher == "torchrun":
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    # stdout is the ONE JSON line and nothing else (RCCL's version banner, printed on stdout when the communicator is built,
    # must not land in front of it: bench.py keeps descriptor 1 for the line)
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["forced_collective_path"] is True and out["value"] > 0
    rk = out["ranks"]
    assert rk["backend"] == "nccl (RCCL)" and rk["world_size"] == 1 and rk["distinct_devices"] == 1
    p = rk["per_rank"][0]
    assert p["rank"] == 0 and p["device_uuid"] and p["seconds"] > 0
    # 70 steps of a 58-ply steady-state pool: every slot finishes at least one game inside the timed region
    assert p["games_finished"] >= 512
    pooled = out["pooled"]
    assert pooled["collectives_in_timed_region"] == 1 and pooled["dropped_rows"] == 0
    assert pooled["games"] >= p["games_finished"] and pooled["rows"] >= 20 * pooled["games"]   # (staggered starts: the first games of the pool are short)
    assert pooled["bytes_received_per_rank"] == pooled["block_bytes_per_rank"]  # world size 1: one block
    assert abs(p["games_finished"] / p["seconds"] - out["value"]) <= 0.02 * out["value"]  # the device all_reduces returned this rank's own figures
    assert "bytes received per rank" in out["config"]["parallelism"] and out["roofline"]["frac"] > 0


def test_net_evaluators_serve_the_small_reversi_boards():
    """the reference plays Reversi on 6x6 (its GUI) and 4x4 (its demo) too: those boards sit in the top-left corner of the
    net's 8x8 planes, so the same net serves them -- self-play with the exact-fp32 net in the loop is bit-exact vs the
    oracle on both sizes, the bf16 MFMA net plays legal games, and MCTSPlayer / NetPlayer accept the boards."""
    import betazero_amd as bz
    from betazero_amd.net import DeviceNet
    m = _net(32, 2, seed=6)
    dn, on = DeviceNet.from_module(m, 8), orc.Net(32, 2, 64, m.flat_params())
    for game, og, size in (("reversi6", orc.GAME_REVERSI6, 6), ("reversi4", orc.GAME_REVERSI4, 4)):
        eng = _engine(game, 6, 14, "net_f32", net=dn, temp_moves=4, seed=5, game_id_base=20)
        eng.run_iteration()
        ex = eng.examples()
        winners, lens = eng.winners()
        for g in range(6):
            r = orc.selfplay_game(og, 20 + g, 14, orc.EVAL_NET_F32, 4, 0, 5, net=on)
            mk = ex.game == 20 + g
            assert lens[0, g] == len(r["own"]) and winners[0, g] == r["winner"]
            assert np.array_equal(ex.act[mk], r["act"]) and np.array_equal(ex.pi[mk].view(np.uint32), r["pi"].view(np.uint32))
        assert ((ex.own | ex.opp) & ~np.uint64(sum(((1 << size) - 1) << (8 * r_) for r_ in range(size)))).max() == 0
    big = DeviceNet.from_module(_net(128, 6, bf16=True), 8)
    eng = _engine("reversi6", 8, 16, "net_bf16", net=big, temp_moves=4)
    eng.run_iteration()
    assert (eng.winners()[1][0] >= 5).all()
    b6 = bz.ReversiBoard(size=6)
    assert bz.MCTSPlayer(1, sims=20, net=big).get_move(b6) in b6.generate_possible_moves(1)
    assert bz.NetPlayer(1, big).get_move(b6) in b6.generate_possible_moves(1)


def test_external_torch_evaluator_in_the_search():
    """SelfPlayEngine.search_external / MCTSPlayer(evaluator=callable): any torch module as the leaf evaluator.  (1) a
    callable that reproduces the synthetic hash evaluator gives the fused kernel's root statistics bit for bit; (2) a torch
    MLP over the 9 tic-tac-toe cells (the shape of the reference's TicTacToeNet, SL/neural_networks.py) drives legal games
    through the reference-style loop."""
    import random
    import betazero_amd as bz
    d = np.load(os.path.join(G, "ttt_exhaustive.npz"))
    live = d["pos"][d["pos"][:, 4] == 0][:64]
    tm = np.where(live[:, 2] == 1, 1, -1).astype(np.int8)
    own = np.where(tm == 1, live[:, 0], live[:, 1]).astype(np.uint64)
    opp = np.where(tm == 1, live[:, 1], live[:, 0]).astype(np.uint64)

    def hash_fn(o, p, kind):
        oc, pc = o.cpu().numpy().view(np.uint64), p.cpu().numpy().view(np.uint64)
        lg, v = np.zeros((len(oc), 9), np.float32), np.zeros(len(oc), np.float32)
        for i in range(len(oc)):
            lg[i], v[i] = orc.eval_hash(int(oc[i]), int(pc[i]), 9)
        return torch.from_numpy(lg).to(DEV), torch.from_numpy(v).to(DEV)
    e1, e2 = _engine("ttt", 64, 30, "external"), _engine("ttt", 64, 30, "hash")
    for e in (e1, e2):
        e.set_roots(own, opp, tm)
    e1.search_external(hash_fn)
    e2.search()
    a, b = e1.root_stats(), e2.root_stats()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(9, 32), torch.nn.ReLU(), torch.nn.Linear(32, 10)).to(DEV)

    def mlp_fn(o, p, kind):
        sh = torch.arange(9, device=DEV)
        x = ((o[:, None] >> sh) & 1).float() - ((p[:, None] >> sh) & 1).float()  # canonical: +1 = side to move
        with torch.no_grad():
            y = mlp(x)
        return y[:, :9], torch.tanh(y[:, 9])
    random.seed(1)
    pl = bz.MCTSPlayer(1, sims=40, evaluator=mlp_fn)
    positions, winner = bz.TicTacToeHeadless(pl, bz.RandomPlayer()).play()
    assert winner in (1, 0, -1) and 6 <= len(positions) <= 10
