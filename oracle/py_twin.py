"""Build-authored Python MCTS twin over ANY board classes that expose the
reference's Game API  -- TEST INFRASTRUCTURE ONLY.

Two users:
  * oracle/gen_golden.py binds it to the IMPORTED reference boards (build container only) to
    generate fixture F7: every env transition inside the search is then reference-computed;
  * bench.py's cpu_baseline leg binds it to betazero_amd's API-compatible boards and times it on
    one host core ("python_loop": the closest analogue of "the reference Python CPU path" that
    can run on the GPU box, where /root/reference does not exist -- SURVEY.md 8(d)(ii)).
The reference has no MCTS; the turn loop follows reversi_terminal.py:16-38 and
the trajectory contract tic_tac_toe.py:13-34.  numpy-float32 arithmetic, one rounding per
operation, in the order of the spec -- so results are bit-identical to oracle/bz_oracle.c."""
import numpy as np

M64 = (1 << 64) - 1


def rev_bits(board, player):
    """bit 8*r+c for every cell equal to player (all sizes share the 8-stride)."""
    b = 0
    n = board.shape[0]
    for r in range(n):
        for c in range(n):
            if board[r][c] == player:
                b |= 1 << (8 * r + c)
    return b


def rev_mask(moves):
    m = 0
    for r, c in moves:
        m |= 1 << (8 * r + c)
    return m


def ttt_bits(board, player):
    b = 0
    for r in range(3):
        for c in range(3):
            if board[r][c] == player:
                b |= 1 << (3 * r + c)
    return b



f32 = np.float32


def mix64(x):
    x &= M64
    x ^= x >> 33
    x = (x * 0xFF51AFD7ED558CCD) & M64
    x ^= x >> 33
    x = (x * 0xC4CEB9FE1A85EC53) & M64
    x ^= x >> 33
    return x


def rng_draw(seed, gid, ply):
    h = mix64((seed * 0x9E3779B97F4A7C15 + gid) & M64)
    return mix64(h ^ ((ply * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & M64))


def expf_spec(x):
    """exp for x <= 0, single float32 roundings, no fma."""
    x = f32(x)
    if x < f32(-87.0):
        return f32(0.0)
    t = x * f32(1.44269504)
    n = np.floor(t + f32(0.5)).astype(np.float32)
    r = x - n * f32(0.693359375)
    r = r - n * f32(-2.12194440e-4)
    p = f32(1.9875691500e-4)
    for c in (1.3981999507e-3, 8.3334519073e-3, 4.1665795894e-2, 1.6666665459e-1, 5.0000001201e-1):
        p = p * r + f32(c)
    rr = r * r
    p = p * rr
    p = p + r
    p = p + f32(1.0)
    scale = np.array([(int(n) + 127) << 23], dtype=np.uint32).view(np.float32)[0]
    return f32(p * scale)


def logf_spec(x):
    """ln for normal x > 0 (Cephes logf), single float32 roundings, no fma."""
    b = int(np.array([x], dtype=np.float32).view(np.uint32)[0])
    e = (b >> 23) - 127
    m = np.array([(b & 0x007FFFFF) | 0x3F800000], dtype=np.uint32).view(np.float32)[0]
    if m > f32(1.41421356):
        m = m * f32(0.5)
        e = e + 1
    z = m - f32(1.0)
    p = f32(7.0376836292e-2)
    for c in (-1.1514610310e-1, 1.1676998740e-1, -1.2420140846e-1, 1.4249322787e-1, -1.6668057665e-1,
              2.0000714765e-1, -2.4999993993e-1, 3.3333331174e-1):
        p = p * z + f32(c)
    zz = z * z
    y = z * zz
    y = y * p
    fe = f32(e)
    t = fe * f32(-2.12194440e-4)
    y = y + t
    t = f32(0.5) * zz
    y = y - t
    r = z + y
    t = fe * f32(0.693359375)
    r = r + t
    return f32(r)


def u01_spec(bits):
    return (f32(bits >> 41) + f32(0.5)) * f32(1.0 / 8388608.0)


def rng_noise(seed, gid, ply, idx):
    h = rng_draw(seed ^ 0xD1B54A32D192ED03, gid, ply)
    return mix64((h + idx * 0x9E3779B97F4A7C15) & M64)


def gamma_spec(alpha, seed, gid, ply, edge):
    """Gamma(alpha, 1), 0 < alpha <= 1: exponential at alpha = 1, Johnk's generator below (<= 16 attempts)"""
    alpha = f32(alpha)
    base = edge * 64
    if not (alpha < f32(1.0)):
        return f32(-logf_spec(u01_spec(rng_noise(seed, gid, ply, base))))
    ia, ib = f32(1.0) / alpha, f32(1.0) / (f32(1.0) - alpha)
    x, s, k = f32(0.0), f32(0.0), base
    for t in range(16):
        k = base + 3 * t
        lu = logf_spec(u01_spec(rng_noise(seed, gid, ply, k))) * ia
        lv = logf_spec(u01_spec(rng_noise(seed, gid, ply, k + 1))) * ib
        x = expf_spec(lu if lu < f32(0.0) else f32(0.0))
        y = expf_spec(lv if lv < f32(0.0) else f32(0.0))
        s = x + y
        if s <= f32(1.0):
            break
    if not (s > f32(0.0)):
        return f32(0.0)
    e = -logf_spec(u01_spec(rng_noise(seed, gid, ply, k + 2)))
    g = e * x
    return f32(g / s)


def eval_hash(own, opp, na):
    h = mix64(((own * 0x9E3779B97F4A7C15) & M64) ^ mix64((opp + 0x632BE59BD9B4E019) & M64))
    logits = []
    for a in range(na):
        q = mix64((h + a * 0xD6E8FEB86659FD93) & M64)
        logits.append(f32((q >> 40) - (1 << 23)) * f32(1.0 / 4194304.0))
    q = mix64(h ^ 0xA5A5A5A5A5A5A5A5)
    v = f32((q >> 40) - (1 << 23)) * f32(1.0 / 8388608.0)
    return logits, v


class Twin:
    """Sequential MCTS (spec M1-M5) over board objects with the reference's Game API:
    every env transition goes through generate_possible_moves / make_move / is_game_over / get_score
    of the board classes handed in."""

    def __init__(self, game, eval_kind, c_puct=1.5, boards=None, dir_alpha=0.0, dir_eps=0.0, reuse=False):
        # boards = (ReversiBoard, TicTacToeBoard) classes with the reference's Game API
        self.ReversiBoard, self.TicTacToeBoard = boards
        # opt-in search features; the noise key (seed, gid, ply) is set per search
        self.dir_alpha, self.dir_eps, self.reuse = f32(dir_alpha), f32(dir_eps), reuse
        self.noise_key = (0, 0, 0)
        self.size = {"reversi6": 6, "reversi4": 4}.get(game, 8)
        self.label = game
        game = "reversi" if game.startswith("reversi") else game
        self.game, self.eval_kind, self.c = game, eval_kind, f32(c_puct)
        self.na = 9 if game == "ttt" else 65

    # --- env through the reference only
    def moves(self, b, p):
        if self.game == "ttt":
            return [3 * r + c for r, c in b.generate_possible_moves()]
        return [8 * r + c for r, c in b.generate_possible_moves(p)]

    def play(self, b, p, a):
        if self.game == "ttt":
            return b.make_move(a // 3, a % 3, p)
        if a == 64:
            return b
        return b.make_move(a // 8, a % 8, p)

    def terminal(self, b):
        if self.game == "ttt":
            over, w = b.is_game_over()
            return over, (w if over else 0)
        if b.is_game_over():
            return True, b.get_score()[0]
        return False, 0

    def bits(self, b, p):
        fn = ttt_bits if self.game == "ttt" else rev_bits
        return fn(b.board, p), fn(b.board, -p)

    def evaluate(self, b, p):
        if self.eval_kind == "uniform":
            return [f32(0.0)] * self.na, f32(0.0)
        own, opp = self.bits(b, p)
        return eval_hash(own, opp, self.na)

    def expand(self, node):
        logits, v = self.evaluate(node["b"], node["p"])
        mv = self.moves(node["b"], node["p"])
        if not mv:
            node["edges"] = [{"a": 64, "N": 0, "W": f32(0), "P": f32(1), "child": None}]
            return v
        m = max(logits[a] for a in mv)
        es = [expf_spec(logits[a] - m) for a in mv]
        s = f32(0.0)
        for e in es:
            s = s + e
        node["edges"] = [{"a": a, "N": 0, "W": f32(0), "P": f32(e / s), "child": None} for a, e in zip(mv, es)]
        return v

    def root_noise(self, root):
        """P' = (1 - eps) P + eps eta, eta ~ Dirichlet(alpha) over the root's edges in ascending action order"""
        if not (self.dir_eps > 0) or root["edges"] is None:
            return
        seed, gid, ply = self.noise_key
        g = [gamma_spec(self.dir_alpha, seed, gid, ply, i) for i in range(len(root["edges"]))]
        gs = f32(0.0)
        for x in g:
            gs = gs + x
        if not (gs > 0):
            return
        keep = f32(1.0) - self.dir_eps
        for e, x in zip(root["edges"], g):
            t1 = keep * e["P"]
            t2 = self.dir_eps * f32(x / gs)
            e["P"] = f32(t1 + t2)

    def new_node(self, b, p):
        over, w = self.terminal(b)
        return {"b": b, "p": p, "term": over, "tv": w * p, "edges": None}

    def simulate(self, root):
        node, path = root, []
        while True:
            if node["term"]:
                v = f32(node["tv"])
                break
            sumN = sum(e["N"] for e in node["edges"])
            sq = np.sqrt(f32(max(sumN, 1)))
            best, bests = None, f32(-np.inf)
            for e in node["edges"]:
                q = e["W"] / f32(e["N"]) if e["N"] > 0 else f32(0.0)
                u = self.c * e["P"]
                u = u * sq
                u = u / (f32(1.0) + f32(e["N"]))
                s = q + u
                if s > bests:
                    best, bests = e, s
            path.append(best)
            if best["child"] is not None:
                node = best["child"]
                continue
            ch = self.new_node(self.play(node["b"], node["p"], best["a"]), -node["p"])
            best["child"] = ch
            v = f32(ch["tv"]) if ch["term"] else self.expand(ch)
            break
        val = -v
        for e in reversed(path):
            e["N"] += 1
            e["W"] = f32(e["W"] + val)
            val = -val

    def search(self, b, p, sims):
        root = self.new_node(b, p)
        assert not root["term"]
        self.expand(root)
        self.root_noise(root)
        for _ in range(sims):
            self.simulate(root)
        return root

    def selfplay(self, gid, sims, temp_moves, openings, seed):
        if self.game == "ttt":
            b = self.TicTacToeBoard()
        else:
            b = self.ReversiBoard(size=self.size)
        p, made, passes = 1, 0, 0
        if self.game == "reversi" and openings and self.size == 8:
            k = gid % 12
            for pick in (k // 3, k % 3):
                a = self.moves(b, p)[pick]
                b = self.play(b, p, a)
                p, made = -p, made + 1
        ex = []
        kept = None  # subtree reuse: the node that becomes the next root, statistics and all
        while True:
            self.noise_key = (seed, gid, made)
            if kept is not None:
                root = kept
                self.root_noise(root)
                for _ in range(sims):
                    self.simulate(root)
            else:
                root = self.search(b, p, sims)
            sumN = sum(e["N"] for e in root["edges"])
            pi = [f32(0.0)] * self.na
            for e in root["edges"]:
                pi[e["a"]] = f32(e["N"]) / f32(sumN)
            if made < temp_moves:
                r = rng_draw(seed, gid, made) % sumN
                cum = 0
                for e in root["edges"]:
                    cum += e["N"]
                    if cum > r:
                        pick = e
                        break
            else:
                pick, bn = root["edges"][0], 0
                for e in root["edges"]:
                    if e["N"] > bn:
                        pick, bn = e, e["N"]
            own, opp = self.bits(b, p)
            ex.append((own, opp, pi, p, pick["a"]))
            b = self.play(b, p, pick["a"])
            p, made = -p, made + 1
            over, w = self.terminal(b)
            if over:
                return ex, w, passes
            keep_node, keep_N = pick["child"], pick["N"]
            if not self.moves(b, p):
                p, passes = -p, passes + 1
                if keep_node is not None:  # the kept root lies behind the child's only edge, the pass
                    pe = keep_node["edges"][0]
                    keep_node, keep_N = pe["child"], pe["N"]
            kept = keep_node if (self.reuse and keep_node is not None and keep_N + sims + 2 <= 4 * (sims + 2)) else None
