"""GPU tests of the hand-written training kernels (csrc/bz_train.hip, through the C ABI): the residual tower's forward
with saved activations, backward-data and backward-weights against torch.autograd in fp32 on the same (bf16-rounded)
weights and inputs.  The reference has no conv net, so the oracle here is the plain PyTorch fp32 module
of the same op; tolerances are bf16's (8 significant bits per stored activation / gradient) and are stated per check."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(C, L, n, seed):
    g = torch.Generator().manual_seed(seed)
    bf = lambda t: t.bfloat16().float()  # noqa: E731
    x0 = bf(torch.relu(torch.randn((n, C, 8, 8), generator=g)))                # the stem's output: non-negative
    W = bf(torch.randn((L, C, C, 3, 3), generator=g) * (1.5 / (9 * C)) ** 0.5)   # keeps the activations O(1) through the tower
    b = bf(torch.randn((L, C), generator=g) * 0.1)
    gy = bf(torch.randn((n, C, 8, 8), generator=g))
    return [t.to(DEV) for t in (x0, W, b, gy)]


def tower_reference(x0, W, b, round_bf16=True, relu_masks=None):
    """the same residual tower in plain torch (conv2d + relu + skip) -- the autograd reference of the GPU tests.
    round_bf16: round every layer's output to bf16 (straight-through), as the kernels store it.  relu_masks: a list
    of L boolean [n, C, 8, 8] tensors; layer l's ReLU then is "multiply by relu_masks[l]" -- with the kernels' own
    patterns (act[l + 1] > 0) the reference takes the same branch of every ReLU as the kernels did."""
    import torch.nn.functional as F
    rnd = (lambda t: t + (t.bfloat16().float() - t).detach()) if round_bf16 else (lambda t: t)
    act = (lambda z, l: F.relu(z)) if relu_masks is None else (lambda z, l: z * relu_masks[l])
    a = x0
    for blk in range(W.shape[0] // 2):
        h = rnd(act(F.conv2d(a, W[2 * blk], b[2 * blk], padding=1), 2 * blk))
        a = rnd(act(F.conv2d(h, W[2 * blk + 1], b[2 * blk + 1], padding=1) + a, 2 * blk + 1))
    return a


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


# (at 64 channels batches below 1536 run 4 positions per workgroup, larger ones 8: both geometries are covered)
@pytest.mark.parametrize("C,L,n", [(64, 4, 16), (128, 2, 8), (128, 6, 12), (64, 8, 40), (64, 2, 1536)])
def test_tower_forward_and_saved_activations_vs_torch(C, L, n):
    """every layer's stored activation equals the torch reference (conv2d + bias + skip + relu, outputs rounded to
    bf16 like the kernel's) to 2 bf16 ulps of the tensor's range (measured ~4e-3 relative to max); the function's
    result is the last stored activation.  (The ReLU bits kept for the backward are checked through the gradients.)"""
    import torch.nn.functional as F
    from betazero_amd.train_kernels import TowerPlan, tower_apply
    x0, W, b, _ = _case(C, L, n, 1)
    plan = TowerPlan(C, L, n)
    y = tower_apply(x0, W, b, plan)
    torch.cuda.synchronize()
    a = x0
    for l in range(L):
        z = F.conv2d(a if l % 2 == 0 else h, W[l], b[l], padding=1)  # noqa: F821
        if l % 2 == 0:
            h = torch.relu(z).bfloat16().float()
            want = h
        else:
            a = torch.relu(z + a).bfloat16().float()
            want = a
        got = plan.acts[l + 1].float().view(n, 8, 8, C).permute(0, 3, 1, 2)
        assert _rel(got, want) < 1.6e-2, (l, _rel(got, want))
        # the bulk agrees much better than the worst element (a bf16 ulp of a large value)
        assert float((got - want).abs().mean() / want.abs().mean()) < 2e-3, l
    assert torch.equal(y, plan.acts[L].float().view(n, 8, 8, C).permute(0, 3, 1, 2))


# ((128, 12, 1024) is the shape bench.py's `secondary.train_step` times: 12 conv layers x 128 channels x batch 1024 --
# bz_train_wgrad_splits depends on the batch, so the timed shape is a tested shape)
@pytest.mark.parametrize("C,L,n", [(64, 4, 16), (128, 4, 32), (64, 8, 64), (128, 12, 64), (64, 4, 1536), (128, 12, 1024)])
def test_tower_gradients_vs_torch_autograd_fp32(C, L, n):
    """d/dx0, d/dW, d/db of sum(y * gy) through the kernels against torch.autograd over the fp32 reference tower (its
    layer outputs rounded to bf16 with a straight-through gradient, as the kernels store them).  The kernels keep
    activations AND gradients in bf16 between layers, the reference keeps gradients in fp32: agreement to ~1 % of
    each tensor's range is what bf16 storage allows (measured on MI355X: mean error 2e-3 .. 3e-3 of the mean magnitude,
    dW / db within 1.2 % of their range)."""
    from betazero_amd.train_kernels import TowerPlan, tower_apply
    x0, W, b, gy = _case(C, L, n, 2)
    plan = TowerPlan(C, L, n)
    xs, Ws, bs = (t.clone().requires_grad_(True) for t in (x0, W, b))
    y = tower_apply(xs, Ws, bs, plan)
    (y * gy).sum().backward()
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x0, W, b))
    # the reference takes the kernels' branch of every ReLU (their stored activations say which): a pre-activation at
    # a rounding boundary would otherwise be 0 on one side and tiny on the other, and the two gradients part ways
    # down the skip connections -- a property of ReLU nets (it grows with depth: 5 % at 12 layers), not of the kernels
    masks = [(plan.acts[l + 1] > 0).view(n, 8, 8, C).permute(0, 3, 1, 2) for l in range(L)]
    yr = tower_reference(xr, Wr, br, round_bf16=True, relu_masks=masks)
    (yr * gy).sum().backward()
    torch.cuda.synchronize()
    # what remains is bf16 storage of the gradients between layers (the reference keeps them in fp32)
    def close(a, r, what):
        d, scale = (a - r).abs().detach(), r.abs().max().detach()
        frac_off = float((d > 0.03 * scale).float().mean())
        mean_rel = float(d.mean() / r.abs().mean())
        assert frac_off < max(1e-3, 2.5 / a.numel()) and mean_rel < 1.5e-2, (what, frac_off, mean_rel)
        return round(mean_rel, 5)
    errs = {"y": close(y, yr, "y"), "dx0": close(xs.grad, xr.grad, "dx0"), "dW": close(Ws.grad, Wr.grad, "dW"),
            "db": close(bs.grad, br.grad, "db"), "dW max": round(_rel(Ws.grad, Wr.grad), 5)}
    print(f"C={C} L={L} n={n}: mean |kernel - autograd| / mean |autograd| =", errs)
    # per layer, per tap and per 32 x 32 (co, ci) tile: a mirrored tap, a swapped (co, ci) or one wave's tile gone wrong
    # in one layer would hide in a global figure -- its cosine with the reference would be ~0 (or negative), not ~1
    cosine = lambda a, r: float(torch.nn.functional.cosine_similarity(a.flatten(), r.flatten(), dim=0))  # noqa: E731
    worst = 1.0
    for l in range(L):
        for t in range(9):
            a, r = Ws.grad[l, :, :, t // 3, t % 3], Wr.grad[l, :, :, t // 3, t % 3]
            for co in range(0, C, 32):
                for ci in range(0, C, 32):
                    worst = min(worst, cosine(a[co:co + 32, ci:ci + 32], r[co:co + 32, ci:ci + 32]))
    print("worst cosine over (layer, tap, 32x32 tile) of dW:", round(worst, 5))
    assert worst > 0.995, worst
    assert cosine(Ws.grad, Wr.grad) > 0.9995 and cosine(bs.grad, br.grad) > 0.9995 and cosine(xs.grad, xr.grad) > 0.999


def test_tower_gradients_do_not_depend_on_batch_slicing_and_refuse_bad_shapes():
    """the weight-gradient kernel splits the batch over workgroups (bz_train_wgrad_splits): the sum over the slices is
    the whole batch's gradient -- two half-batches add up to the full batch within fp32 summation noise; shapes the
    kernels are not built for are refused with a message, not run."""
    from betazero_amd.train_kernels import TowerPlan, tower_apply
    C, L, n = 64, 2, 32
    x0, W, b, gy = _case(C, L, n, 3)

    def grads(lo, hi):
        plan = TowerPlan(C, L, hi - lo)
        Ws = W.clone().requires_grad_(True)
        (tower_apply(x0[lo:hi], Ws, b, plan) * gy[lo:hi]).sum().backward()
        return Ws.grad
    full, a, bb = grads(0, n), grads(0, n // 2), grads(n // 2, n)
    assert _rel(a + bb, full) < 1e-5
    with pytest.raises(ValueError):
        TowerPlan(64, 2, 12)      # not a multiple of the 8 positions a workgroup holds at C = 64
    with pytest.raises(ValueError):
        TowerPlan(256, 2, 8)      # the training kernels serve 64 and 128 channels
    assert np.isfinite(float(full.abs().sum()))


@pytest.mark.parametrize("step_kernels", [True, False])
def test_graphed_train_step_on_the_tower_kernels_tracks_the_autograd_step(step_kernels):
    """GraphedTrainStep on the HIP kernels -- step_kernels=True: the whole forward / losses / backward (StepPlan, 9
    launches, no autograd); False: the residual tower's three kernels inside torch autograd (stem, heads, losses in
    torch) -- against train_step through stock autograd (MIOpen) from the same weights on the same batches: the same
    loss step by step within bf16 tolerance, and after 8 steps the tower's weights have moved the same way (cosine of
    the two weight updates > 0.9; Adam's first steps are sign-like, so tiny gradients may disagree in sign)."""
    import copy
    from betazero_amd.engine import DeviceExamples, Examples
    from betazero_amd.net import PolicyValueNet
    from betazero_amd.train import GraphedTrainStep, make_optimizer, train_step
    rng = np.random.default_rng(11)
    n = 1024
    own = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64)
    opp = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64) & ~own
    pi = rng.random((n, 65)).astype(np.float32); pi /= pi.sum(1, keepdims=True)
    z = rng.integers(-1, 2, n).astype(np.int8)
    ex = DeviceExamples.from_host(Examples(own, opp, pi, z, np.ones(n, np.int8), np.zeros(n, np.uint8), np.arange(n),
                                           np.zeros(n, np.int32), 8))
    torch.manual_seed(5)
    m1 = PolicyValueNet(64, 2, 64, fused_tower=True).cuda()
    m2 = copy.deepcopy(m1)
    w0 = m1.tower_w.detach().clone()
    g = GraphedTrainStep(m1, lr=1e-3, batch=128, step_kernels=step_kernels)
    assert g.plan is not None and (g.step_plan is not None) == step_kernels  # the kernels are in the graph
    opt = make_optimizer(m2, lr=1e-3)
    gen = torch.Generator(device=DEV).manual_seed(0)
    for step in range(8):
        idx = torch.randint(0, n, (128,), device=DEV, generator=gen)
        l1 = g(ex, idx).cpu().numpy()
        l2 = torch.stack(train_step(m2, opt, ex, idx)).cpu().numpy()
        assert np.isfinite(l1).all() and np.abs(l1 - l2).max() < 3e-2 * max(1.0, np.abs(l2).max()), (step, l1, l2)
    d1, d2 = (m1.tower_w.detach() - w0).flatten(), (m2.tower_w.detach() - w0).flatten()
    cos = float(torch.nn.functional.cosine_similarity(d1, d2, dim=0))
    print("cosine of the tower's weight updates after 8 steps, kernels vs autograd:", round(cos, 4), "last losses", l1, l2)
    assert float(d1.abs().max()) > 0 and cos > 0.9
    for name in ("stem", "pol", "polfc", "val", "v1", "v2"):   # the ends moved the same way too
        a, b = getattr(m1, name).weight.detach().flatten(), getattr(m2, name).weight.detach().flatten()
        assert float(torch.nn.functional.cosine_similarity(a, b, dim=0)) > 0.999, name


# ---- the ends of the step (csrc/bz_train_ends.hip): each kernel alone against torch fp32 on the SAME inputs, then the whole step

def _net_case(C, NB, n, seed, VH=64):
    from betazero_amd.net import PolicyValueNet
    from betazero_amd.train_kernels import StepPlan
    torch.manual_seed(seed)
    m = PolicyValueNet(C, NB, VH, fused_tower=True).cuda()
    with torch.no_grad():   # biases away from 0 and head weights large enough that every ReLU / tanh is exercised on both sides
        for p in m.parameters():
            if p.dim() == 1:
                p.normal_(0.0, 0.2)
        m.pol.weight.mul_(3.0); m.val.weight.mul_(3.0); m.polfc.weight.mul_(2.0)
    rng = np.random.default_rng(seed)
    own = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64) | (rng.integers(0, 2, n).astype(np.uint64) << np.uint64(63))
    opp = (rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64) | (rng.integers(0, 2, n).astype(np.uint64) << np.uint64(63))) & ~own
    own[0], opp[0] = np.uint64(0), np.uint64(0)                        # an empty board
    own[1], opp[1] = np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64(0)       # a full one (bit 63 set: int64 sign)
    pi = rng.random((n, 65)).astype(np.float32) ** 4
    pi /= pi.sum(1, keepdims=True)
    pi[2] = 0.0; pi[2, 64] = 1.0                                       # all mass on "pass"
    z = rng.integers(-1, 2, n).astype(np.int8)
    t = lambda a: torch.as_tensor(a).to(DEV)  # noqa: E731
    return m, StepPlan(m, n), t(own.view(np.int64)), t(opp.view(np.int64)), t(pi), t(z)


def _planes(own, opp):
    from betazero_amd.train import planes_from_bits
    return planes_from_bits(own, opp)


@pytest.mark.parametrize("C,n", [(64, 8), (128, 20)])
def test_stem_kernel_and_its_weight_gradient_vs_torch(C, n):
    """act[0] = relu(conv3x3(planes)) straight from the bitboards: equal to torch's fp32 conv2d rounded to bf16 up to one
    bf16 ulp (fp32 summation order); the stem's weight / bias gradient from a given g[0] equals torch autograd's of
    sum(pre-activation * g[0] * (act[0] > 0)) to fp32 accuracy (1e-5 of the tensor's range; same bf16 inputs on both sides)."""
    import torch.nn.functional as F
    from betazero_amd import _lib
    m, plan, own, opp, pi, z = _net_case(C, 1, n, 21)
    L, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    plan.set_batch(own, opp, pi, z)
    _lib.check(L.bz_train_stem_fwd(plan.batch_desc.data_ptr(), n, m.stem.weight.data_ptr(), m.stem.bias.data_ptr(), C, plan.acts[0].data_ptr(), st))
    x = _planes(own, opp)
    want = F.relu(F.conv2d(x, m.stem.weight, m.stem.bias, padding=1)).permute(0, 2, 3, 1).reshape(n, 64, C)
    got = plan.acts[0].float()
    assert float((got - want.bfloat16().float()).abs().max()) <= 2.0 ** -7 * float(want.abs().max()), "more than one bf16 ulp off"
    assert float((got != want.bfloat16().float()).float().mean()) < 2e-3      # (and almost everywhere identical)
    assert torch.equal(got[0], torch.relu(m.stem.bias.detach()).bfloat16().float().expand(64, C))   # the empty board: relu(bias) in every cell
    # weight gradient
    g0 = torch.randn((n, 64, C), device=DEV).bfloat16()
    plan.gs[0].copy_(g0)
    _lib.check(L.bz_train_stem_wgrad(plan.batch_desc.data_ptr(), plan.acts[0].data_ptr(), plan.gs[0].data_ptr(), n, C, plan.stem_partial.data_ptr(), st))
    _lib.check(L.bz_train_finish(_byref(plan._partials), _byref(plan._grads), C, plan.L, plan.VH, n, plan.losses.data_ptr(), None, st))
    w, b = m.stem.weight.detach().clone().requires_grad_(True), m.stem.bias.detach().clone().requires_grad_(True)
    pre = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(n, 64, C)
    (pre * (g0.float() * (plan.acts[0] > 0))).sum().backward()
    assert _rel(m.stem.weight.grad, w.grad) < 1e-5 and _rel(m.stem.bias.grad, b.grad) < 1e-5


def _byref(x):
    import ctypes
    return ctypes.byref(x)


def _heads_reference(m, x, pi, z):
    """heads + losses of PolicyValueNet in plain fp32 torch on act[L] given as [n, 64, C] (what net.py's forward does
    after the tower, train.py's two losses)"""
    import torch.nn.functional as F
    n, C = x.shape[0], x.shape[2]
    xc = x.view(n, 8, 8, C).permute(0, 3, 1, 2)
    p = m.polfc(F.relu(m.pol(xc)).flatten(1))
    v = torch.tanh(m.v2(F.relu(m.v1(F.relu(m.val(xc)).flatten(1))))).squeeze(-1)
    ce = -(pi * F.log_softmax(p, dim=1)).sum(1).mean()
    mse = F.mse_loss(v, z.float())
    return ce + mse, ce, mse


@pytest.mark.parametrize("C,n,VH", [(64, 8, 64), (128, 20, 64), (64, 1032, 24)])
def test_heads_kernel_losses_and_gradients_vs_torch_fp32(C, n, VH):
    """the heads kernel on a given act[L]: losses, d loss / d act[L] (times act[L] > 0, stored bf16) and the gradients of
    all ten head parameter tensors against torch autograd in fp32 on the same bf16 activations.  fp32 on both sides:
    1e-4 of each tensor's range (summation order); g[L] to one bf16 ulp.  n = 1032 > 4 x 256 makes workgroups take more
    than one pass; VH = 24 leaves lanes of the value head idle."""
    from betazero_amd import _lib
    m, plan, own, opp, pi, z = _net_case(C, 1, n, 22, VH)
    L, st, Ly = _lib.lib(), torch.cuda.current_stream().cuda_stream, plan.L
    x = torch.relu(torch.randn((n, 64, C), device=DEV) - 0.3).bfloat16()
    plan.acts[Ly].copy_(x)
    plan.set_batch(own, opp, pi, z)
    _lib.check(L.bz_train_heads(plan.acts[Ly].data_ptr(), plan.batch_desc.data_ptr(), n, C, VH, _byref(plan._head), plan.gs[Ly].data_ptr(),
                                plan.hv.data_ptr(), plan.dl.data_ptr(), plan.dv1.data_ptr(), plan.heads_partial.data_ptr(), st))
    _lib.check(L.bz_train_heads_wgrad(plan.hv.data_ptr(), plan.dl.data_ptr(), plan.dv1.data_ptr(), n, VH, plan.heads_w_partial.data_ptr(), st))
    _lib.check(L.bz_train_finish(_byref(plan._partials), _byref(plan._grads), C, Ly, VH, n, plan.losses.data_ptr(), None, st))
    got = {k: getattr(m, k).weight.grad.clone() for k in ("pol", "polfc", "val", "v1", "v2")}
    gotb = {k: getattr(m, k).bias.grad.clone() for k in ("pol", "polfc", "val", "v1", "v2")}
    losses, g_top = plan.losses.clone(), plan.gs[Ly].float().clone()
    for p in m.parameters():
        p.grad = None
    xr = x.float().requires_grad_(True)
    want = _heads_reference(m, xr, pi, z)
    want[0].backward()
    assert np.allclose(losses[:3].cpu().numpy(), [float(w) for w in want], rtol=2e-5, atol=1e-6), (losses, want)
    for k in got:
        assert _rel(got[k], getattr(m, k).weight.grad) < 1e-4, (k, _rel(got[k], getattr(m, k).weight.grad))
        assert _rel(gotb[k], getattr(m, k).bias.grad) < 1e-4, (k, "bias")
    gx = xr.grad * (x > 0)
    assert float((g_top - gx).abs().max()) <= 2.0 ** -8 * float(gx.abs().max()) + 1e-12, "g[L] more than bf16 rounding off"
    assert float(g_top.abs().max()) > 0 and float((g_top != 0).float().mean()) > 0.05


# ((128, 6, 1024) = the net and batch bench.py's `secondary.train_step` times)
@pytest.mark.parametrize("C,NB,n", [(64, 2, 64), (128, 3, 32), (128, 6, 1024)])
def test_whole_step_on_the_kernels_vs_torch_autograd_fp32(C, NB, n):
    """StepPlan.grads -- stem, tower, heads, losses and every gradient on the kernels -- against autograd through the
    plain fp32 torch forward of the same module (PolicyValueNet.forward without a plan) on the same batch: the losses
    within 2 % (bf16 activations through the tower), every parameter's gradient within bf16 tolerance of autograd's
    (cosine > 0.99; > 0.97 for the stem, whose gradient has passed every ReLU boundary of the tower).  This is the
    wiring test -- each kernel's arithmetic is pinned much tighter by the tests above on equal inputs."""
    import torch.nn.functional as F
    m, plan, own, opp, pi, z = _net_case(C, NB, n, 23)
    losses = plan.grads(own, opp, pi, z).clone()
    got = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert set(got) == {k for k, _ in m.named_parameters()} and all(bool(torch.isfinite(g).all()) for g in got.values())
    for p in m.parameters():
        p.grad = None
    logits, v = m(_planes(own, opp))
    ce = -(pi * F.log_softmax(logits, dim=1)).sum(1).mean()
    mse = F.mse_loss(v, z.float())
    (ce + mse).backward()
    want = [float(ce + mse), float(ce), float(mse)]
    assert np.allclose(losses[:3].cpu().numpy(), want, rtol=2e-2, atol=2e-3), (losses, want)
    assert float(losses[3]) == 0.0   # the error word: no row index was out of range
    cos = {k: float(F.cosine_similarity(got[k].flatten(), p.grad.flatten(), dim=0)) for k, p in m.named_parameters() if p.numel() > 1}
    mag = {k: float(got[k].norm() / p.grad.norm().clamp(min=1e-20)) for k, p in m.named_parameters()}
    print("cosine of kernel vs autograd gradients:", {k: round(c, 4) for k, c in cos.items()})
    for k, c in cos.items():
        assert c > (0.97 if k.startswith("stem") else 0.99), (k, c)
    # magnitudes: tensors with >= 64 elements (the 1-, 2- and 3-element bias gradients of the 1x1 convolutions are sums with
    # heavy cancellation over cells -- 15 % off in norm at batch 64 from bf16 activations alone; the heads test above pins
    # them to 1e-4 on equal inputs)
    assert all(0.9 < r < 1.1 for k, r in mag.items() if got[k].numel() >= 64), mag
    with pytest.raises(AssertionError):
        plan.grads(own, opp, pi, z)      # ... and StepPlan notices that its gradient tensors were swapped out


def test_step_gathers_its_batch_rows_itself_and_flags_bad_indices():
    """the kernels read the batch through the device descriptor: rows idx of a larger data set give bit for bit the
    losses and gradients of the same rows handed over contiguously; an index outside the data set does not fault (the
    kernels read the last / first row in its place) but is an ERROR: the step's error word (losses[3], bz_abi.h) counts
    every such batch position, stays set over later (clean) steps until it is read, and check_rows() raises IndexError;
    pointing the plan at another data set changes nothing but the 48 bytes."""
    n, rows = 32, 304
    m, plan, own, opp, pi, z = _net_case(64, 1, rows, 24)
    from betazero_amd.train_kernels import StepPlan
    plan = StepPlan(m, n)
    gen = torch.Generator(device=DEV).manual_seed(1)
    idx = torch.randint(0, rows, (n,), device=DEV, generator=gen)
    a = plan.grads(own, opp, pi, z, idx).clone()
    ga = {k: p.grad.clone() for k, p in m.named_parameters()}
    b = plan.grads(own[idx].contiguous(), opp[idx].contiguous(), pi[idx].contiguous(), z[idx].contiguous()).clone()
    assert torch.equal(a, b) and all(torch.equal(ga[k], p.grad) for k, p in m.named_parameters())
    assert float(a[3]) == 0.0 and plan.bad_rows() == 0
    plan.check_rows()                                                 # nothing to complain about
    bad = idx.clone()
    bad[3], bad[7], bad[20] = rows + 1000, -5, rows                   # (rows itself is one past the end)
    good = idx.clone()
    good[3], good[7], good[20] = rows - 1, 0, rows - 1
    c = plan.grads(own, opp, pi, z, bad).clone()
    assert float(c[3]) == 3.0                                         # three batch positions were out of range
    d = plan.grads(own, opp, pi, z, good).clone()
    assert torch.equal(c[:3], d[:3]) and not torch.equal(c[:3], a[:3])   # clamped, not a fault ...
    assert float(d[3]) == 3.0                                         # ... and the word is sticky over a clean step
    c2 = plan.grads(own, opp, pi, z, bad).clone()
    assert float(c2[3]) == 6.0                                        # it counts
    with pytest.raises(IndexError, match="6 batch position"):
        plan.check_rows()
    assert plan.bad_rows() == 0                                       # reading resets it
    a = a[:3]
    # another data set (new tensors, new addresses): only the descriptor changes -- a captured graph would keep working
    own2, opp2, pi2, z2 = (t.roll(7, 0).contiguous() for t in (own, opp, pi, z))
    e = plan.grads(own2, opp2, pi2, z2, idx).clone()
    f = plan.grads(own, opp, pi, z, (idx - 7) % rows).clone()      # the same rows of the original tensors
    e, f = e[:3], f[:3]
    assert torch.equal(e, f) and not torch.equal(e, a)
    graph = torch.cuda.CUDAGraph()
    plan.set_batch(own, opp, pi, z, idx)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.launch()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        plan.launch()
    graph.replay()
    assert torch.equal(plan.losses[:3], a)
    plan.set_batch(own2, opp2, pi2, z2, idx)                       # ... and it does: same graph, other data set
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(plan.losses[:3], e)
    plan.set_batch(own, opp, pi, z, bad)                           # the captured step flags bad indices too
    graph.replay()
    graph.replay()
    assert plan.bad_rows() == 6
    with pytest.raises(ValueError):
        plan.set_batch(own, opp, pi, z, idx[:8])              # not a whole batch of indices
    with pytest.raises(ValueError):
        plan.set_batch(own[:8], opp[:8], pi[:8], z[:8])       # fewer rows than the batch, no index
    with pytest.raises(ValueError, match="empty"):
        plan.set_batch(own[:0], opp[:0], pi[:0], z[:0], idx)  # an empty data set has no row to clamp to


@pytest.mark.parametrize("warmup", [0, 4])
def test_adam_kernel_equals_torch_adam(warmup):
    """6 steps with the update applied by k_train_adam behind k_train_finish (step count and learning-rate warm-up kept on the device)
    against torch.optim.Adam fed the SAME gradients (the ones the kernel also leaves in .grad; its rate set by hand to
    lr * min(1, t / warmup)): every parameter equal to 1e-5 of an lr-sized step plus a few fp32 ulps after each step (fp32
    arithmetic in a different order), the device counter at 6.  (Letting the second model compute its own gradients would test chaos, not
    Adam: a 1e-9 difference in a weight flips bf16 roundings in the tower and moves single updates by percents of lr.)"""
    import copy
    n = 64
    m1, plan1, own, opp, pi, z = _net_case(64, 2, n, 25)
    m0 = copy.deepcopy(m1)
    m2 = copy.deepcopy(m1)
    for p in m2.parameters():
        p.grad = None
    lr = 3e-3
    plan1.enable_adam(lr, warmup_steps=warmup)
    opt = torch.optim.Adam(m2.parameters(), lr=lr)
    plan1.set_batch(own, opp, pi, z)
    for t in range(1, 7):
        plan1.step()
        for g in opt.param_groups:
            g["lr"] = lr * min(1.0, t / warmup) if warmup else lr
        for a, b in zip(m1.parameters(), m2.parameters()):
            b.grad = a.grad.clone()
        opt.step()
        for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
            assert float((a - b).abs().max()) <= 1e-5 * lr * t + 5e-7 * float(b.abs().max()), (t, k, float((a - b).abs().max()))   # a few fp32 ulps of the weight
    assert plan1.adam_t == 6
    moved = max(float((a - b).abs().max()) for a, b in zip(m1.parameters(), m0.parameters()))
    assert moved > 1e-3          # ... and the parameters did move
    plan1.reset_adam(steps_done=0)
    assert plan1.adam_t == 0 and all(float(v.abs().max()) == 0.0 for v in plan1.adam_v.values())
