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


@pytest.mark.parametrize("C,L,n", [(64, 4, 16), (128, 2, 8), (128, 6, 12), (64, 8, 40)])
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


@pytest.mark.parametrize("C,L,n", [(64, 4, 16), (128, 4, 32), (64, 8, 64), (128, 12, 64)])
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


def test_graphed_train_step_on_the_tower_kernels_tracks_the_autograd_step():
    """GraphedTrainStep with the residual tower on the HIP kernels (forward, backward-data, backward-weights inside
    the captured graph; stem, heads, losses and Adam in torch) against train_step through stock autograd (MIOpen) from
    the same weights on the same batches: the same loss step by step within bf16 tolerance, and after 8 steps the
    tower's weights have moved the same way (cosine of the two weight updates > 0.9; Adam's first steps are sign-like,
    so tiny gradients may disagree in sign)."""
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
    g = GraphedTrainStep(m1, lr=1e-3, batch=128)
    assert g.plan is not None  # the kernels are in the graph
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
