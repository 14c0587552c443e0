#!/usr/bin/env python3
"""Why did channels-last training diverge in round 3 (profiles/r03_az_loop_channels_last_failure.txt: value loss stuck
at 0.95 in one run, non-finite weights in two)?  From the evidence that can be regenerated: ONE step's gradients of the
az_loop net (64 channels, 4 blocks, batch 1024) through stock autograd in four forms -- fp32 NCHW (the reference), bf16
autocast NCHW, bf16 autocast channels-last eager, bf16 autocast channels-last inside a captured HIP graph -- compared
parameter by parameter (max |g - g_ref| / max |g_ref|, cosine), then 60 steps of each bf16 form from the same weights
on the same batches with the per-step weight norm and loss.  python tools/exp_channels_last.py"""
import copy
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd.net import PolicyValueNet  # noqa: E402
from betazero_amd.train import planes_from_bits  # noqa: E402

torch.manual_seed(0)
rng = np.random.default_rng(0)
B, n = 1024, 8192
a = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
b = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
own = torch.as_tensor((a & ~b).view(np.int64)).cuda()
opp = torch.as_tensor((b & ~a).view(np.int64)).cuda()
pi = torch.rand((n, 65), device="cuda"); pi /= pi.sum(1, keepdim=True)
z = torch.randint(-1, 2, (n,), device="cuda").float()
base = PolicyValueNet(64, 4, 64).cuda()


def loss_of(m, idx, autocast, cl):
    x = planes_from_bits(own[idx], opp[idx])
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        lg, v = m(x)
    ce = -(pi[idx] * F.log_softmax(lg.float(), dim=1)).sum(1).mean()
    return ce + F.mse_loss(v.float(), z[idx]), ce


def grads(autocast, cl):
    m = copy.deepcopy(base)
    if cl:
        m = m.to(memory_format=torch.channels_last)
    idx = torch.arange(B, device="cuda")
    loss, _ = loss_of(m, idx, autocast, cl)
    loss.backward()
    return {k: p.grad.detach().float().clone() for k, p in m.named_parameters()}, float(loss)


ref, l_ref = grads(False, False)
print(f"one step, batch {B}: loss fp32 NCHW {l_ref:.5f}")
for label, ac, cl in (("bf16 autocast NCHW", True, False), ("bf16 autocast channels-last", True, True), ("fp32 channels-last", False, True)):
    g, l = grads(ac, cl)
    worst = max(((float((g[k] - ref[k]).abs().max() / ref[k].abs().max()), k) for k in ref), key=lambda t: t[0])
    cos = min((float(F.cosine_similarity(g[k].flatten(), ref[k].flatten(), dim=0)), k) for k in ref)
    nonfinite = [k for k in g if not bool(torch.isfinite(g[k]).all())]
    print(f"  {label:30s} loss {l:.5f}  worst max-rel gradient error {worst[0]:.4f} ({worst[1]})  lowest cosine {cos[0]:.5f} ({cos[1]})  non-finite: {nonfinite}")


def run(label, cl, graph, steps=60, lr=2e-3):
    m = copy.deepcopy(base)
    if cl:
        m = m.to(memory_format=torch.channels_last)
    opt = torch.optim.Adam(m.parameters(), lr=lr, capturable=graph)
    gen = torch.Generator(device="cuda").manual_seed(1)
    idxs = [torch.randint(0, n, (B,), device="cuda", generator=gen) for _ in range(steps)]
    sidx = idxs[0].clone()
    out = []

    def step():
        loss, ce = loss_of(m, sidx, True, cl)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return torch.stack([loss.detach(), ce.detach()])
    g = None
    if graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        lrs = [pg["lr"] for pg in opt.param_groups]
        for pg in opt.param_groups:
            pg["lr"] = 0.0
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        for pg, l in zip(opt.param_groups, lrs):
            pg["lr"] = l
        for st in opt.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            res = step()
    for i in range(steps):
        sidx.copy_(idxs[i])
        if g is not None:
            g.replay()
            r = res.clone()
        else:
            r = step()
        wn = torch.sqrt(sum((p.detach().float() ** 2).sum() for p in m.parameters()))
        out.append((float(r[0]), float(r[0] - r[1]), float(wn)))
    fin = all(bool(torch.isfinite(p).all()) for p in m.parameters())
    print(f"  {label:44s} loss / value MSE / |w| at steps 1, 10, 30, 60: " +
          "  ".join(f"{out[i][0]:.4f} / {out[i][1]:.4f} / {out[i][2]:.3f}" for i in (0, 9, 29, 59)) + f"  finite weights: {fin}")


print("60 Adam steps (lr 2e-3, the az_loop setting) from the same weights on the same batches:")
run("bf16 NCHW eager", False, False)
run("bf16 NCHW graph", False, True)
run("bf16 channels-last eager", True, False)
run("bf16 channels-last graph", True, True)


def closed_loop(iters=1, seed=0, layout="channels_last", warmup=0, kernels=False):
    """the round-3 failure was in the CLOSED LOOP (tools/az_loop.py, ~2400 steps per iteration, weights pushed into the
    engine between iterations): the same loop here with the channels-last graphed step, checking after every 100 steps
    which parameter (if any) has left the finite numbers, and the value loss per iteration."""
    from betazero_amd.augment import augment_examples
    from betazero_amd.engine import PipelinedSelfPlay, concat_device_examples
    from betazero_amd.net import DeviceNet
    from betazero_amd.train import GraphedTrainStep, refresh_device_net

    class ChannelsLastStep(GraphedTrainStep):
        def __init__(self, module, **kw):
            super().__init__(module, **kw)
            if layout == "channels_last":
                self.module.to(memory_format=torch.channels_last)

        def _step(self):
            x = planes_from_bits(self.own, self.opp)
            if layout == "channels_last":
                x = x.contiguous(memory_format=torch.channels_last)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.autocast):
                logits, v = self.module(x)
            logits, v = logits.float(), v.float()
            ce = -(self.pi * F.log_softmax(logits, dim=1)).sum(1).mean()
            mse = F.mse_loss(v, self.z.to(torch.float32))
            loss = ce + mse
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            # the largest |gradient| of every parameter tensor, BEFORE the optimiser consumes it (part of the captured step)
            self.gmax = torch.stack([p.grad.detach().float().abs().max() for p in self.module.parameters()])
            self.optimizer.step()
            return torch.cat([torch.stack([loss.detach(), ce.detach(), mse.detach()]), self.gmax])

    torch.manual_seed(seed)
    gen = torch.Generator(device="cuda:0").manual_seed(seed)
    module = PolicyValueNet(64, 4, 64, fused_tower=kernels)
    # the autograd forms of the step, CAPTURED (the subject of this experiment; GraphedTrainStep's defaults are now the
    # all-kernel step with the Adam kernel, and autograd steps run eagerly unless capture_autograd=True)
    step = (GraphedTrainStep if kernels else ChannelsLastStep)(module, lr=2e-3, batch=1024, lr_warmup_steps=warmup,
                                                              step_kernels=False, fused_adam=False, capture_autograd=True)
    if kernels:  # the product step has no gradient-maximum tail: wrap it
        inner = step._step

        def with_gmax():
            x = planes_from_bits(step.own, step.opp)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                logits, v = module(x, plan=step.plan)
            logits, v = logits.float(), v.float()
            ce = -(step.pi * F.log_softmax(logits, dim=1)).sum(1).mean()
            mse = F.mse_loss(v, step.z.to(torch.float32))
            loss = ce + mse
            step.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            gmax = torch.stack([p.grad.detach().float().abs().max() for p in module.parameters()])
            step.optimizer.step()
            return torch.cat([torch.stack([loss.detach(), ce.detach(), mse.detach()]), gmax])
        step._step = with_gmax
        del inner
    dnet = DeviceNet.from_module(copy.deepcopy(module).cpu().round_to_bf16_(), 2048)
    window = []
    for it in range(1, iters + 1):
        sp = PipelinedSelfPlay("reversi", 2048, 64, "net_bf16", dnet, pipelines=2, temp_moves=10, openings=1, seed=seed * 1000 + it,
                               dirichlet_alpha=0.3, dirichlet_eps=0.25)
        sp.run_iteration()
        aug = augment_examples(sp.device_examples(), dedupe=True)
        del sp
        window = (window + [aug])[-3:]
        data = concat_device_examples(window)
        steps = max(1, len(data) // 1024)
        losses, bad = [], None
        for k in range(steps):
            idx = torch.randint(0, len(data), (1024,), device="cuda:0", generator=gen)
            losses.append(step(data, idx))
            if bad is None and (k % 100 == 99 or k == steps - 1):
                for name, p in module.named_parameters():
                    if not bool(torch.isfinite(p).all()):
                        bad = (k, name)
                        break
        full = torch.stack(losses).cpu().numpy()
        ls, gm = full[:, :3], full[:, 3:]
        names = [k for k, _ in module.named_parameters()]
        med = np.nanmedian(gm, axis=0)
        with np.errstate(invalid="ignore"):
            odd = np.argwhere(~np.isfinite(gm) | (gm > 1e3 * med[None, :]))
        if len(odd):
            k0 = int(odd[:, 0].min())
            who = [(names[j], float(gm[k0, j]), float(med[j])) for j in odd[odd[:, 0] == k0][:, 1]]
            print(f"    first step with a non-finite or > 1000 x median gradient maximum: step {k0}: (parameter, max |g| at that step, its median over the iteration) = {who[:6]}")
            print(f"    loss at steps {k0 - 2} .. {k0 + 2}: {ls[max(0, k0 - 2):k0 + 3].round(4).tolist()}")
        else:
            print("    no gradient maximum is non-finite or above 1000 x its median in this iteration")
        zero_frac = (gm == 0).mean(axis=0)
        print("    fraction of steps in which a parameter's whole gradient is exactly 0: " +
              ", ".join(f"{nm} {zf:.2f}" for nm, zf in zip(names, zero_frac)))
        firstz = {nm: int(np.argmax(gm[:, j] == 0)) if (gm[:, j] == 0).any() else None for j, nm in enumerate(names)}
        print("    first step with an all-zero gradient:", {k: v for k, v in firstz.items() if v is not None})
        # is the net really dead at this point (an EAGER forward / backward on a batch says so), or does only the replay see zeros?
        m2 = copy.deepcopy(module)
        idx = torch.randint(0, len(data), (1024,), device="cuda:0", generator=gen)
        x = planes_from_bits(data.own[idx], data.opp[idx])
        with torch.autocast("cuda", dtype=torch.bfloat16):
            st = F.relu(m2.stem(x))
            lg, v = m2(x)
        l2 = -(data.pi[idx] * F.log_softmax(lg.float(), dim=1)).sum(1).mean() + F.mse_loss(v.float(), data.z[idx].float())
        l2.backward()
        eg = {k: float(p.grad.abs().max()) for k, p in m2.named_parameters()}
        print(f"    eager check on the weights after this iteration: stem output zeros {float((st == 0).float().mean()):.3f}, value range "
              f"[{float(v.min()):.3f}, {float(v.max()):.3f}], eager gradient maxima: " + ", ".join(f"{k} {g:.2e}" for k, g in eg.items()))
        print(f"  iteration {it}: {steps} steps, loss first / last tenth {ls[:steps // 10].mean(0).round(4).tolist()} / "
              f"{ls[-(steps // 10):].mean(0).round(4).tolist()}, losses finite: {bool(np.isfinite(ls).all())}, first non-finite parameter: {bad}", flush=True)
        if bad is not None:
            k = bad[0]
            print(f"    losses around step {k}: {ls[max(0, k - 102):k + 1:17].round(3).tolist()}")
            return
        refresh_device_net(dnet, module)


if "--loop" in sys.argv:
    for layout in ("channels_last", "nchw"):
        for s in (0, 1, 2, 3):
            print(f"closed loop, first iteration, graphed stock-autograd step, layout {layout}, seed {s}:")
            closed_loop(seed=s, layout=layout)
if "--warmup" in sys.argv:
    for wu in ((300,) if "--only300" in sys.argv else (0, 300)):
        for s in (0, 1, 2, 3):
            print(f"closed loop, first iteration, the product step (HIP tower kernels, NHWC), lr warm-up {wu} steps, seed {s}:")
            closed_loop(seed=s, layout="nchw", warmup=wu, kernels=True)
