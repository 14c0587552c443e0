"""The residual tower's training step on the hand-written HIP kernels of csrc/bz_train.hip (bz_train_* in
include/bz_abi.h): forward with saved activations, backward-data, backward-weights -- as ONE torch.autograd.Function,
so that stem, heads, losses and the optimiser stay plain torch (the loop shape of src/tic_tac_toe/SL/train.py:85-136).

    plan = TowerPlan(C, n_layers, batch)                    # every device buffer of one training shape, allocated once
    y = tower_apply(x0, W, b, plan)                         # x0 [n, C, 8, 8] = relu(stem(planes)); W [L, C, C, 3, 3]; b [L, C]
    loss(y).backward()                                      # d/dx0, d/dW, d/db through the kernels

bf16 activations and gradients, fp32 accumulation, fp32 master weights (the kernels repack W into their bf16 fragment
streams every step -- 2 x 147 k values per layer at C = 128).  There is no CPU path: without the HIP library or a GPU
this raises.  PolicyValueNet uses these kernels when its forward() is handed a TowerPlan (fused_tower form); its plain
forward() is the torch definition of the same net (CPU tests, the fp32 reference of the GPU tests).

StepPlan goes one further: the WHOLE step's forward, losses and backward on kernels (csrc/bz_train_ends.hip adds the stem,
the heads with their losses, the reduction of every partial sum into the parameters' .grad tensors, and Adam) -- 9
launches for the gradients, a tenth for the update, no autograd, no torch kernel; the batch's rows are gathered by the
kernels themselves.  That is what GraphedTrainStep captures by default."""
import ctypes as ct

import torch

from . import _lib


class TowerPlan:
    """device buffers of one (channels, conv layers, batch) training shape:
    acts / gs [L + 1, n, 64, C] bf16 (act[a], g[a] of bz_abi.h), the ReLU bit masks, the two weight-fragment streams,
    the weight-gradient partials and C zeros.  Static addresses: a step can be captured into a HIP graph."""

    def __init__(self, channels, n_layers, batch, device="cuda:0"):
        _lib.require_gpu()
        L = _lib.lib()
        self.C, self.L, self.n, self.device = channels, n_layers, batch, torch.device(device)
        P = L.bz_train_positions_per_workgroup(channels)
        if P <= 0 or n_layers < 2 or n_layers % 2:
            raise ValueError("the training kernels are built for 64 or 128 channels and an even number of conv layers")
        if batch % P:
            raise ValueError(f"batch must be a multiple of {P} at {channels} channels (positions resident per workgroup)")
        dev = self.device
        self.acts = torch.zeros((n_layers + 1, batch, 64, channels), dtype=torch.bfloat16, device=dev)
        self.gs = torch.zeros((n_layers + 1, batch, 64, channels), dtype=torch.bfloat16, device=dev)
        self.masks = torch.zeros(L.bz_train_mask_bytes(channels, n_layers, batch), dtype=torch.uint8, device=dev)
        nb = L.bz_train_wf_bytes(channels, n_layers)
        self.wf_fwd = torch.zeros(nb, dtype=torch.uint8, device=dev)   # (the 2 taps of padding behind the stream stay zero)
        self.wf_bwd = torch.zeros(nb, dtype=torch.uint8, device=dev)
        self.splits = L.bz_train_wgrad_splits(channels, n_layers, batch)
        self.partial = torch.zeros((n_layers, self.splits, 9, channels, channels), dtype=torch.float32, device=dev)
        self.db_partial = torch.zeros((n_layers, L.bz_train_wgrad_bias_rows(channels, self.splits), channels), dtype=torch.float32, device=dev)
        self.zeros_c = torch.zeros(channels, dtype=torch.float32, device=dev)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream


class _TowerFn(torch.autograd.Function):
    """x0 / the result / their gradients are [n, C, 8, 8] (nhwc=False: what a torch conv net passes around) or
    [n, 64, C] (nhwc=True: the kernels' own layout -- no permute copies at either end)"""

    @staticmethod
    def forward(ctx, x0, W, b, plan, nhwc):
        L, p = _lib.lib(), plan
        n, C, Ly = p.n, p.C, p.L
        assert x0.shape == ((n, 64, C) if nhwc else (n, C, 8, 8)), "input shape does not match the plan"
        assert W.shape == (Ly, C, C, 3, 3) and b.shape == (Ly, C), "weight shapes do not match the plan"
        Wc, bc = W.detach().contiguous().float(), b.detach().contiguous().float()
        with torch.cuda.device(p.device):
            p.acts[0].copy_(x0.detach() if nhwc else x0.detach().permute(0, 2, 3, 1).reshape(n, 64, C))
            _lib.check(L.bz_train_pack_weights(Wc.data_ptr(), C, Ly, p.wf_fwd.data_ptr(), p.wf_bwd.data_ptr(), p._stream()))
            _lib.check(L.bz_train_tower_fwd(p.acts[0].data_ptr(), p.wf_fwd.data_ptr(), bc.data_ptr(), C, Ly, n,
                                            p.acts[1].data_ptr(), p.masks.data_ptr(), p._stream()))
        ctx.plan, ctx.nhwc, ctx.in_dtype = p, nhwc, x0.dtype
        y = p.acts[Ly]
        return y.to(x0.dtype, copy=True) if nhwc else y.view(n, 8, 8, C).permute(0, 3, 1, 2).to(x0.dtype, copy=True)

    @staticmethod
    def backward(ctx, gy):
        L, p = _lib.lib(), ctx.plan
        n, C, Ly = p.n, p.C, p.L
        with torch.cuda.device(p.device):
            # g[L] = d loss / d (pre-activation of act[L]): the incoming gradient times the last layer's ReLU pattern
            gyn = gy if ctx.nhwc else gy.permute(0, 2, 3, 1).reshape(n, 64, C)
            torch.mul(gyn, p.acts[Ly] > 0, out=p.gs[Ly])
            _lib.check(L.bz_train_tower_bwd(p.gs[Ly].data_ptr(), p.wf_bwd.data_ptr(), p.zeros_c.data_ptr(), p.masks.data_ptr(),
                                            C, Ly, n, p.gs[0].data_ptr(), p._stream()))
            _lib.check(L.bz_train_wgrad(p.acts[0].data_ptr(), p.gs[1].data_ptr(), C, Ly, n, p.splits, p.partial.data_ptr(),
                                        p.db_partial.data_ptr(), p._stream()))
        # sum over the batch slices, then [L, tap, ci, co] -> torch's [L, co, ci, 3, 3]
        dW = p.partial.sum(1).view(Ly, 3, 3, C, C).permute(0, 4, 3, 1, 2).contiguous()
        db = p.db_partial.sum(1)
        g0 = p.gs[0]
        gx0 = g0.to(ctx.in_dtype, copy=True) if ctx.nhwc else g0.view(n, 8, 8, C).permute(0, 3, 1, 2).to(ctx.in_dtype, copy=True)
        return gx0, dW, db, None, None


class _RowsLinear(torch.autograd.Function):
    """y = x @ W.T + b for x [rows, K] with MANY rows (batch x 64 cells) and small K / N (the stem: K = 32, the head
    convolutions: N = 32).  The forward and d/dx are ordinary GEMMs; d/dW = dy.T @ x is a [N x rows] x [rows x K]
    product whose long axis is the reduction: the BLAS picks a kernel without a K split for it (measured 174-239 us for
    a 3 x 64 result), so it is done as a batched GEMM over 64 row chunks plus a sum (~10 us)."""

    @staticmethod
    def forward(ctx, x, W, b):
        # under autocast the product runs in bf16 (this Function is opaque to autocast, so it casts itself)
        dt = torch.bfloat16 if (x.is_cuda and torch.is_autocast_enabled()) else x.dtype
        xb, Wb = x.to(dt), W.to(dt)
        ctx.save_for_backward(xb, Wb)
        ctx.dts = (x.dtype, W.dtype, b.dtype)
        with torch.autocast("cuda", enabled=False):
            return torch.addmm(b.to(dt), xb, Wb.t())

    @staticmethod
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        gy = gy.to(x.dtype)
        gx = (gy @ W).to(ctx.dts[0]) if ctx.needs_input_grad[0] else None
        rows = x.shape[0]
        ch = 64 if rows % 64 == 0 else 1
        gyc = gy.view(ch, rows // ch, -1).transpose(1, 2)
        gW = torch.bmm(gyc, x.view(ch, rows // ch, -1)).sum(0, dtype=torch.float32)
        # the bias gradient the same way (x a column of ones) rather than as gy.sum(0): a 65,536-row column sum is a
        # multi-block torch reduction, and inside replayed HIP graphs such sums came back wrong once in a few hundred
        # steps on this stack (all zeros here; 6e32 in the stock channels-last path -- profiles/r04_channels_last_cause.txt)
        # (the cause, found in round 5: that reduction zeroes its semaphores with a captured cudaMemsetAsync, and captured memset
        # nodes do not reliably execute on replay on this stack -- profiles/r05_graph_memset_node.txt; GraphedTrainStep no
        # longer captures the autograd forms of the step unless asked to, train.py)
        gb = torch.bmm(gyc, gy.new_ones((ch, rows // ch, 1))).sum(0, dtype=torch.float32).squeeze(1)
        return gx, gW.to(ctx.dts[1]), gb.to(ctx.dts[2])


class StepPlan(TowerPlan):
    """forward + losses + backward (+ Adam) of one training step of `module` (PolicyValueNet(fused_tower=True), on the
    device) at a fixed batch size, entirely on the HIP kernels:

        plan = StepPlan(module, batch)
        plan.set_batch(own, opp, pi, z, idx)     # the data set's tensors and the batch's row indices (idx None: rows 0..batch-1)
        losses = plan.grads()                    # [loss, policy CE, value MSE, error word] (device, static); every p.grad is set
        optimizer.step()                         # torch's -- or instead of the two lines above:
        plan.enable_adam(lr); losses = plan.step()    # + the Adam update as a tenth launch (torch.optim.Adam's arithmetic)

    own / opp: int64 tensors holding the uint64 bitboards [rows]; pi fp32 [rows, 65]; z int8 [rows]; idx int64 [batch].
    The kernels read the batch through a 48-byte descriptor in device memory (bz_train_batch) and gather the rows
    themselves: a captured step keeps working when set_batch() points it at new tensors -- nothing is copied.  The
    gradient tensors are allocated once and installed as the parameters' .grad (static addresses: the step can be
    captured into a HIP graph; do not call zero_grad(set_to_none=True) in between -- every launch overwrites them)."""

    NAMES = ("stem_w", "stem_b", "tower_w", "tower_b", "pol_w", "pol_b", "polfc_w", "polfc_b", "val_w", "val_b", "v1_w", "v1_b", "v2_w", "v2_b")

    def __init__(self, module, batch, device="cuda:0"):
        if not getattr(module, "fused_tower", False):
            raise ValueError("StepPlan needs PolicyValueNet(..., fused_tower=True)")
        if module.VH > 64:
            raise ValueError("the head kernels hold the value head's hidden layer in one wavefront: value_hidden <= 64")
        super().__init__(module.C, 2 * module.NB, batch, device)
        L, dev = _lib.lib(), self.device
        self.module, self.VH = module, module.VH
        named = {"stem_w": module.stem.weight, "stem_b": module.stem.bias, "tower_w": module.tower_w, "tower_b": module.tower_b,
                 "pol_w": module.pol.weight, "pol_b": module.pol.bias, "polfc_w": module.polfc.weight, "polfc_b": module.polfc.bias,
                 "val_w": module.val.weight, "val_b": module.val.bias, "v1_w": module.v1.weight, "v1_b": module.v1.bias,
                 "v2_w": module.v2.weight, "v2_b": module.v2.bias}
        for k, p in named.items():
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError(f"StepPlan: parameter {k} must be a contiguous fp32 tensor on {dev}")
            p.grad = torch.zeros_like(p)
        self.params = named
        sizes = (ct.c_int32 * 6)()
        _lib.check(L.bz_train_ends_sizes(self.C, batch, sizes))
        f32 = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)  # noqa: E731
        self.stem_partial, self.heads_partial, self.heads_w_partial = f32(sizes[0], sizes[1]), f32(sizes[2], sizes[3]), f32(sizes[4], sizes[5])
        self.hv, self.dl, self.dv1 = f32(batch, 192), f32(batch, 65), f32(batch, 64)
        self.losses = f32(4)   # loss, CE, MSE and the error word: batch positions whose row index was out of range, summed over steps
        self._head = _lib.TrainHeadParams(**{k: named[k].data_ptr() for k, _ in _lib.TrainHeadParams._fields_})
        self._grads = _lib.TrainTensors(**{k: named[k].grad.data_ptr() for k in self.NAMES})
        self._partials = _lib.TrainPartials(tower=self.partial.data_ptr(), tower_b=self.db_partial.data_ptr(), stem=self.stem_partial.data_ptr(),
                                            heads=self.heads_partial.data_ptr(), heads_w=self.heads_w_partial.data_ptr(), splits=self.splits)
        self.batch_desc = torch.zeros(ct.sizeof(_lib.TrainBatch), dtype=torch.uint8, device=dev)
        self._batch_refs = None
        self._adam, self.adam_m, self.adam_v, self.hyper = None, None, None, None

    # ---- the batch
    def set_batch(self, own, opp, pi, z, idx=None):
        """point the step at rows `idx` (int64 [batch], on the device; None: rows 0..batch-1) of the data set (own, opp, pi,
        z).  Asynchronous on the current stream; the tensors are kept alive until the next call."""
        n, dev = self.n, self.device
        rows = int(own.shape[0])
        if rows < 1:
            raise ValueError("set_batch: the data set is empty")
        ok = (own.dtype == torch.int64 and opp.dtype == torch.int64 and own.shape == (rows,) and opp.shape == (rows,) and
              pi.dtype == torch.float32 and pi.shape == (rows, 65) and z.dtype == torch.int8 and z.shape == (rows,) and
              all(t.is_contiguous() and t.device == dev for t in (own, opp, pi, z)))
        if not ok:
            raise ValueError("set_batch: own / opp int64 [rows], pi fp32 [rows, 65], z int8 [rows], contiguous, on the plan's device")
        if idx is None:
            if rows < n:
                raise ValueError(f"set_batch: the data set has {rows} rows, the batch needs {n}")
        elif not (idx.dtype == torch.int64 and idx.shape == (n,) and idx.is_contiguous() and idx.device == dev):
            raise ValueError(f"set_batch: idx must be a contiguous int64 tensor of {n} row numbers on the plan's device")
        d = _lib.TrainBatch(own=own.data_ptr(), opp=opp.data_ptr(), pi=pi.data_ptr(), z=z.data_ptr(),
                            idx=idx.data_ptr() if idx is not None else None, n_rows=rows)
        key = bytes(d)
        if self._batch_refs is None or self._batch_refs[0] != key:   # (a pageable host copy: ordered on the current stream, done on return)
            with torch.cuda.device(dev):
                self.batch_desc.copy_(torch.frombuffer(bytearray(key), dtype=torch.uint8))
        self._batch_refs = (key, own, opp, pi, z, idx)

    # ---- the optimiser as the step's tenth launch
    def enable_adam(self, lr, betas=(0.9, 0.999), eps=1e-8, warmup_steps=0):
        """Adam (torch.optim.Adam's arithmetic, no weight decay / amsgrad: what the reference constructs, train.py:87) applied
        by k_train_adam right behind k_train_finish (one coalesced pass over all 14 tensors).  State: adam_m / adam_v (dicts of tensors like the parameters) and
        the device block `hyper` = {lr, steps done, warm-up steps, 0}: the kernel advances the step count itself and ramps
        the rate over the first warmup_steps steps (lr * min(1, t / warmup_steps)), so consecutive steps need no host write."""
        self.adam_m = {k: torch.zeros_like(p) for k, p in self.params.items()}
        self.adam_v = {k: torch.zeros_like(p) for k, p in self.params.items()}
        self.hyper = torch.zeros(4, dtype=torch.float32, device=self.device)
        T = _lib.TrainTensors
        self._adam = _lib.TrainAdam(hyper=self.hyper.data_ptr(), beta1=betas[0], beta2=betas[1], eps=eps,
                                    p=T(**{k: self.params[k].data_ptr() for k in self.NAMES}),
                                    m=T(**{k: self.adam_m[k].data_ptr() for k in self.NAMES}),
                                    v=T(**{k: self.adam_v[k].data_ptr() for k in self.NAMES}))
        self.reset_adam(lr, warmup_steps)

    def reset_adam(self, lr=None, warmup_steps=None, steps_done=0):
        """zero moments, step count `steps_done`, (new) rate / warm-up"""
        if lr is not None:
            self.lr = float(lr)
        if warmup_steps is not None:
            self.warmup = int(warmup_steps)
        for d in (self.adam_m, self.adam_v):
            for t in d.values():
                t.zero_()
        with torch.cuda.device(self.device):
            self.hyper.copy_(torch.tensor([self.lr, float(steps_done), float(self.warmup), 0.0], dtype=torch.float32))

    def set_lr(self, lr):
        """change the (peak) learning rate; moments and step count stay"""
        self.lr = float(lr)
        with torch.cuda.device(self.device):
            self.hyper[0:1].copy_(torch.tensor([self.lr], dtype=torch.float32))

    @property
    def adam_t(self):
        """steps done so far (reads the device counter: synchronises)"""
        return int(self.hyper[1].item()) if self.hyper is not None else 0

    # ---- the launches
    def launch(self, adam=False):
        """the 9 (adam: 10) launches of one step on the current stream (no host work besides: this is what a HIP graph captures)"""
        L, p, n, Cc, Ly = _lib.lib(), self.params, self.n, self.C, self.L
        if self._batch_refs is None:
            raise RuntimeError("StepPlan: set_batch() first")
        if adam and self._adam is None:
            raise RuntimeError("StepPlan: enable_adam() first")
        for k, t in p.items():   # (an optimiser that swapped a gradient tensor out would leave the kernels writing into a dead one)
            assert t.grad is not None and t.grad.data_ptr() == getattr(self._grads, k), f"the .grad of {k} was replaced"
        with torch.cuda.device(self.device):
            st, bd = self._stream(), self.batch_desc.data_ptr()
            chk = _lib.check
            chk(L.bz_train_stem_fwd(bd, n, p["stem_w"].data_ptr(), p["stem_b"].data_ptr(), Cc, self.acts[0].data_ptr(), st))
            chk(L.bz_train_pack_weights(p["tower_w"].data_ptr(), Cc, Ly, self.wf_fwd.data_ptr(), self.wf_bwd.data_ptr(), st))
            chk(L.bz_train_tower_fwd(self.acts[0].data_ptr(), self.wf_fwd.data_ptr(), p["tower_b"].data_ptr(), Cc, Ly, n, self.acts[1].data_ptr(),
                                     self.masks.data_ptr(), st))
            chk(L.bz_train_heads(self.acts[Ly].data_ptr(), bd, n, Cc, self.VH, ct.byref(self._head), self.gs[Ly].data_ptr(),
                                 self.hv.data_ptr(), self.dl.data_ptr(), self.dv1.data_ptr(), self.heads_partial.data_ptr(), st))
            chk(L.bz_train_tower_bwd(self.gs[Ly].data_ptr(), self.wf_bwd.data_ptr(), self.zeros_c.data_ptr(), self.masks.data_ptr(), Cc, Ly, n,
                                     self.gs[0].data_ptr(), st))
            chk(L.bz_train_wgrad(self.acts[0].data_ptr(), self.gs[1].data_ptr(), Cc, Ly, n, self.splits, self.partial.data_ptr(),
                                 self.db_partial.data_ptr(), st))
            chk(L.bz_train_stem_wgrad(bd, self.acts[0].data_ptr(), self.gs[0].data_ptr(), n, Cc, self.stem_partial.data_ptr(), st))
            chk(L.bz_train_heads_wgrad(self.hv.data_ptr(), self.dl.data_ptr(), self.dv1.data_ptr(), n, self.VH, self.heads_w_partial.data_ptr(), st))
            chk(L.bz_train_finish(ct.byref(self._partials), ct.byref(self._grads), Cc, Ly, self.VH, n, self.losses.data_ptr(),
                                  ct.byref(self._adam) if adam else None, st))
        return self.losses

    def bad_rows(self, reset=True):
        """the step's error word (bz_abi.h, bz_train_finish): how many batch positions since the last reset had a row index
        outside the data set -- the kernels clamped them, i.e. trained on row 0 / the last row instead.  Reads the device
        (synchronises); call it where the losses are looked at."""
        n = int(self.losses[3].item())
        if n and reset:
            self.losses[3:4].zero_()
        return n

    def check_rows(self):
        """raise IndexError if any step since the last check gathered a row outside the data set (what torch.index_select,
        the loop shape of SL/train.py:102-108, would have raised at once)"""
        n = self.bad_rows()
        if n:
            raise IndexError(f"StepPlan: {n} batch position(s) had a row index outside the data set since the last check; "
                             "the kernels clamp instead of faulting, so those steps trained on the wrong rows")

    def grads(self, own=None, opp=None, pi=None, z=None, idx=None):
        """forward, losses, backward: every parameter's .grad is set; returns the static [loss, CE, MSE, error word] tensor.
        (own, opp, pi, z[, idx]) given: set_batch() first."""
        if own is not None:
            self.set_batch(own, opp, pi, z, idx)
        return self.launch(adam=False)

    def step(self):
        """grads() and the Adam update: 10 launches (enable_adam() first)"""
        return self.launch(adam=True)


def rows_linear(x, W, b):
    return _RowsLinear.apply(x, W, b)


def tower_apply(x0, W, b, plan):
    """x0 [n, C, 8, 8] (the stem's output after its ReLU) -> the tower's output [n, C, 8, 8]; differentiable in x0, W, b"""
    return _TowerFn.apply(x0, W, b, plan, False)


def tower_apply_nhwc(x0, W, b, plan):
    """the same on [n, 64 cells, C] tensors -- the kernels' own layout (what PolicyValueNet's fused path passes)"""
    return _TowerFn.apply(x0, W, b, plan, True)
