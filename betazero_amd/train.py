"""Closing the AlphaZero loop: one optimisation step on (s, pi, z)
with stock PyTorch-ROCm autograd.  Loop shape after src/tic_tac_toe/SL/train.py:85-136
(cross-entropy on the policy, Adam lr 1e-4), plus the value MSE the reference lacks.

The data path is device-resident: DeviceExamples in (from gather_examples_device / augment_examples), batches are
index-selected on the GPU, the two input planes are expanded from the bitboards by torch ops on the GPU, forward and
backward run under bf16 autocast (fp32 master weights, fp32 losses).  Nothing is copied to the host; the three loss
values come back as device scalars (call .item() when you want to look at them, not every step)."""
import torch
import torch.nn.functional as F

from .engine import DeviceExamples, Examples

_SHIFTS = {}


def planes_from_bits(own, opp):
    """int64 tensors holding uint64 bitboards [n] -> float planes [n, 2, 8, 8] (bit 8*r+c), on their device.
    (>> on int64 is arithmetic, which still leaves bit s of the pattern at bit 0 for every s <= 63.)"""
    dev = own.device
    sh = _SHIFTS.get(dev)
    if sh is None:
        sh = _SHIFTS[dev] = torch.arange(64, dtype=torch.int64, device=dev)
    a = ((own[:, None] >> sh) & 1).to(torch.float32).view(-1, 8, 8)
    b = ((opp[:, None] >> sh) & 1).to(torch.float32).view(-1, 8, 8)
    return torch.stack([a, b], dim=1)


def make_optimizer(module, lr=1e-4):
    return torch.optim.Adam(module.parameters(), lr=lr)  # train.py:87,192


def train_step(module, optimizer, ex, idx=None, device="cuda:0", autocast=True):
    """one Adam step on the rows `idx` of ex (Reversi 8x8 net).  ex: DeviceExamples (stays on the GPU; idx a device
    index tensor or None = all rows) or host Examples (uploaded first).  Returns (loss, policy CE, value MSE) as
    detached device scalars."""
    if isinstance(ex, Examples):
        ex = DeviceExamples.from_host(ex, device)
    dev = ex.own.device
    if next(module.parameters()).device != dev:
        module.to(dev)
    module.train()
    if idx is None:
        own, opp, pi, z = ex.own, ex.opp, ex.pi, ex.z
    else:
        idx = torch.as_tensor(idx, device=dev)
        own, opp, pi, z = ex.own[idx], ex.opp[idx], ex.pi[idx], ex.z[idx]
    x = planes_from_bits(own, opp)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        logits, v = module(x)
    logits, v = logits.float(), v.float()
    ce = -(pi * F.log_softmax(logits, dim=1)).sum(1).mean()
    mse = F.mse_loss(v, z.to(torch.float32))
    loss = ce + mse
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    optimizer.step()
    return loss.detach(), ce.detach(), mse.detach()


def holdout_split(n_rows, val_fraction=0.2, generator=None, device="cuda:0"):
    """the reference's train / validation split (SL/train.py:66-76: val_size = int(total_size * validation_split), a
    random split of the ROWS) as two device index tensors (train_idx, val_idx) into a data set of n_rows rows"""
    perm = torch.randperm(int(n_rows), generator=generator, device=device)
    n_val = int(int(n_rows) * val_fraction)
    return perm[n_val:], perm[:n_val]


def select_rows(ex, idx):
    """DeviceExamples holding the rows `idx` (a device index tensor) of ex"""
    return DeviceExamples(own=ex.own[idx], opp=ex.opp[idx], pi=ex.pi[idx], z=ex.z[idx], mover=ex.mover[idx], act=ex.act[idx],
                          game=ex.game[idx], ply=ex.ply[idx], size=ex.size)


@torch.no_grad()
def validate(device_net, ex, idx=None):
    """The validation line of the reference's training loop (SL/train.py:121-146: loss and accuracy on the held-out rows
    after every epoch) for (s, pi, z) examples, on the net the SEARCH uses -- the engine's bf16 MFMA forward
    (DeviceNet.forward), i.e. on the weights as they will play: mean policy cross-entropy against pi, value MSE against z,
    and top-1 agreement (argmax of the logits == argmax of pi; the reference's accuracy is `predicted == actions.argmax`,
    :139-142).  Everything stays on the GPU; returns a dict of Python floats (one synchronisation)."""
    own, opp, pi, z = (ex.own, ex.opp, ex.pi, ex.z) if idx is None else (ex.own[idx], ex.opp[idx], ex.pi[idx], ex.z[idx])
    n = int(own.shape[0])
    if n == 0:
        return {"rows": 0, "policy_ce": None, "value_mse": None, "top1": None}
    ce = torch.zeros((), dtype=torch.float64, device=own.device)
    se, hit = torch.zeros_like(ce), torch.zeros_like(ce)
    for a in range(0, n, device_net.max_batch):
        b = min(n, a + device_net.max_batch)
        logits, v = device_net.forward(own[a:b].contiguous(), opp[a:b].contiguous())
        ce += -(pi[a:b] * F.log_softmax(logits, dim=1)).sum()
        se += ((v - z[a:b].to(torch.float32)) ** 2).sum()
        hit += (logits.argmax(1) == pi[a:b].argmax(1)).sum()
    ce, se, hit = (float(t) / n for t in (ce, se, hit))
    return {"rows": n, "loss": ce + se, "policy_ce": ce, "value_mse": se, "top1": hit}


class GraphedTrainStep:
    """train_step() with every buffer static and, for the all-kernel step, captured ONCE into a HIP graph and replayed (ten
    launches back to back instead of ten host calls).  Same arithmetic as train_step (same module, Adam, bf16 autocast),
    fixed batch size.

        step = GraphedTrainStep(module, lr=2e-3, batch=1024, na=65)
        loss, ce, mse = step(examples, idx)      # idx: device index tensor of exactly `batch` rows
        step.check()                             # where the losses are read: raises if an index was out of range

    The optimiser lives inside (the Adam kernel, or torch's Adam with capturable=True: its step counter is a device tensor).
    The forms of the step that go through torch autograd run eagerly unless capture_autograd=True (see __init__)."""

    def __init__(self, module, lr=1e-4, batch=1024, na=65, device="cuda:0", autocast=True, tower_kernels=None, lr_warmup_steps=0,
                 step_kernels=None, fused_adam=None, capture_autograd=False):
        # (NCHW on purpose: channels-last convolutions measured ~20 % faster per step in tools/bench_train.py, but the
        # closed loop then failed to learn the value head in one run and produced non-finite weights in two others --
        # profiles/r03_az_loop_channels_last_failure.txt -- so that layout is not offered)
        self.module, self.batch, self.dev, self.autocast = module.to(device), batch, torch.device(device), autocast
        # tower_kernels: run the residual tower's forward / backward on the hand-written HIP kernels (csrc/bz_train.hip)
        # instead of MIOpen; default: whenever the module keeps its tower stacked (PolicyValueNet(fused_tower=True))
        if tower_kernels is None:
            tower_kernels = bool(getattr(module, "fused_tower", False))
        # step_kernels: the WHOLE forward / losses / backward on the HIP kernels (train_kernels.StepPlan: 9 launches + Adam's, no
        # autograd -- stem, heads and losses too); default: whenever the tower kernels are on and the step runs in bf16.
        # step_kernels=False keeps stem / heads / losses in torch autograd around the tower kernels (the round-4 form).
        if step_kernels is None:
            step_kernels = tower_kernels and autocast and na == 65 and getattr(module, "VH", 65) <= 64
        self.plan, self.step_plan = None, None
        if tower_kernels or step_kernels:
            from .train_kernels import StepPlan, TowerPlan
            if not getattr(module, "fused_tower", False):
                raise ValueError("tower_kernels / step_kernels need PolicyValueNet(..., fused_tower=True)")
            if step_kernels:
                self.step_plan = self.plan = StepPlan(self.module, batch, device)
                self.idx = torch.zeros(batch, dtype=torch.int64, device=self.dev)   # the batch's rows: the kernels gather them themselves
            else:
                self.plan = TowerPlan(module.C, 2 * module.NB, batch, device)
        # lr_warmup_steps > 0: the learning rate ramps linearly from lr / warmup to lr over the first steps.  Adam's first
        # updates move every weight by ~lr whatever the gradient's size; at the loop demo's lr = 2e-3 that kills the
        # single-channel value head's ReLU (and in 2 of 4 seeds every head ReLU, i.e. the whole net) within the first
        # 1-40 steps, in either memory layout -- profiles/r04_channels_last_cause.txt.  The rate lives in a device
        # scalar (capturable Adam reads it inside the graph) that __call__ sets before each replay.
        self.lr, self.warmup, self.steps_done = float(lr), int(lr_warmup_steps), 0
        self.lr_t = torch.tensor(self._lr_at(0), dtype=torch.float32, device=self.dev)
        # fused_adam (default with step_kernels): the Adam update as the step's tenth kernel (StepPlan.enable_adam: step count
        # and warm-up live on the device; a step is then ONE graph launch and a copy of the batch's row numbers).  Otherwise torch's:
        # (fused: the whole Adam update of all parameter tensors is one kernel instead of ~10 multi-tensor ones)
        self.fused_adam = bool(self.step_plan is not None) if fused_adam is None else bool(fused_adam)
        if self.fused_adam:
            if self.step_plan is None:
                raise ValueError("fused_adam needs step_kernels")
            self.step_plan.enable_adam(self.lr, warmup_steps=self.warmup)
            self.optimizer = None
        else:
            self.optimizer = torch.optim.Adam(module.parameters(), lr=self.lr_t, capturable=True, fused=True)
        # capture_autograd: a step that goes through torch autograd (step_kernels=False, or a module without the fused tower) is
        # NOT captured into a HIP graph unless asked for -- it runs eagerly, same arithmetic.  Captured autograd steps returned
        # wrong gradients once in a few hundred replays on this stack (a bias gradient of 6e32 at replay 305 / 306, an all-zero
        # one at replay 702 / 703: profiles/r04_channels_last_cause.txt).  Cause (profiles/r05_graph_memset_node.txt,
        # tools/exp_graph_reduction.py): torch's multi-block reductions zero their semaphores with cudaMemsetAsync on the launch
        # stream -- a memset NODE under capture -- and a captured memset node does not reliably zero its buffer on replay here
        # (reproduced with the memset alone).  The all-kernel step (step_kernels) contains no memset node and no torch kernel
        # and is captured as before.
        self.capture = self.step_plan is not None or bool(capture_autograd)
        self.own = torch.zeros(batch, dtype=torch.int64, device=self.dev)
        self.opp = torch.zeros(batch, dtype=torch.int64, device=self.dev)
        self.pi = torch.full((batch, na), 1.0 / na, dtype=torch.float32, device=self.dev)
        self.z = torch.zeros(batch, dtype=torch.int8, device=self.dev)
        self.graph, self.out = None, None

    def _lr_at(self, k):
        return self.lr * min(1.0, (k + 1) / self.warmup) if self.warmup > 0 else self.lr

    def _step(self):
        if self.step_plan is not None:
            if self.fused_adam:
                return self.step_plan.step()           # 10 launches: gradients and the Adam update
            losses = self.step_plan.grads()            # sets every parameter's .grad
            self.optimizer.step()
            return losses
        x = planes_from_bits(self.own, self.opp)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.autocast):
            logits, v = self.module(x, plan=self.plan) if self.plan is not None else self.module(x)
        logits, v = logits.float(), v.float()
        ce = -(self.pi * F.log_softmax(logits, dim=1)).sum(1).mean()
        mse = F.mse_loss(v, self.z.to(torch.float32))
        loss = ce + mse
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        return torch.stack([loss.detach(), ce.detach(), mse.detach()])

    def _capture(self):
        """3 eager steps on a side stream (library handles, autotuning, optimiser state) on the batch __call__ has just
        placed in the static buffers, with lr scaled to 0 so that the weights do not move and the optimiser state zeroed
        afterwards; then the capture (which launches nothing: the first real step is the first replay)"""
        self.module.train()
        self.lr_t.zero_()
        if self.fused_adam:
            self.step_plan.set_lr(0.0)
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                self._step()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        if self.fused_adam:   # the warm-up must not count as steps / leave moments behind
            self.step_plan.reset_adam(self.lr, self.warmup, steps_done=self.steps_done)
        else:
            for st in self.optimizer.state.values():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        v.zero_()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._step()

    def __call__(self, ex, idx):
        assert idx.numel() == self.batch, "GraphedTrainStep replays a fixed batch size"
        if self.step_plan is not None:   # the kernels read rows idx of the data set themselves: one 8-byte-per-row copy, no gathers
            self.idx.copy_(idx)
            self.step_plan.set_batch(ex.own, ex.opp, ex.pi, ex.z, self.idx)
        else:
            torch.index_select(ex.own, 0, idx, out=self.own)
            torch.index_select(ex.opp, 0, idx, out=self.opp)
            torch.index_select(ex.pi, 0, idx, out=self.pi)
            torch.index_select(ex.z, 0, idx, out=self.z)
        if not self.capture:   # an autograd form of the step: eager (see capture_autograd in __init__)
            self.module.train()
            self.lr_t.fill_(self._lr_at(self.steps_done))
            out = self._step()
            self.steps_done += 1
            return out[:3].clone()
        if self.graph is None:
            self._capture()
        if not self.fused_adam:
            self.lr_t.fill_(self._lr_at(self.steps_done))
        self.graph.replay()
        self.steps_done += 1
        return self.out[:3].clone()

    def check(self):
        """raise IndexError if a step since the last check was handed a row index outside the data set (the all-kernel step
        clamps instead of faulting and counts the event in its error word; the autograd forms raise inside index_select).
        Synchronises: call it where the losses are read, not every step."""
        if self.step_plan is not None:
            self.step_plan.check_rows()


def refresh_device_net(device_net, module):
    """push the trained weights into the HIP engine's net (bf16-rounded copy for the MFMA path).  Raises
    FloatingPointError when a parameter is not finite (one isfinite reduction per refresh): the losses of a step are
    computed BEFORE its optimiser update, so a last step that overflows is invisible in them, and a net with such
    weights makes every search return garbage (the engine then raises ERR_EVAL_NONFINITE far from the cause)."""
    import copy
    bad = [n for n, p in module.named_parameters() if not bool(torch.isfinite(p.detach()).all())]
    if bad:
        raise FloatingPointError(f"refresh_device_net: non-finite values in {len(bad)} parameter tensor(s), first: {bad[0]}; "
                                 "the engine's net was NOT updated")
    m = copy.deepcopy(module).cpu()
    m.round_to_bf16_()
    device_net.update(m.flat_params())
