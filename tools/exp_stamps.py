import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd import _lib
from betazero_amd.net import DeviceNet, PolicyValueNet
B = 4096
FP8 = os.environ.get('FP8') == '1'
rng = np.random.default_rng(0)
x = rng.integers(0, 2**63, size=B, dtype=np.int64); y = rng.integers(0, 2**63, size=B, dtype=np.int64)
own = torch.as_tensor(x & ~y).cuda(); opp = torch.as_tensor(y & ~x).cuda()
torch.manual_seed(0)
mod = PolicyValueNet(128, 6, 64).round_to_bf16_()
if os.environ.get("ZERO") == "1":  # all-zero weights: the same instruction stream on operands that toggle nothing (low power)
    with torch.no_grad():
        for prm in mod.parameters(): prm.zero_()
net = DeviceNet.from_module(mod, B)
for _ in range(int(os.environ.get("WARM", "3000"))): net.forward(own, opp, fp8=FP8)  # >= 2 s of back-to-back launches before the stamps are read (DVFS settles)
torch.cuda.synchronize()
L = C.CDLL(_lib.SO)
nb = 1024 if os.environ.get("NB_WG") is None else int(os.environ["NB_WG"])
buf = np.zeros(max(nb, 2048 + 8) * 8, np.uint64)
L.bz_debug_read.argtypes = [C.c_void_p, C.c_int64]
assert L.bz_debug_read(buf.ctypes.data, buf.nbytes) == 0
d = buf[:nb * 8].reshape(nb, 8).astype(np.float64)
ch = buf[8 * 2048:8 * 2048 + 64].astype(np.float64)
print("per WG (wave 0), cycles:  kloop %.0f  epilogue %.0f  barrier %.0f   layers-total %.0f  kernel-total %.0f" % tuple(d[:, i].mean() for i in (0, 1, 2, 3, 4)))
clk = d[:, 4] / d[:, 5] * 100e6
print("in-kernel clock GHz: mean %.3f min %.3f max %.3f" % (clk.mean() / 1e9, clk.min() / 1e9, clk.max() / 1e9))
# MFMA floor of the K-loop: 9 taps x 8 k-steps x 8 units x 32 cycles; the row-tile units skip one unit in 6 of the 9 taps
# (fp8: 2 k-steps of K = 64 per tap, 64 cycles per MFMA; two workgroups per CU share each SIMD)
units = 72 if os.environ.get("NO_ROWT") == "1" else 6 * 7 + 3 * 8
floor = units * 2 * 64 if FP8 else units * 8 * 32
print("per layer: kloop %.0f (MFMA floor %d)  epilogue %.0f  barrier %.0f" % (d[:, 0].mean() / 12, floor, d[:, 1].mean() / 12, d[:, 2].mean() / 12))
t0 = d[:, 7]; print("WG start spread (us): ", np.percentile((t0 - t0.min()) / 100, [0, 25, 50, 75, 100]))
print("WG duration us: mean %.1f" % (d[:, 5].mean() / 100))
if not FP8 and ch[32] > 0:  # bf16 row-tile kernel: cycles per conv tap (workgroup 0, wave 0; each stamp drains the LDS queue)
    per = ch[:9] / ch[32:41]
    print("per tap (dy, dx), cycles incl. the stamp's drain: " + "  ".join("(%+d,%+d) %.0f" % (t // 3 - 1, t % 3 - 1, per[t]) for t in range(9)))
