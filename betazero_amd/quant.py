"""fp8 (OCP e4m3fn) fake-quantisation of the policy/value net for the fp8 MFMA path
(BASELINE config 5).  Specification ("fp8 net"):
  * tower conv weights and the two head conv1x1 weights: per OUTPUT CHANNEL power-of-two scale
    s = 2^floor(log2(448 / max|w|)), w_q = e4m3(w * s) (RNE, saturating), effective weight w_q / s;
  * tower activations are stored as e4m3(x * 16) (fixed activation scale 2^4, saturating at 448);
  * stem weights, biases and the FC layers stay as in the bf16 net.
The engine derives the same scales from the effective weights, so passing the fake-quantised
parameter vector to bz_net_create is all that is needed."""
import numpy as np
import torch


def e4m3_round(x):
    """round float array to the nearest e4m3fn value (RNE, saturate at +-448, subnormal step 2^-9)"""
    x = np.asarray(x, dtype=np.float32)
    a = np.minimum(np.abs(x), np.float32(448.0))
    _, ex = np.frexp(np.maximum(a, np.float32(2.0 ** -20)))
    e = np.maximum(ex - 1, -6)
    step = np.ldexp(np.float32(1.0), e - 3).astype(np.float32)
    q = np.minimum(np.round(a / step) * step, np.float32(448.0)).astype(np.float32)
    return np.where(a == 0, np.float32(0.0), np.copysign(q, x)).astype(np.float32)


def channel_scale(w_rows):
    """power-of-two scale per row so that max|row| * s lands in (224, 448]"""
    m = np.abs(w_rows).reshape(w_rows.shape[0], -1).max(1)
    mant, ex = np.frexp(np.where(m > 0, np.float32(448.0) / np.maximum(m, 1e-30), 1.0).astype(np.float32))
    return np.ldexp(np.float32(1.0), ex - 1).astype(np.float32)


@torch.no_grad()
def fake_quantize_fp8_(module):
    """in place: bf16-round everything, then e4m3 fake-quantise the tower / head conv weights"""
    module.round_to_bf16_()
    convs = [c for pair in zip(module.c1, module.c2) for c in pair] + [module.pol, module.val]
    for c in convs:
        w = c.weight.detach().cpu().numpy()
        s = channel_scale(w).reshape(-1, 1, 1, 1)
        c.weight.copy_(torch.from_numpy(e4m3_round(w * s) / s))
    return module
