import os, sys, subprocess, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betazero_amd import _lib
from betazero_amd.net import DeviceNet, PolicyValueNet
B = 4096
rng = np.random.default_rng(0)
x = rng.integers(0, 2**63, size=B, dtype=np.int64); y = rng.integers(0, 2**63, size=B, dtype=np.int64)
own = torch.as_tensor(x & ~y).cuda(); opp = torch.as_tensor(y & ~x).cuda()
L = _lib.lib()
def run(tag, mod, iters=300, smi=False):
    net = DeviceNet.from_module(mod, B)
    for _ in range(20): net.forward(own, opp)
    torch.cuda.synchronize()
    pw = []
    stop = False
    def poll():
        while not stop:
            try:
                o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                pw.append(" | ".join(l.strip() for l in o.splitlines() if "Power" in l or "sclk" in l))
            except Exception as e:
                pw.append(str(e))
            time.sleep(0.2)
    if smi:
        th = threading.Thread(target=poll); th.start()
    L.bz_profile_reset(); L.bz_profile_enable(1)
    t0 = time.time()
    while time.time() - t0 < (3.0 if smi else 0.0) or iters > 0:
        net.forward(own, opp); iters -= 1
        if iters % 50 == 0: torch.cuda.synchronize()
    L.bz_profile_enable(0)
    n, t, ms = _lib.profile_read()["tower"]
    stop = True
    if smi: th.join()
    print(f"{tag}: tower {ms/t*1e3:.1f} us"); 
    for p in pw[-3:]: print("   ", p)
torch.manual_seed(0)
m = PolicyValueNet(128, 6, 64).round_to_bf16_()
run("random-init net", m, smi=True)
z = PolicyValueNet(128, 6, 64)
with torch.no_grad():
    for p in z.parameters(): p.zero_()
run("all-zero net   ", z, smi=True)
run("random-init net", m)
