"""Build recipe for libbz_hip.so (gfx950 only; hipcc cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libbz_hip.so")
SRC = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
DEPS = SRC + sorted(glob.glob(os.path.join(HERE, "csrc", "*.h"))) + [os.path.join(HERE, "..", "include", "bz_abi.h")]
# -ffp-contract=off: the tree kernels and the f32 parity net must not fuse a*b+c
# (bit-exact parity with the oracle); fmaf is written explicitly where wanted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(p) > t for p in DEPS if os.path.exists(p))


def build(force=False, verbose=False):
    if not force and not stale():
        return SO
    cmd = [hipcc()] + FLAGS + os.environ.get("BZ_EXTRA_HIPCC_FLAGS", "").split() + ["-o", SO] + SRC
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force=True, verbose=True)
