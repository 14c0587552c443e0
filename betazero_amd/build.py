"""Build recipe for libbz_hip.so (gfx950 only; hipcc cross-compiles without a GPU).

The product library is always built with exactly FLAGS: nothing from the environment reaches its
compile line.  Diagnostic variants (in-kernel stamps, ...) are built by build_variant() into a
SEPARATE file, build/variants/libbz_hip.<name>.so (outside the package), and are only ever loaded when BZ_HIP_SO points at them
(betazero_amd/_lib.py); they never overwrite the product library.  The flag set a library was
built with is compiled into it (bz_build_info()) and a hash of it sits next to the .so, so a
library built with other flags is stale no matter what its mtime says."""
import glob
import hashlib
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


def _flag_hash(extra):
    return hashlib.sha256(" ".join(FLAGS + list(extra)).encode()).hexdigest()[:16]


VARIANT_DIR = os.path.join(HERE, "..", "build", "variants")  # git-ignored, travels to the GPU box, never inside the package


def variant_path(name):
    os.makedirs(VARIANT_DIR, exist_ok=True)
    return os.path.abspath(os.path.join(VARIANT_DIR, f"libbz_hip.{name}.so"))


def stale(so=SO, extra=()):
    if not os.path.exists(so):
        return True
    try:
        if open(so + ".flags").read().strip() != _flag_hash(extra):
            return True
    except OSError:
        return True
    t = os.path.getmtime(so)
    return any(os.path.getmtime(p) > t for p in DEPS if os.path.exists(p))


def _compile(so, extra, verbose):
    info = " ".join(f for f in list(extra) if f.startswith("-D")) or "product"
    cmd = [hipcc()] + FLAGS + list(extra) + [f'-DBZ_BUILD_INFO="{info}"', "-o", so + ".tmp"] + SRC
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(so + ".tmp", so)  # an interrupted compile leaves the old library in place
    with open(so + ".flags", "w") as f:
        f.write(_flag_hash(extra) + "\n")
    return so


def build(force=False, verbose=False):
    """the product library: fixed flags, nothing taken from the environment"""
    if not force and not stale():
        return SO
    return _compile(SO, (), verbose)


def build_variant(name, extra_flags, force=False, verbose=False):
    """a diagnostic variant (e.g. name="stamps", extra_flags=["-DBZ_EXP_STAMPS"]) in its own file;
    load it with BZ_HIP_SO=<path> BZ_ALLOW_EXPERIMENT=1"""
    assert name and name != "so"
    extra = ["-DBZ_EXPERIMENT"] + list(extra_flags)
    so = variant_path(name)
    if not force and not stale(so, extra):
        return so
    return _compile(so, extra, verbose)


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":
        print(build_variant(sys.argv[2], sys.argv[3:], force=True, verbose=True))
    else:
        build(force=True, verbose=True)
