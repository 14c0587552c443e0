#!/bin/bash
# Ceiling of relaxing the per-layer barrier of the bf16 tower (per-row ready counters instead of
# __syncthreads()): the kernel WITHOUT the barrier (timing only -- its results are wrong) against the product kernel,
# interleaved on ONE device.  Whatever a row-granular hand-over could gain is bounded by this difference.
# The variant goes to its own file (build/variants/libbz_hip.nobarrier.so).  Run through gpurun.
set -e
cd "$(dirname "$0")/.."
SO=$(python -c "from betazero_amd import build; print(build.build_variant('nobarrier', ['-DBZ_EXP_NO_LAYER_BARRIER']))")
for i in 1 2 3; do
  echo "== product (one barrier per layer)"; python tools/bench_net.py 4096 1000 | grep -E "forward|tower"
  echo "== no per-layer barrier (timing only)"; BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 python tools/bench_net.py 4096 1000 | grep -E "forward|tower"
done
