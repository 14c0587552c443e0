#!/bin/bash
# Round 5: the bf16 tower is power-bound (profiles/r01_power_clock_rocm_smi.txt: 1320 W at 2.0 GHz on random data against
# 985 W at 2.4 GHz on zero weights), so what can still move it is energy per MFMA.  Does the ISSUE ORDER of the same MFMAs
# matter?  Product: per unit the two channel halves back to back (consecutive MFMAs share the activation operand, the
# weight operand alternates).  -DBZ_EXP_MFMA_AMAJOR: all units with channel half 0, then all with half 1 (consecutive
# MFMAs share the weight operand).  Outputs are bit-identical.  Interleaved on ONE device.  Run through gpurun.
set -e
cd "$(dirname "$0")/.."
SO=$(python -c "from betazero_amd import build; print(build.build_variant('amajor', ['-DBZ_EXP_MFMA_AMAJOR']))")
for i in 1 2 3; do
  echo "== product (unit-major: shared activation operand)"; python tools/bench_net.py 4096 1500 | grep -E "forward|tower"
  echo "== channel-half-major (shared weight operand)"; BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 python tools/bench_net.py 4096 1500 | grep -E "forward|tower"
done
