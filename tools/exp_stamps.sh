#!/bin/bash
# In-kernel stamps (s_memtime per phase, in-kernel clock = d s_memtime / d s_memrealtime x 100 MHz) of the
# fused net kernel.  Builds the diagnostic variant into its OWN file (build/variants/libbz_hip.stamps.so);
# the product library is not touched.  FP8=1 for the fp8 kernel.  Run on the GPU box (gpurun).
set -e
cd "$(dirname "$0")/.."
# TAPS=1 adds a stamp pair around every conv tap (per-tap cycles; perturbs the layer totals by the stamps' LDS drains)
if [ "$TAPS" = 1 ]; then
  SO=$(python -c "from betazero_amd import build; print(build.build_variant('stampstaps', ['-DBZ_EXP_STAMPS', '-DBZ_EXP_STAMPS_TAPS']))")
else
  SO=$(python -c "from betazero_amd import build; print(build.build_variant('stamps', ['-DBZ_EXP_STAMPS']))")
fi
BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 python tools/exp_stamps.py
