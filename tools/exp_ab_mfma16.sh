#!/bin/bash
# does the bf16 tower gain from the 16x16x32 MFMA shape at the power wall?  TIMING-ONLY variant
# (-DBZ_EXP_MFMA16: every 32x32x16 MFMA issued as two 16x16x32 on the same operand registers -- same MACs, same LDS and
# weight traffic, same 8 MFMAs' worth of flops per 1-KB weight fragment; the results are wrong) against the product
# kernel, interleaved on ONE device, plus the in-kernel clock of both from the stamped builds.  Run through gpurun.
# (Record of the round-4 experiment, profiles/r04_ab_tower_16x16.txt: it was run while the 128-channel tower was still on
# 32x32x16.  That tower has since MOVED to the 16x16x32 path for real -- Tw<128, 4, true> -- which -DBZ_EXP_MFMA16 does not
# touch; the flag still applies to the geometries that kept 32x32x16, e.g. `python tools/bench_net.py` on a 64-channel net.)
set -e
cd "$(dirname "$0")/.."
if [ "$FP8" = 1 ]; then  # the same question for the fp8 tower: v_mfma_scale_f32_16x16x128_f8f6f4 against 32x32x64
  SO=$(python -c "from betazero_amd import build; print(build.build_variant('mfma16fp8', ['-DBZ_EXP_MFMA16_FP8']))")
  ST=$(python -c "from betazero_amd import build; print(build.build_variant('stamps', ['-DBZ_EXP_STAMPS']))")
  ST16=$(python -c "from betazero_amd import build; print(build.build_variant('stampsmfma16fp8', ['-DBZ_EXP_STAMPS', '-DBZ_EXP_MFMA16_FP8']))")
  for i in 1 2 3; do
    echo "== product (v_mfma_scale_f32_32x32x64_f8f6f4)"; python tools/bench_net.py 8192 1000 fp8 | grep -E "forward|tower"
    echo "== timing-only 2 x v_mfma_scale_f32_16x16x128_f8f6f4 per unit"; BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 python tools/bench_net.py 8192 1000 fp8 | grep -E "forward|tower"
  done
  echo "== stamps, product shape"; FP8=1 BZ_HIP_SO="$ST" BZ_ALLOW_EXPERIMENT=1 python tools/exp_stamps.py
  echo "== stamps, 16x16x128 timing-only"; FP8=1 BZ_HIP_SO="$ST16" BZ_ALLOW_EXPERIMENT=1 python tools/exp_stamps.py
  exit 0
fi
SO=$(python -c "from betazero_amd import build; print(build.build_variant('mfma16', ['-DBZ_EXP_MFMA16']))")
ST=$(python -c "from betazero_amd import build; print(build.build_variant('stamps', ['-DBZ_EXP_STAMPS']))")
ST16=$(python -c "from betazero_amd import build; print(build.build_variant('stampsmfma16', ['-DBZ_EXP_STAMPS', '-DBZ_EXP_MFMA16']))")
for i in 1 2 3; do
  echo "== product (v_mfma_f32_32x32x16_bf16)"; python tools/bench_net.py 4096 1000 | grep -E "forward|tower"
  echo "== timing-only 2 x v_mfma_f32_16x16x32_bf16 per unit"; BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 python tools/bench_net.py 4096 1000 | grep -E "forward|tower"
done
echo "== stamps, product shape"; BZ_HIP_SO="$ST" BZ_ALLOW_EXPERIMENT=1 python tools/exp_stamps.py
echo "== stamps, 16x16x32 timing-only"; BZ_HIP_SO="$ST16" BZ_ALLOW_EXPERIMENT=1 python tools/exp_stamps.py
