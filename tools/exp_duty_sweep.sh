#!/bin/bash
# Duty sweep of the fused bf16 net kernel: the same kernel with 0 / 1 / 2 / 4 x 8 idle issue cycles behind every MFMA
# (diagnostic variants libbz_hip.duty<N>.so with in-kernel stamps; the product library is not touched).  If the chip is
# holding its clock down under the MFMA load, the in-kernel clock must RISE as the duty falls and the time per forward
# must grow by less than the added cycles.  Prints, per variant: cycles per layer, in-kernel clock, WG duration.
set -e
cd "$(dirname "$0")/.."
for n in ${SWEEP:-0 f1 f2 f4 1 2 4}; do
  case "$n" in
    0) FL="['-DBZ_EXP_STAMPS']";;
    f*) FL="['-DBZ_EXP_STAMPS', '-DBZ_EXP_NOP1=${n#f}']";;      # fine steps: s_nop 0 x k
    *) FL="['-DBZ_EXP_STAMPS', '-DBZ_EXP_NOPS=$n']";;           # coarse steps: s_nop 7 x k
  esac
  SO=$(python -c "from betazero_amd import build; print(build.build_variant('duty$n', $FL))")
  echo "== variant $n"
  BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 python tools/exp_stamps.py | grep -v amdgpu.ids
  BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 python tools/bench_net.py 4096 200 | grep tower
done
