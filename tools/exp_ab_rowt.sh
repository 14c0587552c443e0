#!/bin/bash
# A/B of the row-tile units of the bf16 tower (8.3 % of the MFMAs skipped) against position-major units on the same
# LDS layout, interleaved on ONE device.  The variant goes to its own file (libbz_hip.norowt.so).  Run through gpurun.
set -e
cd "$(dirname "$0")/.."
SO=$(python -c "from betazero_amd import build; print(build.build_variant('norowt', ['-DBZ_EXP_NO_ROWT']))")
for i in 1 2 3; do
  echo "== product (row-tile units)"; python tools/bench_net.py 4096 1000 | grep -E "forward|tower"
  echo "== position-major units";    BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 python tools/bench_net.py 4096 1000 | grep -E "forward|tower"
done
