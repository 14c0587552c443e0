#!/bin/bash
# Round 5: what would hiding the tower's epilogue behind MFMAs gain AT MOST?  The kernel is power-bound, so fewer cycles do not
# translate one to one into less time.  -DBZ_EXP_EPILOGUE_HALF skips half of every layer's epilogue (the b = 0 quarters: results are
# wrong, stale activations stay in LDS) -- the time of that build is a lower bound for any scheme that overlaps that half with the
# K-loop.  Interleaved with the product on ONE device.  Run through gpurun.
set -e
cd "$(dirname "$0")/.."
SO=$(python -c "from betazero_amd import build; print(build.build_variant('ephalf', ['-DBZ_EXP_EPILOGUE_HALF']))")
for i in 1 2 3; do
  echo "== product"; python tools/bench_net.py 4096 1500 | grep -E "forward|tower"
  echo "== half of the epilogue skipped (timing only)"; BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 python tools/bench_net.py 4096 1500 | grep -E "forward|tower"
done
