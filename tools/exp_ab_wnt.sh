#!/bin/bash
# Do the tower's weight fragments have to evict the tree from the L2?  A/B of non-temporal weight loads
# (-DBZ_EXP_WEIGHTS_NT, its own build/variants/libbz_hip.wnt.so) against the product library, interleaved on ONE device:
# the tower alone (tools/bench_net.py), then the single-pipeline bench (tree step and tower alternate on one stream:
# `select` = k_tree_step by HIP events) and the default two-pipeline bench.  Run through gpurun.
set -e
cd "$(dirname "$0")/.."
SO=$(python -c "from betazero_amd import build; print(build.build_variant('wnt', ['-DBZ_EXP_WEIGHTS_NT']))")
show='import json,sys; d=json.load(sys.stdin); k=d["kernel_ms_total"]; print("  games/s %.1f  tower frac %.4f  tree step %.2f us  tower %.1f us per launch" % (d["value"], d["roofline"]["frac"], k["select"]/d["roofline_tree"]["launches"]*1e3, k["tower"]/d["roofline"]["launches"]*1e3))'
for i in 1 2; do
  echo "== product"; python tools/bench_net.py 4096 500 | grep -E "tower"
  python bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$show"
  python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$show"
  echo "== non-temporal weight loads"; BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 python tools/bench_net.py 4096 500 | grep -E "tower"
  BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 python bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$show"
  BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$show"
done
