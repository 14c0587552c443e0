#!/bin/bash
# Evidence collection for one round, run on the GPU box through gpurun:
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_round.sh r03 [sections]'
# Raw rocprofv3 output goes to gpurun_out/<tag>prof/; tools/summarize_prof.py then writes the small files
# that are committed under profiles/.  Counters are collected in their own passes (--pmc with
# --kernel-trace only), the program sits directly after `--` (no env/bash hop under the profiler).
TAG=${1:-r02}
WHAT=${2:-all}
OUT=gpurun_out/${TAG}prof
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { [ "$WHAT" = all ] || echo "$WHAT" | grep -qw "$1"; }
run() { echo "== $*" >&2; timeout -k 10 "$@"; echo "   rc=$?" >&2; }

if has stats; then
  # per-kernel durations of the same command the bench line comes from (agreement check)
  run 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_s2" -o s2 -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/bench_under_rocprof_streams2.json" 2> "$OUT/stats_s2.err"
  run 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_s1" -o s1 -- python3 bench.py --steps 6 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary > "$OUT/bench_under_rocprof_streams1.json" 2> "$OUT/stats_s1.err"
  run 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats_ttt" -o ttt -- python3 bench.py --workload ttt --no-cpu-baseline > "$OUT/bench_ttt_under_rocprof.json" 2> "$OUT/stats_ttt.err"
  run 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats_env" -o env -- python3 tools/bench_env.py > "$OUT/bench_env.txt" 2> "$OUT/stats_env.err"
fi
if has pmc_tower; then
  run 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d "$OUT/pmc_tower_sq" -o t -- python3 tools/bench_net.py 4096 60 > "$OUT/pmc_tower_sq.txt" 2>&1
  run 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES -d "$OUT/pmc_tower_inst" -o t -- python3 tools/bench_net.py 4096 60 > "$OUT/pmc_tower_inst.txt" 2>&1
  run 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_tower_fetch" -o t -- python3 tools/bench_net.py 4096 60 > "$OUT/pmc_tower_fetch.txt" 2>&1
  run 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_tower_write" -o t -- python3 tools/bench_net.py 4096 60 > "$OUT/pmc_tower_write.txt" 2>&1
  run 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_tower_tcc" -o t -- python3 tools/bench_net.py 4096 60 > "$OUT/pmc_tower_tcc.txt" 2>&1
fi
if has pmc_bench; then
  # HBM traffic of the tower AT THE BENCH'S OWN BATCH (two pipelines of 2048 games: ~1850 packed positions per launch):
  # the same command as the bench line, counters in their own passes
  run 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_bench_fetch" -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/pmc_bench_fetch.json" 2> "$OUT/pmc_bench_fetch.err"
  run 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_bench_write" -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/pmc_bench_write.json" 2> "$OUT/pmc_bench_write.err"
fi
if has pmc_cfg5sp; then
  # secondary.cfg5_selfplay's shape: the fp8 tower as the in-loop evaluator of 8192 games (two pipelines of 4096), counters per launch
  run 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_cfg5sp_fetch" -o t -- python3 bench.py --precision fp8 --games 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/pmc_cfg5sp_fetch.json" 2> "$OUT/pmc_cfg5sp_fetch.err"
  run 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_cfg5sp_write" -o t -- python3 bench.py --precision fp8 --games 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/pmc_cfg5sp_write.json" 2> "$OUT/pmc_cfg5sp_write.err"
fi
if has pmc_ttt; then
  for gw in 4 0; do   # 4 = the TTT-specialised fused search at its default lanes (cfg 2 as bench.py runs it), 0 = the generic fused kernel (--ttt-lanes -1)
    if [ $gw = 0 ]; then LANES="--ttt-lanes -1"; else LANES="--ttt-lanes $gw"; fi
    run 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d "$OUT/pmc_ttt_sq_gw$gw" -o t -- python3 bench.py --workload ttt $LANES --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_ttt_sq_gw$gw.json" 2> "$OUT/pmc_ttt_sq_gw$gw.err"
    run 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_ttt_tcc_gw$gw" -o t -- python3 bench.py --workload ttt $LANES --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_ttt_tcc_gw$gw.json" 2> "$OUT/pmc_ttt_tcc_gw$gw.err"
    run 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_ttt_fetch_gw$gw" -o t -- python3 bench.py --workload ttt $LANES --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_ttt_fetch_gw$gw.json" 2> "$OUT/pmc_ttt_fetch_gw$gw.err"
    run 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_ttt_write_gw$gw" -o t -- python3 bench.py --workload ttt $LANES --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_ttt_write_gw$gw.json" 2> "$OUT/pmc_ttt_write_gw$gw.err"
  done
fi
if has pmc_env; then
  run 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT/pmc_env_sq" -o t -- python3 tools/bench_env.py > "$OUT/pmc_env_sq.txt" 2>&1
  run 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_env_fetch" -o t -- python3 tools/bench_env.py > "$OUT/pmc_env_fetch.txt" 2>&1
  run 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_env_write" -o t -- python3 tools/bench_env.py > "$OUT/pmc_env_write.txt" 2>&1
fi
if has tree; then
  # k_tree_step: per-simulation durations from the kernel trace (fixed cost vs cost per level), stamps per phase
  run 600 bash tools/exp_tree.sh "${TAG}" > "$OUT/tree_trace.txt" 2>&1
  SO=$(python3 -c "from betazero_amd import build; print(build.build_variant('treestamps', ['-DBZ_EXP_TREE_STAMPS']))")
  BZ_HIP_SO="$SO" BZ_ALLOW_EXPERIMENT=1 run 300 python3 tools/exp_tree_stamps.py 800 > "$OUT/tree_stamps.txt" 2>&1
fi
if has pmc_tree; then
  # cfg 3's tree step (hidden behind the net in the bench): a short single-pipeline run, counters per launch
  run 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d "$OUT/pmc_tree_sq" -o t -- python3 bench.py --steps 1 --warmup 0 --sims 200 --streams 1 --no-cpu-baseline --no-secondary > "$OUT/pmc_tree_sq.json" 2> "$OUT/pmc_tree_sq.err"
  run 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_tree_tcc" -o t -- python3 bench.py --steps 1 --warmup 0 --sims 200 --streams 1 --no-cpu-baseline --no-secondary > "$OUT/pmc_tree_tcc.json" 2> "$OUT/pmc_tree_tcc.err"
  run 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_tree_fetch" -o t -- python3 bench.py --steps 1 --warmup 0 --sims 200 --streams 1 --no-cpu-baseline --no-secondary > "$OUT/pmc_tree_fetch.json" 2> "$OUT/pmc_tree_fetch.err"
  run 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_tree_write" -o t -- python3 bench.py --steps 1 --warmup 0 --sims 200 --streams 1 --no-cpu-baseline --no-secondary > "$OUT/pmc_tree_write.json" 2> "$OUT/pmc_tree_write.err"
fi
if has pmc_fp8; then
  run 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d "$OUT/pmc_fp8_sq" -o t -- python3 tools/bench_net.py 8192 60 fp8 > "$OUT/pmc_fp8_sq.txt" 2>&1
  run 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fp8_fetch" -o t -- python3 tools/bench_net.py 8192 60 fp8 > "$OUT/pmc_fp8_fetch.txt" 2>&1
  run 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_fp8_write" -o t -- python3 tools/bench_net.py 8192 60 fp8 > "$OUT/pmc_fp8_write.txt" 2>&1
fi
if has clock; then
  # in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz) and per-phase stamps of the fused net kernel
  run 300 bash tools/exp_stamps.sh > "$OUT/tower_clock_bf16.txt" 2>&1
  FP8=1 run 300 bash tools/exp_stamps.sh > "$OUT/tower_clock_fp8.txt" 2>&1
  TAPS=1 run 300 bash tools/exp_stamps.sh > "$OUT/tower_taps_bf16.txt" 2>&1   # cycles per conv tap (stamped variant)
  ZERO=1 run 300 bash tools/exp_stamps.sh > "$OUT/tower_clock_bf16_zero_weights.txt" 2>&1   # same instruction stream, operands that toggle nothing
fi
if has ab; then
  # cost of the in-library kernel timers on `value`, and 1 / 2 / 3 pipelines, interleaved on ONE device
  for i in 1 2; do
    for v in "--streams 2" "--streams 2 --no-kernel-timers" "--streams 3" "--streams 1"; do
      tagv=$(echo "$v" | tr -d ' -')
      run 200 python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary $v > "$OUT/ab_${tagv}_$i.json" 2> /dev/null
    done
  done
fi
if has train; then
  # the training step: per-kernel table of 200 replays of the captured step (tower on the HIP kernels), and the microbench
  run 300 rocprofv3 --kernel-trace --stats -d "$OUT/train_step" -o t -- python3 tools/prof_train_step.py > "$OUT/train_step.txt" 2>&1
  run 300 python3 tools/bench_train.py 64 4 1024 > "$OUT/bench_train.txt" 2>&1
  run 300 python3 tools/bench_train.py 128 6 1024 --quick > "$OUT/bench_train_128ch_6blocks.txt" 2>&1
fi
# the result databases are tens of MB: summarise here, ship only the summaries
python3 tools/summarize_prof.py "$TAG" --out "$OUT/summary" >&2
find "$OUT" -name "*_results.db" -delete
ls -R "$OUT" | head -80 >&2
