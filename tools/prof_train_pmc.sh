#!/bin/bash
# counter passes over the training kernels alone (tools/prof_train_kernels.py), each counter set in its own run:
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/prof_train_pmc.sh 128 6 1024'
# raw databases under gpurun_out/r04prof_train/<pass>/, summarised by tools/summarize_prof.py r04 --src gpurun_out/r04prof_train
C=${1:-128}; NB=${2:-6}; B=${3:-1024}
OUT=$PWD/gpurun_out/r04prof_train
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() { name=$1; shift; echo "== $name" >&2; timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/train${C}_$name" -o t -- python3 "$GRAFT_REPO_ROOT/tools/prof_train_kernels.py" $C $NB $B > "$OUT/train${C}_$name.txt" 2>&1; echo "   rc=$?" >&2; }
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE &&
run inst SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
# the databases are too big to travel back (64 MiB limit on gpurun_out/): summarise here, keep only the CSVs
cd "$GRAFT_REPO_ROOT" && python3 tools/summarize_prof.py r04 --src "$OUT" --out "$OUT/summary" && rm -rf "$OUT"/train${C}_*/
