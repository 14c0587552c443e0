#!/bin/bash
# The copy-out of activations / gradients in k_train_fwd / k_train_bwd: same-box A/Bs of the product library against
# diagnostic variants (build_variant(name, [flag])):
#   copyafter   -DBZ_EXP_COPY_AFTER_BARRIER   row-tile shapes at 64 channels: the copy pass right behind the layer's own barrier
#   nostores    -DBZ_EXP_NO_TRAIN_STORES      timing only: store_tile() stores nothing
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/exp_train_stores.sh'
for rep in 1 2; do
for shape in "128 6 1024" "64 4 2048" "64 4 1024"; do
for so in product copyafter nostores; do
  [ -f build/variants/libbz_hip.$so.so ] || [ $so = product ] || continue
  if [ $so = product ]; then unset BZ_HIP_SO BZ_ALLOW_EXPERIMENT; else export BZ_HIP_SO=$PWD/build/variants/libbz_hip.$so.so BZ_ALLOW_EXPERIMENT=1; fi
  echo "$shape | $so | $(python3 tools/bench_train.py $shape --quick 2>&1 | grep 'k_train_fwd\|k_train_bwd\|incl. Adam' | sed 's/TFLOP.*//; s/bf16 graph  whole step on HIP kernels incl. Adam (10 launches)/step/' | tr -s ' ' | tr '\n' ';')"
done; done; done
