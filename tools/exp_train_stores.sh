#!/bin/bash
# What does the per-layer copy-out of activations / gradients cost k_train_fwd / k_train_bwd, and does it matter where it is
# issued?  Same-box A/B of the product library (copy of layer l's result inside layer l + 1's epilogue) against two diagnostic
# variants: "copyafter" (right behind layer l's barrier, the first placement) and "nostores" (store_tile() stores nothing:
# timing only).  Build:  build_variant("copyafter", ["-DBZ_EXP_COPY_AFTER_BARRIER"]), build_variant("nostores", ["-DBZ_EXP_NO_TRAIN_STORES"])
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/exp_train_stores.sh'
for rep in 1 2; do
for shape in "128 6 1024" "64 4 2048" "64 4 1024"; do
for so in product copyafter nostores; do
  if [ $so = product ]; then unset BZ_HIP_SO BZ_ALLOW_EXPERIMENT; else export BZ_HIP_SO=$PWD/build/variants/libbz_hip.$so.so BZ_ALLOW_EXPERIMENT=1; fi
  echo "$shape | $so | $(python3 tools/bench_train.py $shape --kernels-only 2>&1 | grep 'k_train_fwd\|k_train_bwd' | awk '{printf "%s %s us  ", $1, $2}')"
done; done; done
