#!/bin/bash
# k_train_wgrad: what bounds it?  Same-box timing of the product against diagnostic builds (timing only, results wrong) that
# leave one ingredient out: "wgnofetch" = no input stream from HBM (every stage recomputes the first), "wgnolds" = no
# activation-fragment transpose reads (10 of the 14 LDS reads per k-step), "wgnomfma" = the reads without the MFMAs.
# Build: build_variant("wgnofetch", ["-DBZ_EXP_WGRAD_NO_FETCH"]), ("wgnolds", ["-DBZ_EXP_WGRAD_NO_LDS_READS"]), ("wgnomfma", ["-DBZ_EXP_WGRAD_NO_MFMA"])
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/exp_wgrad_bounds.sh'
for rep in 1 2; do
for shape in "128 6 1024" "64 4 1024"; do
for so in product wgnofetch wgnolds wgnomfma; do
  [ -f build/variants/libbz_hip.$so.so ] || [ $so = product ] || continue
  if [ $so = product ]; then unset BZ_HIP_SO BZ_ALLOW_EXPERIMENT; else export BZ_HIP_SO=$PWD/build/variants/libbz_hip.$so.so BZ_ALLOW_EXPERIMENT=1; fi
  echo "$shape | $so | $(python3 tools/bench_train.py $shape --kernels-only 2>&1 | grep 'k_train_wgrad' | awk '{printf "%s %s us  ", $1, $2}')"
done; done; done
