for f in "" "-DBZ_EXP_NO_EPI" "-DBZ_EXP_NO_BLOAD" "-DBZ_EXP_NO_MFMA" "-DBZ_EXP_NO_BLOAD -DBZ_EXP_NO_EPI"; do
  BZ_EXTRA_HIPCC_FLAGS="$f" python betazero_amd/build.py > /dev/null 2>&1 || echo BUILD FAIL
  echo "== flags: $f"; python tools/bench_net.py 4096 100 | grep tower
done
