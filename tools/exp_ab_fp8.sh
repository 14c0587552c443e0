for i in 1 2; do
for f in "$1" "$2"; do
  BZ_EXTRA_HIPCC_FLAGS="$f" python betazero_amd/build.py > /dev/null 2>&1 || echo BUILD FAIL
  echo "== [$f]"; python tools/bench_net.py 8192 150 fp8 | grep tower
  if [ "$i" = "1" ]; then python -m pytest tests -m gpu -x -q -k "fp8" 2>&1 | tail -1; fi
done; done
python betazero_amd/build.py > /dev/null 2>&1
