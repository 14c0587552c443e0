# A/B of compile-time variants on ONE device: usage: exp_ab.sh "<flagsA>" "<flagsB>" [rounds]
for i in 1 2; do
for f in "$1" "$2"; do
  BZ_EXTRA_HIPCC_FLAGS="$f" python betazero_amd/build.py > /dev/null 2>&1 || echo BUILD FAIL
  echo "== [$f]"; python tools/bench_net.py 4096 150 | grep tower
done; done
python betazero_amd/build.py > /dev/null 2>&1
