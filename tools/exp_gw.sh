for gw in 2 1; do
  sed -i "s/    static constexpr int GW = [0-9]*;   \/\/ .* games per wave/    static constexpr int GW = $gw;   \/\/ $((64/gw)) games per wave/" betazero_amd/csrc/bz_rules.h
  python betazero_amd/build.py > /dev/null 2>&1 || echo BUILD FAIL
  echo "== TTT GW=$gw"; python -m pytest tests -m gpu -x -q -k "cfg2 or golden" 2>&1 | tail -1
  python bench.py --workload ttt --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])"
done
