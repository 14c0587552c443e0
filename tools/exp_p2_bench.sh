BZ_EXTRA_HIPCC_FLAGS="-DBZ_TOWER_P=2" python betazero_amd/build.py > /dev/null 2>&1 || echo BUILD FAIL
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('P=2', {k:d[k] for k in ('value','ms_per_step','kernel_ms_total')}); print(d['roofline']['avg_launch_ms'])"
python betazero_amd/build.py > /dev/null 2>&1
