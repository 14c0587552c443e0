BZ_EXTRA_HIPCC_FLAGS="-DBZ_EXP_STAMPS $1" python betazero_amd/build.py > /dev/null 2>&1 || echo BUILD FAIL
python tools/exp_stamps.py
python betazero_amd/build.py > /dev/null 2>&1
