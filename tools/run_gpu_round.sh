cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_multi_cu.py tests/test_gpu_bit_fingerprint.py -q -x 2>&1 | tail -2
REPS=11 python tools/kernel_times.py c5 c2 c3
