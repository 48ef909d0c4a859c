cd $GRAFT_REPO_ROOT
TEB_AMD_LIB=$PWD/tools/libteb_amd_x_scalar.so timeout 300 python -m pytest tests/test_gpu_bit_fingerprint.py -q 2>&1 | grep -E "passed|failed"
CASES="c4on c4fix c3" REPS=11 tools/ab.sh -r 2 tools/libteb_amd_r02.so tools/libteb_amd_x_lds.so tools/libteb_amd_x_scalar.so
