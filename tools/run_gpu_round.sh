set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/gputest_r03b.txt
CASES="c4on c2" tools/ab.sh -r 3 tools/libteb_amd_r02.so tools/libteb_amd_ar.so tools/libteb_amd_nolog.so > gpurun_out/ab_r03b.txt 2>&1
tail -5 gpurun_out/gputest_r03b.txt; cat gpurun_out/ab_r03b.txt
