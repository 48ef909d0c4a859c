cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/gputest_r03d.txt
tail -12 gpurun_out/gputest_r03d.txt
