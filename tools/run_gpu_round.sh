cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/gputest_r03g.txt
tail -6 gpurun_out/gputest_r03g.txt
REPS=11 python tools/kernel_times.py c4on c4fix c3 c2 c5
