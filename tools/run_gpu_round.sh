cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/micro/block_factor_bench > gpurun_out/block_factor_bench_r03.txt 2>&1; cat gpurun_out/block_factor_bench_r03.txt
