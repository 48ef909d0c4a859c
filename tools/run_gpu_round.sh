cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PROF_DIR=prof_r03 bash tools/profile.sh > gpurun_out/profile_r03_stdout.txt 2>&1
PROF_DIR=prof_r03 python tools/export_profile.py r03 > gpurun_out/export_r03.txt 2>&1
tail -3 gpurun_out/export_r03.txt
mkdir -p gpurun_out/profiles_r03
cp profiles/rocprof_r03_summary.* profiles/r03_bench_*.json gpurun_out/profiles_r03/ 2>/dev/null
cp gpurun_out/prof_r03/mfma_probe.txt gpurun_out/profiles_r03/ 2>/dev/null
rm -rf gpurun_out/prof_r03
ls gpurun_out/profiles_r03
