cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
PROF_DIR=prof_r03 bash tools/profile.sh > gpurun_out/profile_r03_stdout.txt 2>&1
PROF_DIR=prof_r03 python tools/export_profile.py r03 > gpurun_out/export_r03.txt 2>&1
mkdir -p gpurun_out/profiles_r03
cp profiles/rocprof_r03_summary.* profiles/r03_bench_*.json gpurun_out/profiles_r03/ 2>/dev/null
cp gpurun_out/prof_r03/mfma_probe.txt gpurun_out/profiles_r03/ 2>/dev/null
rm -rf gpurun_out/prof_r03
python bench.py > gpurun_out/bench_r03c.json 2> gpurun_out/bench_r03c.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_r03c.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'traffic', d['roofline']['traffic'], d['roofline'].get('traffic_note'))
print(d['plan_latency'])
print({k:(v['kernel_ms']) for k,v in d['secondary'].items()})
"
