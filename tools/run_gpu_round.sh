#!/bin/bash
# One measured pass of a round on the GPU box: GPU tests, the profile of the bench command (rocprofv3 passes, PMC, phase split, band times),
# the product's own phase split, and the bench line.   usage (via gpurun): bash tools/run_gpu_round.sh r05
# Needs tools/libteb_amd_prof.so (bash tools/build_prof.sh, before the call: no compiler run belongs on the GPU budget). Everything lands in
# gpurun_out/ (merged back); copy the files named *_<tag>* into profiles/ and commit them ONCE per tree that is meant to be judged.
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
bash tools/profile.sh > gpurun_out/profile_sh.log 2>&1
python tools/export_profile.py $TAG >> gpurun_out/profile_sh.log 2>&1
python tools/phase_split.py c4on c4 c3 c2 > profiles/phase_split_$TAG.txt 2>&1
cp profiles/*_$TAG* gpurun_out/ 2>/dev/null
for k in trace fetch write; do cp gpurun_out/prof/bench_$k.json gpurun_out/${TAG}_bench_$k.json; done
rm -rf gpurun_out/prof
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python - <<PY
import json
d = json.loads(open('gpurun_out/bench_$TAG.json').read().strip().splitlines()[-1])
r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'kernel ms', r['kernel_ms'], 'frac', r['frac'], 'traffic', r['traffic'], r.get('traffic_note'))
print('phases', r.get('phases', {}).get('share_mean_of_bands'))
print({k: round(v['kernel_ms'], 3) for k, v in d['secondary'].items()})
PY
