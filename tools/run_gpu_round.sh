cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/gputest_r03f.txt
tail -8 gpurun_out/gputest_r03f.txt
REPS=9 python tools/kernel_times.py c5 c4on c2 c3
python bench.py --no-cpu-baseline --no-parity-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'])
print(json.dumps(d['plan_latency'])[:900])
print({k:(v['kernel_ms'], v['ms_per_step']) for k,v in d['secondary'].items()})
"
