cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/micro/latency_probe > gpurun_out/latency_probe_r03.json; cat gpurun_out/latency_probe_r03.json
python bench.py > gpurun_out/bench_r03b.json 2> gpurun_out/bench_r03b.err; tail -3 gpurun_out/bench_r03b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r03b.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'sustained', d.get('sustained'))
print('latency_model', {k: v for k, v in d['roofline'].get('latency_model', {}).items() if k not in ('primitives', 'note')})
print('vs_ref', {k: v for k, v in d['parity_check'].get('vs_reference_code', {}).items() if k not in ('outside','against','tests')})
print('num vs_ref', {k: v for k, v in (d['secondary']['c4_g2o_numeric_jacobians'].get('vs_reference_code') or {}).items() if k not in ('outside',)})
print({k:(v['kernel_ms'], v.get('helpers'), v.get('one_cu_per_band')) for k,v in d['secondary'].items()})
print(d['plan_latency'])
PY
