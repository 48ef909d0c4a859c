import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_gpu_parity as T
from teb_local_planner_amd import planner, _abi
from oracle import oracle_py as orc
seed, b = int(sys.argv[1]), int(sys.argv[2])
cfg, obst, via, batch = T._random_case(seed)
t = cfg.trajectory
fast = not cfg.obstacles.include_dynamic_obstacles
# oracle-resized and GPU-resized initial bands
ores = batch.copy(); 
for k in range(batch.count):
    ores.set_teb(k, *orc.autoresize(*batch.get_teb(k), t.dt_ref, t.dt_hysteresis, t.min_samples, t.max_samples, fast))
s = planner.make_solver(cfg, obst, via, batch); s.optimize(0, 1, False); gres = s.download(batch.copy()); s.close()
cfg.trajectory.teb_autosize = False
def run(state, who):
    if who == "gpu":
        s = planner.make_solver(cfg, obst, via, state); s.optimize(1, 1, True); out = s.download(state.copy()); r = s.results(); s.close(); return out, r
    return orc.optimize_batch(cfg, obst, via, state, inner=1, outer=1)
res = {}
for sname, st in (("oracle-resized", ores), ("gpu-resized", gres)):
    for who in ("gpu", "oracle"):
        res[(sname, who)] = run(st, who)
def diff(a, bb): return max(np.abs(u - v).max() for u, v in zip(a[0].get_teb(b), bb[0].get_teb(b)))
print("same state, gpu vs oracle LM:   oracle-resized %.3g   gpu-resized %.3g" % (diff(res[("oracle-resized", "gpu")], res[("oracle-resized", "oracle")]), diff(res[("gpu-resized", "gpu")], res[("gpu-resized", "oracle")])))
print("same code, oracle- vs gpu-resized state:   oracle LM %.3g   gpu LM %.3g" % (diff(res[("oracle-resized", "oracle")], res[("gpu-resized", "oracle")]), diff(res[("oracle-resized", "gpu")], res[("gpu-resized", "gpu")])))
print("input difference between the two resized states:", max(np.abs(u - v).max() for u, v in zip(ores.get_teb(b), gres.get_teb(b))))
