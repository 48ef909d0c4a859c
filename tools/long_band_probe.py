"""How much of the headline step is HBM contention? The 35 bands that end above 256 poses alone (35 workgroups) vs the full batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teb_local_planner_amd import planner, scenes, _abi
cfg, obst, via, batch = scenes.scene_c4(B=256, n=200, seed=1004, stride=288)
def run(b, reps=5):
    s = planner.make_solver(cfg, obst, via, b); s.snapshot(); ms = []
    for _ in range(reps):
        s.restore(); s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize(); ms.append(s.last_kernel_ms())
    n = s.pose_counts(); s.close(); return float(np.median(ms)), n
ms, n = run(batch)
print("all 256 bands: %.3f ms; poses after %d..%d" % (ms, n.min(), n.max()))
for name, sel in (("35 longest", np.argsort(-n)[:35]), ("35 shortest", np.argsort(n)[:35]), ("the longest alone", np.argsort(-n)[:1]), ("128 longest", np.argsort(-n)[:128])):
    sub = _abi.TebBatchHost(len(sel), 288)
    for k, b in enumerate(sel):
        sub.set_teb(k, *batch.get_teb(int(b))); sub.has_vel_goal[k] = batch.has_vel_goal[int(b)]
    m2, n2 = run(sub)
    print("%-18s: %.3f ms; poses after %d..%d" % (name, m2, n2.min(), n2.max()))
