import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from teb_local_planner_amd import scenes, planner, _abi
cfg, obst, via, batch = scenes.scene_c2(stride=208)
for cap in (208, 256, 343, 501):
    b = _abi.TebBatchHost(1, cap)
    b.set_teb(0, *batch.get_teb(0))
    s = planner.make_solver(cfg, obst, via, b)
    s.snapshot(); ms = []
    for _ in range(6):
        s.restore(); s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize(); ms.append(s.last_kernel_ms())
    print("capacity", cap, "kernel ms", round(float(np.median(ms)), 3), "poses", s.pose_counts())
    s.close()
