"""How much of the headline step is the tail of the pose-count distribution: one workgroup per band, the launch ends with the slowest
band; bands above 256 poses need two passes of every per-pose loop (256 lanes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import scenes, planner
for length in (20.0, 18.5, 17.5, 16.5):
    cfg, obst, via, batch = scenes.scene_c4(B=256, n=200, stride=288, length=length)
    s = planner.make_solver(cfg, obst, via, batch)
    s.snapshot()
    ms = []
    for _ in range(5):
        s.restore()
        s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize()
        ms.append(s.last_kernel_ms())
    n = s.pose_counts(); r = s.results()
    print("length %.1f m: poses after %d..%d (mean %.0f, %d bands > 256)  kernel %.2f ms  units %d  -> %.0f k units/s" %
          (length, n.min(), n.max(), n.mean(), int((n > 256).sum()), np.median(ms), int(r.lm_iterations.sum()), r.lm_iterations.sum() / np.median(ms)))
    s.close()
