"""How much does a band's optimizeTEB slow down when the other CUs are busy with the same work? The batch is B copies of ONE band of
the headline scene (same obstacles, capacity 288), so every workgroup does identical work and the kernel time is the time of one band
under B-fold load: the growth from B = 1 to B = 256 is what the shared levels (L2, Infinity Fabric, HBM: scratch and band-copy traffic)
cost. Run once per library (TEB_AMD_LIB=...)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teb_local_planner_amd import planner, scenes, _abi
cfg, obst, via, full = scenes.scene_c4(stride=288)
for j0 in [int(a) for a in (sys.argv[1:] or ["0"])]:
    x, y, th, dt = full.get_teb(j0)
    for B in (1, 8, 32, 64, 128, 256, 512):
        b = _abi.TebBatchHost(B, 288)
        for k in range(B):
            b.set_teb(k, x, y, th, dt); b.has_vel_goal[k] = 1
        s = planner.make_solver(cfg, obst, via, b); s.snapshot(); ms = []
        for _ in range(5):
            s.restore(); s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize(); ms.append(s.last_kernel_ms())
        r = s.results(); n = s.pose_counts()
        print("band %3d x %3d copies: kernel %.3f ms (min %.3f)  poses %d  trials %d" % (j0, B, np.median(ms), min(ms), int(n[0]), int(r.lm_trials[0])))
        s.close()
