"""Kernel time of long bands (the band-in-HBM layout; more than 512 poses: four poses per lane, round 5): B bands of n poses, 0.25 m per pose,
a point obstacle every 9 poses and a moving one every 67 (tests/test_gpu_parity.py: _long_scene), TebConfig defaults with teb_autosize off,
4 x 5 iterations.   usage (GPU box): python tools/long_band_times.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_gpu_parity as T
from teb_local_planner_amd import planner
for B in (1, 64, 256):
    row = []
    for n in (300, 400, 500, 640, 768, 944):
        rng = np.random.default_rng(n)
        cfg, obst, via, batch = T._long_scene(n, "points", rng, stride=n, ns=[n - (b % 7) for b in range(B)])
        cfg.trajectory.teb_autosize = False
        cfg.trajectory.max_samples = 1000
        s = planner.make_solver(cfg, obst, via, batch)
        s.snapshot(); ms = []
        for _ in range(5):
            s.restore(); s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize(); ms.append(s.last_kernel_ms())
        res = s.results(); s.close()
        row.append("n=%d %.2f ms (%d trials)" % (n, np.median(ms), int(res.lm_trials.sum())))
    print("B=%-4d %s" % (B, "   ".join(row)), flush=True)
