import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from teb_local_planner_amd import scenes, planner, _abi
cfg, obst, via, _ = scenes.scene_c4(B=1, n=200)
cfg.trajectory.max_samples = 500
cfg.trajectory.teb_autosize = False
for n in (343, 344, 500):
    rng = np.random.default_rng(1)
    B = 64
    batch = _abi.TebBatchHost(B, n)
    for b in range(B):
        px, py, th, dt = scenes.sine_band(n, 0.1 * n, rng.uniform(-0.3, 0.3), 1.0, cfg.robot.max_vel_x)
        batch.set_teb(b, px, py, th, dt)
    s = planner.make_solver(cfg, obst, via, batch)
    s.snapshot()
    ms = []
    for _ in range(4):
        s.restore()
        s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize()
        ms.append(s.last_kernel_ms())
    r = s.results()
    print(n, "poses x", B, "bands: kernel ms", np.median(ms), "ok", int((r.status == 0).sum()), "iters", int(r.lm_iterations.sum()))
    s.close()
