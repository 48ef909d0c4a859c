"""Kernel time and solve cycles per LM trial of the C4 batch with EVERY band at a fixed pose count (teb_autosize off, capacity 288: hybrid
solve): how the damped solve scales across the 256-pose boundary.   usage: TEB_AMD_LIB=.. python tools/fixed_n_times.py [n ..]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import planner, scenes, _abi
out = []
for n in [int(a) for a in sys.argv[1:]] or [240, 254, 258, 262, 270, 286]:
    cfg, obst, via, batch = scenes.scene_c4(B=256, n=n, stride=288)
    cfg.trajectory.teb_autosize = False
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(layout="band", fixed_layout=True))
    s.snapshot()
    s.set_phase_log(True)
    ms = []
    for r in range(8):
        s.restore(); s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize()
        ms.append(s.last_kernel_ms())
    log = s.phase_log(); tr = s.results().lm_trials.astype(float)
    out.append("n=%d: %.3f ms, solve %.0f cycles/trial, whole %.0f" % (n, np.median(ms[2:]), (log[:, 4] / tr).mean(), (log[:, 8] / tr).mean()))
    s.close()
print(os.path.basename(os.environ.get("TEB_AMD_LIB", "libteb_amd.so")), " | ".join(out), flush=True)
