"""Kernel time of the BASELINE configurations per layout of the normal matrix (auto / blocks in LDS / band + hybrid solve / band in HBM)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teb_local_planner_amd import planner, scenes, _abi
def c4f():
    a = scenes.scene_c4(stride=208); a[0].trajectory.teb_autosize = False; return a
CASES = [("C1 1x50x3", lambda: scenes.scene_c1()), ("C2 1x200x100", lambda: scenes.scene_c2(stride=208)), ("C3 64x150x200", lambda: scenes.scene_c3(stride=208)),
         ("C4 fixed 200", c4f), ("C4 headline (288)", lambda: scenes.scene_c4(stride=288)), ("HCP-like 5x130x12", lambda: scenes.scene_c3(B=5, n=130, M=12, stride=224))]
for name, mk in CASES:
    row = []
    for lay in ("auto", "cr", "band", "bandg"):
        cfg, obst, via, batch = mk()
        try:
            s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(layout=lay))
        except planner.TebAmdError as e:
            row.append("%s n/a" % lay); continue
        s.snapshot(); ms = []
        for _ in range(5):
            s.restore(); s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize(); ms.append(s.last_kernel_ms())
        s.close()
        row.append("%s %.3f" % (lay, np.median(ms)))
    print("%-20s %s" % (name, "   ".join(row)), flush=True)
