"""median kernel ms of the measured configurations for the build given by TEB_AMD_LIB (A/B of kernel variants: run the libraries
alternately in one gpurun call, e.g. tools/ab.sh)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teb_local_planner_amd import planner, scenes, _abi
def _c4fix():
    a = scenes.scene_c4(stride=208); a[0].trajectory.teb_autosize = False; return a
CASES = {"c4on": lambda: scenes.scene_c4(stride=288), "c4fix": _c4fix, "c2": lambda: scenes.scene_c2(stride=208), "c3": lambda: scenes.scene_c3(stride=208),
         "c5": lambda: scenes.scene_c5(stride=320)}
reps = int(os.environ.get("REPS", "15"))
out = []
for name in (sys.argv[1:] or list(CASES)):
    cfg, obst, via, batch = CASES[name]()
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(generic_config_path=True) if os.environ.get("GENERIC") else None)   # GENERIC=1: never the kernels specialised on the TebConfig defaults
    s.snapshot()
    ms = []
    for r in range(reps + 2):
        s.restore()
        s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize()
        ms.append(s.last_kernel_ms())
    out.append("%s %.3f" % (name, np.median(ms[2:])))
    s.close()
print(os.path.basename(os.environ.get("TEB_AMD_LIB", "libteb_amd.so")), " ".join(out), flush=True)
