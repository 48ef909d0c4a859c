import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import scenes, planner, _abi
N = int(sys.argv[1])
def run(**opt):
    cfg, obst, via, batch = scenes.scene_c5(stride=320)
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(**opt))
    s.optimize(5, 4, True, 100.0, 1.0, False)
    out = s.download(batch.copy()); r = s.results(); info = s.last_launch_info(); s.close()
    return out, r, info
o0, r0, i0 = run(multi_cu=-1, speculative_trials=-1)
bad = 0
for k in range(N):
    o, r, info = run(speculative_trials=-1)
    same = all(np.array_equal(getattr(o, a), getattr(o0, a)) for a in ("x", "y", "theta", "dt")) and np.array_equal(r.chi2, r0.chi2)
    if not same:
        bad += 1
        if bad <= 4: print("run %d: info %s chi2 %s vs %s" % (k, info, r.chi2, r0.chi2), flush=True)
print("C5 automatic helpers %s: %d of %d differ from the one-CU result" % (info, bad, N))
