"""bitwise fingerprint of the optimised headline bands of a build (TEB_AMD_LIB=...)"""
import os, sys, hashlib, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teb_local_planner_amd import planner, scenes
for name, mk in (("c4on", lambda: scenes.scene_c4(stride=288)), ("c2", lambda: scenes.scene_c2(stride=208)), ("c5", lambda: scenes.scene_c5(stride=320)),
                 ("mixed polygon", lambda: scenes.scene_small_mixed(footprint="polygon")), ("mixed two circles", lambda: scenes.scene_small_mixed(footprint="two_circles"))):
    cfg, obst, via, batch = mk()
    s = planner.make_solver(cfg, obst, via, batch)
    s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize()
    out = s.download(batch.copy()); r = s.results()
    h = hashlib.sha256()
    for a in (out.n, out.x, out.y, out.theta, out.dt, r.cost, r.chi2): h.update(np.ascontiguousarray(a).tobytes())
    print(name, h.hexdigest()[:16], "kernel %.3f ms" % s.last_kernel_ms())
    s.close()
