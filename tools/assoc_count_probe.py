import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teb_local_planner_amd import planner, scenes
for name, mk in (("c4", lambda: scenes.scene_c4(stride=288)), ("c2", lambda: scenes.scene_c2(stride=208))):
    cfg, obst, via, batch = mk()
    s = planner.make_solver(cfg, obst, via, batch)
    for b in (0, 3):
        if b >= batch.count: continue
        r = s.debug_linearize(b, int(batch.n[b]))
        c = np.bincount(r["assoc_pose"], minlength=int(batch.n[b]))
        print(name, "band", b, "assoc per pose: mean %.1f max %d  p50 %d p90 %d; per 64-pose wave max:" % (c.mean(), c.max(), np.percentile(c, 50), np.percentile(c, 90)), [int(c[k:k+64].max()) for k in range(0, len(c), 64)])
    s.close()
