"""where the PCIe-inclusive plan() latency of one band goes: host wall clock per step of upload -> optimise -> select -> download (p50)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import scenes, planner
for name, mk in (("c2", lambda: scenes.scene_c2(stride=208)), ("tick-like 5 x 130", lambda: scenes.scene_c4(B=5, n=130, stride=224))):
    cfg, obst, via, batch = mk()
    s = planner.make_solver(cfg, obst, via, batch)
    T = {k: [] for k in ("upload", "optimize", "sync", "select", "results", "download", "total")}
    for rep in range(30):
        hb = batch.copy()
        t0 = time.perf_counter(); s.upload(hb)
        t1 = time.perf_counter(); s.optimize(5, 4, True, 100.0, 1.0, False)
        t2 = time.perf_counter(); s.synchronize()
        t3 = time.perf_counter(); s.select_best(-1, -1)
        t4 = time.perf_counter(); s.results()
        t5 = time.perf_counter(); s.download(hb)
        t6 = time.perf_counter()
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0)): T[k].append(v)
    print(name, "kernel %.3f ms |" % s.last_kernel_ms(), " ".join("%s %.0f us" % (k, 1e6 * np.median(v[5:])) for k, v in T.items()), flush=True)
    s.close()
