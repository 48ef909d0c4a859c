"""Edge construction of the exploration graph at scale: key-point graph over the C4 obstacle table (500 obstacles -> up to 1002 vertices,
10^6 ordered pairs x 500 obstacles). Prints the time of teb_amd_explore_candidates with the path enumeration bounded to one chunk."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import scenes, planner, _abi

cfg, obst, via, batch = scenes.scene_c4(B=1, n=200)
cfg.hcp.simple_exploration = True
cfg.hcp.max_number_classes = 4
s = planner.TebBatchSolver(cfg, 8, 256, len(obst), 1, 1)
s.set_obstacles(obst)
s.set_via_points([])
x, y, th, _ = batch.get_teb(0)
start = [float(x[0]), float(y[0]), float(th[0])]; goal = [float(x[-1]), float(y[-1]), float(th[-1])]
ts = []
for k in range(6):
    s.compact_bands(np.zeros(s.count, np.int32)) if s.count else None
    t0 = time.perf_counter()
    r = s.explore_candidates(start, goal, max_paths=64)
    ts.append(time.perf_counter() - t0)
V, A = s.exploration_graph()
print("obstacles", len(obst), "vertices", r["n_vertices"], "edges", int(A.sum()), "paths examined", r["n_paths"], "bands", r["n_total"])
print("explore_candidates p50 %.2f ms (first call %.2f ms): vertices on the host, %d x %d pair-obstacle tests on the device, adjacency download,"
      " one chunk of paths" % (1e3 * float(np.median(ts[1:])), 1e3 * ts[0], r["n_vertices"] ** 2, len(obst)))
if os.environ.get("WITH_ORACLE"):
    from oracle import oracle_py
    b = _abi.TebBatchHost(8, 256)
    t0 = time.perf_counter()
    o = oracle_py.explore_candidates(cfg, obst, b, 0, -1, start, goal, max_paths=64, vcap=2048, acap=1 << 22)
    print("oracle (one CPU thread): %.2f s; same adjacency: %s" % (time.perf_counter() - t0, bool((np.array([[1 if j in set(row) else 0 for j in range(len(A))] for row in o["adjacency"]], np.uint8) == A).all())))
