import os, sys, faulthandler
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np
import test_reference_backend as T
from random_explore_cases import random_explore_case
from teb_local_planner_amd import _abi
seed = int(sys.argv[1]); which = int(sys.argv[2]); nt = int(sys.argv[3]) if len(sys.argv) > 3 else 3
base = random_explore_case(seed)
cfg = base["cfg"]
cfg.optim.no_inner_iterations = 3; cfg.optim.no_outer_iterations = 2
st, gl = np.array(base["start"]), np.array(base["goal"])
d = (gl[:2] - st[:2]) / np.linalg.norm(gl[:2] - st[:2])
starts = [[st[0] + 0.1 * k * d[0], st[1] + 0.1 * k * d[1], st[2]] for k in range(nt)]
vels = [[0.0, 0, 0], [0.2, 0, 0], [0.25, 0, 0.02]][:nt]
case = dict(cfg=cfg, obst=base["obst"], starts=starts, goals=[list(gl)] * nt, start_vels=vels, via=base.get("via"))
if base.get("initial_plan") is not None:
    px, py, pyaw = base["initial_plan"]
    case["plans"] = []
    for k in range(nt):
        x = px.copy(); y = py.copy(); x[0], y[0] = starts[k][0], starts[k][1]
        case["plans"].append((x, y, pyaw.copy()))
print("running which", which, "ticks", nt, flush=True)
r = T._hcp_ticks(which, case, slots=10, jacobian_mode=_abi.JACOBIAN_G2O_NUMERIC)
for t, x in enumerate(r):
    print("tick", t, [len(b[0]) for b in x["bands"]], x["best"], x["initial"], x["costs"])
c = cfg
print("cfg:", dict(dyn=c.obstacles.include_dynamic_obstacles, simple=c.hcp.simple_exploration, maxc=c.hcp.max_number_classes, inbest=c.hcp.max_number_plans_in_current_class, plan=base.get("initial_plan") is not None, allvia=c.hcp.viapoints_all_candidates, min_samples=c.trajectory.min_samples, M=len(base["obst"])))
