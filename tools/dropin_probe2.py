import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np
import test_reference_backend as T
from random_explore_cases import random_explore_case
from teb_local_planner_amd import _abi
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
base = random_explore_case(seed)
cfg = base["cfg"]
cfg.optim.no_inner_iterations = 3; cfg.optim.no_outer_iterations = 2
st, gl = np.array(base["start"]), np.array(base["goal"])
d = (gl[:2] - st[:2]) / np.linalg.norm(gl[:2] - st[:2])
starts = [[st[0] + 0.1 * k * d[0], st[1] + 0.1 * k * d[1], st[2]] for k in range(3)]
case = dict(cfg=cfg, obst=base["obst"], starts=starts, goals=[list(gl)] * 3, start_vels=[[0.0, 0, 0], [0.2, 0, 0], [0.25, 0, 0.02]], via=base.get("via"))
if base.get("initial_plan") is not None:
    px, py, pyaw = base["initial_plan"]
    case["plans"] = []
    for k in range(3):
        x = px.copy(); y = py.copy(); x[0], y[0] = starts[k][0], starts[k][1]
        case["plans"].append((x, y, pyaw.copy()))
ref = T._hcp_ticks(0, case, slots=10); amd = T._hcp_ticks(1, case, slots=10, jacobian_mode=_abi.JACOBIAN_G2O_NUMERIC)
for t, (r, a) in enumerate(zip(ref, amd)):
    print("tick", t, "ref n", [len(b[0]) for b in r["bands"]], "amd n", [len(b[0]) for b in a["bands"]], "best", r["best"], a["best"], "initial", r["initial"], a["initial"])
    print("   ref cost", r["costs"], "\n   amd cost", a["costs"])
    for k, (u, v) in enumerate(zip(a["bands"], r["bands"])):
        if len(u[0]) == len(v[0]):
            print("   band", k, [float(np.abs(x - y).max()) for x, y in zip(u, v)])
        print("      ref y mid", v[1][len(v[1]) // 2], "amd y mid", u[1][len(u[1]) // 2])
c = cfg
print("cfg:", dict(dyn=c.obstacles.include_dynamic_obstacles, simple=c.hcp.simple_exploration, maxc=c.hcp.max_number_classes, inbest=c.hcp.max_number_plans_in_current_class, plan=base.get("initial_plan") is not None, allvia=c.hcp.viapoints_all_candidates, via=base.get("via"), detours=c.hcp.delete_detours_backwards))
