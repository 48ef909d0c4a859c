import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np
import test_reference_backend as T
from random_explore_cases import random_explore_case
from teb_local_planner_amd import _abi
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def run(mod):
    base = random_explore_case(seed)
    cfg = base["cfg"]
    cfg.optim.no_inner_iterations = 3; cfg.optim.no_outer_iterations = 2
    st, gl = np.array(base["start"]), np.array(base["goal"])
    case = dict(cfg=cfg, obst=base["obst"], starts=[list(st)], goals=[list(gl)], start_vels=[[0.0, 0, 0]], via=base.get("via"))
    if base.get("initial_plan") is not None:
        case["plans"] = [base["initial_plan"]]
    mod(case, base)
    ref = T._hcp_ticks(0, case, slots=10); amd = T._hcp_ticks(1, case, slots=10, jacobian_mode=_abi.JACOBIAN_G2O_NUMERIC)
    out = []
    for k, (u, v) in enumerate(zip(amd[0]["bands"], ref[0]["bands"])):
        out.append((len(u[0]), len(v[0]), max(np.abs(x - y).max() for x, y in zip(u, v)) if len(u[0]) == len(v[0]) else None))
    return len(amd[0]["bands"]), len(ref[0]["bands"]), out
def nop(c, b): pass
def no_opt(c, b): c["cfg"].optim.no_outer_iterations = 0
def no_via(c, b): c["via"] = None
def no_plan(c, b): c.pop("plans", None)
def no_autosize(c, b): c["cfg"].trajectory.teb_autosize = False
def no_dyn(c, b): c["cfg"].obstacles.include_dynamic_obstacles = False
def one_outer(c, b): c["cfg"].optim.no_outer_iterations = 1; c["cfg"].optim.no_inner_iterations = 1
for name, m in (("as is", nop), ("no optimisation", no_opt), ("no via", no_via), ("no plan", no_plan), ("no autosize", no_autosize), ("2-D classes, static association", no_dyn), ("1x1 iteration", one_outer)):
    print(name, run(m))
b = random_explore_case(seed); c = b["cfg"]
print("cfg:", dict(dyn=c.obstacles.include_dynamic_obstacles, simple=c.hcp.simple_exploration, min_samples=c.trajectory.min_samples, backwards=c.trajectory.allow_init_with_backwards_motion,
                   vmax=c.robot.max_vel_x, acc=c.robot.acc_lim_x, plan=b.get("initial_plan") is not None, via=b.get("via"), types=list(b["obst"].type), dynflags=list(b["obst"].dynamic)))
