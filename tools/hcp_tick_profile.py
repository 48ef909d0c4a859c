"""Where a device-resident HomotopyClassPlanner::plan() tick spends its time (host wall clock per phase, p50 over the ticks).
Run on the GPU box: python tools/hcp_tick_profile.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import scenes, planner, _abi

hc = scenes.scene_c4(B=1, n=200)[0]
hc.hcp.max_number_classes = 5
hc.obstacles.include_dynamic_obstacles = os.environ.get("TICK_2D") is None      # TICK_2D=1: HSignature (2-D) instead of HSignature3d
rng = np.random.default_rng(5)
hob = _abi.ObstacleTable()
for _ in range(12):
    hob.add_point(rng.uniform(1.5, 14.5), rng.uniform(-2.5, 2.5))
ticks = 24
hp = planner.HomotopyClassPlanner(hc, hob, [], None, max_tebs=8, max_poses=int(os.environ.get("TICK_MAX_POSES", "224")))
s, h = hp.solver, hc.hcp
T = {k: [] for k in ("update", "signatures", "filter", "detours", "compact", "explore", "optimize", "select", "command", "total")}
for k in range(ticks):
    start, goal, sv = [0.05 * k, 0.0, 0.0], [16.0, 0.0, 0.0], [0.3, 0.0, 0.0]
    t0 = time.perf_counter()
    s.set_config(hc)
    hp.updateAllTEBs(start, goal, sv); t1 = time.perf_counter(); T["update"].append(t1 - t0)
    if s.count > 0:
        s.h_signatures(h.h_signature_prescaler, values=False); t2 = time.perf_counter(); T["signatures"].append(t2 - t1)
        keep, _, _ = s.filter_equivalence_classes(h.h_signature_threshold, hp.best_teb_, h.max_number_plans_in_current_class)
        t3 = time.perf_counter(); T["filter"].append(t3 - t2)
        keep = s.filter_detours(keep, hp.best_teb_); t4 = time.perf_counter(); T["detours"].append(t4 - t3)
        _, hp.best_teb_ = s.compact_bands(keep, hp.best_teb_); t5 = time.perf_counter(); T["compact"].append(t5 - t4)
    else:
        t5 = t1
    s.explore_candidates(start, goal, hc.obstacles.min_obstacle_dist, sv, False, hp.best_teb_); t6 = time.perf_counter(); T["explore"].append(t6 - t5)
    hp.optimizeAllTEBs(hc.optim.no_inner_iterations, hc.optim.no_outer_iterations); s.synchronize(); t7 = time.perf_counter(); T["optimize"].append(t7 - t6)
    hp.selectBestTeb(); t8 = time.perf_counter(); T["select"].append(t8 - t7)
    hp.getVelocityCommand(); t9 = time.perf_counter(); T["command"].append(t9 - t8)
    T["total"].append(t9 - t0)
for k, v in T.items():
    print("%-11s p50 %7.1f us   (n=%d)" % (k, 1e6 * float(np.median(v[2:] if len(v) > 4 else v)), len(v)))
print("kernel ms of the last optimise:", s.last_kernel_ms(), "bands", s.count, "poses", s.pose_counts())
