"""Randomised scenes for the candidate generation (row f3): every obstacle class, both graph types, random planner parameters, optionally
an initial plan. Shared by the CPU pin against the reference (tests/test_reference_pinning.py) and the GPU parity test."""
import numpy as np

from teb_local_planner_amd import _abi
from teb_local_planner_amd.config import TebConfig


def random_explore_case(seed):
    rng = np.random.default_rng(1000 + seed)
    cfg = TebConfig()
    cfg.obstacles.include_dynamic_obstacles = bool(rng.integers(2))
    h = cfg.hcp
    h.simple_exploration = bool(rng.integers(2))
    h.max_number_classes = int(rng.integers(2, 8))
    h.max_number_plans_in_current_class = int(rng.integers(1, 3))
    h.obstacle_heading_threshold = float(rng.choice([0.0, 0.2, 0.45, 0.7]))
    h.roadmap_graph_no_samples = int(rng.integers(6, 22))
    h.roadmap_graph_area_width = float(rng.uniform(3.0, 7.0))
    h.roadmap_graph_area_length_scale = float(rng.choice([1.0, 1.0, 0.8, 1.2]))
    h.h_signature_prescaler = float(rng.choice([1.0, 0.7]))
    h.viapoints_all_candidates = bool(rng.integers(2))
    cfg.trajectory.allow_init_with_backwards_motion = bool(rng.integers(2))
    cfg.trajectory.global_plan_overwrite_orientation = bool(rng.integers(2))
    cfg.trajectory.min_samples = int(rng.integers(3, 6))
    cfg.robot.max_vel_x = float(rng.uniform(0.3, 0.8)); cfg.robot.acc_lim_x = float(rng.uniform(0.3, 1.0))
    ang = rng.uniform(-np.pi, np.pi); L = rng.uniform(4.0, 10.0)
    start = [float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), float(ang + rng.uniform(-1.2, 1.2))]
    goal = [start[0] + L * np.cos(ang), start[1] + L * np.sin(ang), float(ang + rng.uniform(-0.8, 0.8))]
    ob = _abi.ObstacleTable()
    M = int(rng.integers(2, 9 if h.simple_exploration else 14))
    for _ in range(M):
        u, w = rng.uniform(-0.1, 1.1), rng.uniform(-2.2, 2.2)
        cx = start[0] + u * L * np.cos(ang) - w * np.sin(ang); cy = start[1] + u * L * np.sin(ang) + w * np.cos(ang)
        vel = (float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2))) if rng.random() < 0.25 else None
        kind = rng.integers(5)
        if kind == 0:
            ob.add_point(cx, cy, vel=vel)
        elif kind == 1:
            ob.add_circle(cx, cy, float(rng.uniform(0.1, 0.5)), vel=vel)
        elif kind == 2:
            a = rng.uniform(0, np.pi); r = rng.uniform(0.2, 0.8)
            ob.add_line(cx - r * np.cos(a), cy - r * np.sin(a), cx + r * np.cos(a), cy + r * np.sin(a), vel=vel)
        elif kind == 3:
            a = rng.uniform(0, np.pi); r = rng.uniform(0.2, 0.6)
            ob.add_pill(cx - r * np.cos(a), cy - r * np.sin(a), cx + r * np.cos(a), cy + r * np.sin(a), float(rng.uniform(0.1, 0.3)), vel=vel)
        else:
            k = int(rng.integers(3, 6)); aa = np.sort(rng.uniform(0, 2 * np.pi, k)); rr = rng.uniform(0.2, 0.6, k)
            ob.add_polygon([(cx + rr[i] * np.cos(aa[i]), cy + rr[i] * np.sin(aa[i])) for i in range(k)], vel=vel)
    case = dict(cfg=cfg, obst=ob, batch=None, best=-1, start=start, goal=goal, dist_to_obst=float(rng.uniform(0.2, 0.6)))
    if rng.random() < 0.4:
        case["start_vel"] = [float(rng.uniform(0, 0.3)), 0.0, float(rng.uniform(-0.2, 0.2))]; case["free_goal_vel"] = bool(rng.integers(2))
    if rng.random() < 0.5:
        k = int(rng.integers(4, 12)); side = rng.uniform(-1.5, 1.5)
        t = np.linspace(0, 1, k)
        px = start[0] + t * L * np.cos(ang) - side * np.sin(np.pi * t) * np.sin(ang)
        py = start[1] + t * L * np.sin(ang) + side * np.sin(np.pi * t) * np.cos(ang)
        yaw = np.arctan2(np.gradient(py), np.gradient(px)); yaw[0], yaw[-1] = start[2], goal[2]
        case["initial_plan"] = (px, py, yaw)
        case["via"] = [(float(px[k // 2]), float(py[k // 2]))]
    return case
