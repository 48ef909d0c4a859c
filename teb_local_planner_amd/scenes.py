"""Deterministic synthetic scenes for the five BASELINE.json configs (recipe: SURVEY.md §8d).

All randomness comes from numpy.random.default_rng(seed) (PCG64). A scene is
(TebConfig, ObstacleTable, via_points[(x,y)...], TebBatchHost).
"""
import math

import numpy as np

from . import _abi
from .config import TebConfig, RobotFootprintModel, normalize_theta


def _band_from_path(px, py, max_vel_x, theta_goal=None):
    """theta from finite differences, dt_i = ||dS_i|| / max_vel_x."""
    n = len(px)
    th = np.zeros(n)
    for i in range(n - 1):
        th[i] = math.atan2(py[i + 1] - py[i], px[i + 1] - px[i])
    th[n - 1] = th[n - 2] if theta_goal is None else theta_goal
    th = np.array([normalize_theta(t) for t in th])
    ds = np.hypot(np.diff(px), np.diff(py))
    dt = ds / max_vel_x
    return th, dt


def sine_band(n, length, amplitude, half_periods, max_vel_x):
    """Straight line from (0,0) to (length,0) plus a lateral sine perturbation; start/goal heading 0."""
    s = np.linspace(0.0, 1.0, n)
    px = length * s
    py = amplitude * np.sin(math.pi * half_periods * s)
    th, dt = _band_from_path(px, py, max_vel_x, theta_goal=0.0)
    th[0] = 0.0
    return px, py, th, dt


def _min_dist_to_path(p, px, py):
    return float(np.min(np.hypot(px - p[0], py - p[1])))


def scene_c1(with_velocities=False):
    """test_optim_node scene: src/test_optim_node.cpp:106-117,168 (static variant by default)."""
    cfg = TebConfig()
    obst = _abi.ObstacleTable()
    # test_optim_node.cpp:113-117 gives the first two obstacles a velocity (-> dynamic obstacles)
    vels = [(0.1, -0.3), (-0.3, -0.2), None] if with_velocities else [None, None, None]
    for (x, y), v in zip([(-3.0, 1.0), (6.0, 2.0), (0.0, 0.1)], vels):
        obst.add_point(x, y, vel=v)
    n = 50
    px = np.linspace(-4.0, 4.0, n)
    py = np.zeros(n)
    th = np.zeros(n)
    dt = np.full(n - 1, (8.0 / (n - 1)) / cfg.robot.max_vel_x)
    batch = _abi.TebBatchHost(1, 128)
    batch.set_teb(0, px, py, th, dt)
    batch.has_vel_goal[0] = 1  # plan() re-enables vel_goal_ unless free_goal_vel (optimal_planner.cpp:273-276)
    return cfg, obst, [], batch


def _point_obstacles(rng, count, xr, yr, paths, clearance):
    out = []
    while len(out) < count:
        p = (rng.uniform(*xr), rng.uniform(*yr))
        if all(_min_dist_to_path(p, px, py) >= clearance for (px, py) in paths):
            out.append(p)
    return out


def scene_c2(n=200, M=100, seed=1002, stride=None, length=20.0):
    cfg = TebConfig()
    rng = np.random.default_rng(seed)
    px, py, th, dt = sine_band(n, length, 0.5, 3.0, cfg.robot.max_vel_x)  # 1.5 periods = 3 half periods
    obst = _abi.ObstacleTable()
    for p in _point_obstacles(rng, M, (1.0, length - 1.0), (-3.0, 3.0), [(px, py)], 0.3):
        obst.add_point(*p)
    batch = _abi.TebBatchHost(1, stride or n)
    batch.set_teb(0, px, py, th, dt)
    batch.has_vel_goal[0] = 1
    return cfg, obst, [], batch


def _multi_band_scene(B, n, length, M_static, M_dyn, yr, seed, amp_range=(-2.0, 2.0), stride=None):
    cfg = TebConfig()
    rng = np.random.default_rng(seed)
    batch = _abi.TebBatchHost(B, stride or n)
    paths = []
    for b in range(B):
        amp = rng.uniform(*amp_range)
        hp = int(rng.integers(1, 4))  # 1..3 half periods
        px, py, th, dt = sine_band(n, length, amp, float(hp), cfg.robot.max_vel_x)
        batch.set_teb(b, px, py, th, dt)
        batch.has_vel_goal[b] = 1
        paths.append((px, py))
    obst = _abi.ObstacleTable()
    for p in _point_obstacles(rng, M_static, (1.0, length - 1.0), yr, [], 0.0):
        obst.add_point(*p)
    for p in _point_obstacles(rng, M_dyn, (1.0, length - 1.0), yr, [], 0.0):
        obst.add_point(p[0], p[1], vel=(rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5)))
    return cfg, obst, [], batch


def scene_c3(B=64, n=150, M=200, seed=1003, stride=None):
    return _multi_band_scene(B, n, 15.0, M, 0, (-4.0, 4.0), seed, stride=stride)


def scene_c4(B=256, n=200, M_static=450, M_dyn=50, seed=1004, stride=None, length=20.0):
    cfg, obst, via, batch = _multi_band_scene(B, n, length, M_static, M_dyn, (-5.0, 5.0), seed, stride=stride)
    cfg.obstacles.include_dynamic_obstacles = True
    return cfg, obst, via, batch


def scene_c4_via(B=256, n=200, seed=1004, stride=None, length=20.0):
    """The C4 batch with one configuration flag off the TebConfig defaults: via-points (EdgeViaPoint, src/optimal_planner.cpp:675-718) -
    three of them along the corridor, weight_viapoint = 1, enabled on every candidate. Used by bench.py (secondary.c4_with_via_points)
    and the configuration-profile tests: the kernels specialised on the defaults do not fold this flag."""
    cfg, obst, via, batch = scene_c4(B=B, n=n, seed=seed, stride=stride, length=length)
    cfg.optim.weight_viapoint = 1.0
    via = [(0.25 * length, 0.3), (0.5 * length, -0.2), (0.75 * length, 0.25)]
    batch.via_points_enabled[:] = 1
    return cfg, obst, via, batch


def scene_c4_flag(B=256, n=200, seed=1004, stride=None, length=20.0):
    """The C4 batch with a cost-term flag off the TebConfig defaults that only the *_LIGHT kinds (and the generic kernel) take:
    weight_shortest_path = 1 (EdgeShortestPath)."""
    cfg, obst, via, batch = scene_c4(B=B, n=n, seed=seed, stride=stride, length=length)
    cfg.optim.weight_shortest_path = 1.0
    return cfg, obst, via, batch


def scene_c5(n=300, M=300, seed=1005, stride=None, length=30.0):
    cfg = TebConfig()
    cfg.robot.min_turning_radius = 1.0
    cfg.optim.weight_kinematics_turning_radius = 1.0
    cfg.robot.max_vel_x_backwards = 0.2
    cfg.robot_model = RobotFootprintModel.polygon([(-0.3, -0.25), (0.9, -0.25), (0.9, 0.25), (-0.3, 0.25)])
    rng = np.random.default_rng(seed)
    px, py, th, dt = sine_band(n, length, 0.5, 3.0, cfg.robot.max_vel_x)
    obst = _abi.ObstacleTable()
    count = 0
    while count < M:
        c = (rng.uniform(2.0, length - 2.0), rng.uniform(-5.0, 5.0))
        if _min_dist_to_path(c, px, py) < 0.8 + 0.5:
            continue
        k = int(rng.integers(3, 7))
        r = rng.uniform(0.15, 0.5)
        rot = rng.uniform(0.0, 2 * math.pi)
        verts = [(c[0] + r * math.cos(rot + 2 * math.pi * j / k), c[1] + r * math.sin(rot + 2 * math.pi * j / k))
                 for j in range(k)]
        obst.add_polygon(verts)
        count += 1
    batch = _abi.TebBatchHost(1, stride or n)
    batch.set_teb(0, px, py, th, dt)
    batch.has_vel_goal[0] = 1
    return cfg, obst, [], batch


def scene_small_mixed(seed=7, B=3, n=24, stride=96, footprint="point", with_dynamic=True, with_via=True):
    """Small mixed scene touching every obstacle type; used by parity tests (seconds on the oracle)."""
    cfg = TebConfig()
    rng = np.random.default_rng(seed)
    if footprint == "circular":
        cfg.robot_model = RobotFootprintModel.circular(0.2)
    elif footprint == "two_circles":
        cfg.robot_model = RobotFootprintModel.two_circles(0.3, 0.2, 0.15, 0.25)
    elif footprint == "line":
        cfg.robot_model = RobotFootprintModel.line((-0.2, 0.0), (0.4, 0.0))
    elif footprint == "polygon":
        cfg.robot_model = RobotFootprintModel.polygon([(-0.2, -0.15), (0.4, -0.15), (0.4, 0.15), (-0.2, 0.15)])
    batch = _abi.TebBatchHost(B, stride)
    L = 6.0
    for b in range(B):
        px, py, th, dt = sine_band(n, L, rng.uniform(-0.8, 0.8), float(rng.integers(1, 3)), cfg.robot.max_vel_x)
        batch.set_teb(b, px, py, th, dt)
        batch.has_vel_goal[b] = 1
        batch.has_vel_start[b] = 1
        batch.vel_start[b] = (0.1, 0.0, 0.05)
    obst = _abi.ObstacleTable()
    obst.add_point(1.5, 0.45)
    obst.add_point(3.2, -0.6)
    obst.add_circle(2.4, 0.9, 0.2)
    obst.add_line(4.0, 0.7, 4.8, 1.2)
    obst.add_pill(1.0, -0.9, 1.8, -1.1, 0.15)
    obst.add_polygon([(4.5, -0.5), (5.1, -0.8), (5.0, -0.2)])
    obst.add_polygon([(3.0, 1.2), (3.4, 1.2), (3.4, 1.6), (3.0, 1.6)])
    if with_dynamic:
        obst.add_point(2.0, -1.5, vel=(0.05, 0.12))
        obst.add_circle(5.0, 1.5, 0.15, vel=(-0.1, -0.1))
        obst.add_line(0.5, 1.5, 1.0, 1.2, vel=(0.08, -0.05))
        obst.add_polygon([(3.8, -1.6), (4.2, -1.6), (4.0, -1.2)], vel=(0.0, 0.1))
    via = [(2.0, 0.2), (4.0, -0.1)] if with_via else []
    return cfg, obst, via, batch
