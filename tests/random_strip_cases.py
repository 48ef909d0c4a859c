"""Random inputs for the producers / consumers of the state strip (rows f1, f2), shared by the CPU pin and the GPU parity test."""
import numpy as np

from teb_local_planner_amd import _abi
from teb_local_planner_amd.config import TebConfig


def random_strip_case(seed):
    rng = np.random.default_rng(5000 + seed)
    ang = rng.uniform(-np.pi, np.pi); L = rng.uniform(0.3, 9.0)
    wrap = lambda a: float((a + np.pi) % (2 * np.pi) - np.pi)   # angles as a PoseSE2 built from a pose message carries them: [-pi, pi)
    start = [float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), wrap(ang + rng.uniform(-3.0, 3.0))]
    goal = [start[0] + L * np.cos(ang), start[1] + L * np.sin(ang), float(rng.uniform(-np.pi, np.pi))]
    max_vel_x = float(rng.uniform(0.2, 1.0)); max_vel_theta = float(rng.choice([0.0, 0.3, 1.0])); acc = float(rng.uniform(0.2, 1.5))
    min_samples = int(rng.integers(3, 9)); backwards = bool(rng.integers(2))
    k = int(rng.integers(2, 40)); side = rng.uniform(-1.5, 1.5)
    t = np.linspace(0, 1, k)
    px = start[0] + t * L * np.cos(ang) - side * np.sin(np.pi * t) * np.sin(ang) + rng.normal(0, 0.02, k)
    py = start[1] + t * L * np.sin(ang) + side * np.sin(np.pi * t) * np.cos(ang) + rng.normal(0, 0.02, k)
    if rng.random() < 0.2 and k > 3 and max_vel_theta > 0:
        px[2], py[2] = px[1], py[1]   # a repeated plan point (with max_vel_theta = 0 its time difference is 0: the reference asserts dt > 0)
    pyaw = rng.uniform(-np.pi, np.pi, k)
    line = (start, goal, float(rng.choice([0.0, 0.1, 0.37])), max_vel_x, min_samples, backwards)
    plan = (px, py, pyaw, max_vel_x, max_vel_theta, bool(rng.integers(2)), min_samples, backwards)
    path = (px, py, max_vel_x, max_vel_theta, acc if rng.random() < 0.7 else None, start[2] if rng.random() < 0.7 else None,
            goal[2] if rng.random() < 0.7 else None, min_samples, backwards)
    # a band to prune / read out: the path band with noisy time differences
    cfg = TebConfig()
    cfg.robot.max_vel_y = float(rng.choice([0.0, 0.3]))
    cfg.trajectory.min_samples = min_samples
    new_start = [float(px[min(2, k - 1)] + rng.normal(0, 0.05)), float(py[min(2, k - 1)] + rng.normal(0, 0.05)), float(rng.uniform(-np.pi, np.pi))]
    new_goal = [goal[0] + float(rng.normal(0, 0.3)), goal[1] + float(rng.normal(0, 0.3)), float(rng.uniform(-np.pi, np.pi))]
    prune = (new_start if rng.random() < 0.8 else None, new_goal if rng.random() < 0.8 else None, min_samples)
    consumer = (int(rng.choice([1, 2, 5, 100])), int(rng.choice([0, 2])), [float(rng.uniform(-0.3, 0.5)), float(rng.uniform(-0.1, 0.1)), float(rng.uniform(-0.3, 0.3))],
                [float(rng.uniform(0, 0.3)), 0.0, float(rng.uniform(-0.1, 0.1))])
    return dict(line=line, plan=plan, path=path, prune=prune, consumer=consumer, cfg=cfg, dt_noise=rng.uniform(0.5, 1.5, 4096))


def band_for_readout(oracle, case, stride=512):
    """A one-band batch made from the path case (time differences scaled by noise), with start / goal velocities set."""
    x, y, th, dt = oracle.init_trajectory_path(*case["path"])
    dt = dt * case["dt_noise"][:len(dt)]
    b = _abi.TebBatchHost(1, stride)
    b.set_teb(0, x, y, th, dt)
    b.vel_start[0] = case["consumer"][2]; b.vel_goal[0] = case["consumer"][3]
    return b
