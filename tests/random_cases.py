"""Randomised small scenes with every option of the path toggled at random (shared by the GPU parity test and the reference pin)."""
import numpy as np

from teb_local_planner_amd import scenes


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    fp = ["point", "circular", "two_circles", "line", "polygon"][seed % 5]
    cfg, obst, via, batch = scenes.scene_small_mixed(seed=seed, B=3, n=int(rng.integers(12, 40)), stride=320, footprint=fp,
                                                     with_dynamic=bool(rng.integers(0, 2)), with_via=bool(rng.integers(0, 2)))
    if rng.random() < 0.3:
        cfg.robot.max_vel_y = 0.3; cfg.robot.acc_lim_y = 0.4; cfg.robot.max_vel_trans = 0.5
        # the generated bands move at exactly max_vel_x, and EdgeVelocityHolonomic bounds vx with epsilon = 0
        # (edge_velocity.h:265): a residual sitting on that kink flips with the last bit of sin / cos. Move off the kink.
        batch.dt *= float(rng.choice([0.7, 1.3]))
        # ... and off the second one: headings exactly along the motion make the lateral velocity +-1e-17, whose sign then
        # decides the branch of pI(vy, 0, 0) when the forward speed exceeds max_vel_trans (remaining lateral budget 0)
        batch.theta += rng.uniform(-0.3, 0.3, batch.theta.shape)
    if rng.random() < 0.3:
        cfg.robot.min_turning_radius = float(rng.uniform(0.3, 1.0))
    if rng.random() < 0.3:
        cfg.trajectory.exact_arc_length = True
    if rng.random() < 0.25:
        cfg.obstacles.legacy_obstacle_association = True
        cfg.obstacles.obstacle_poses_affected = int(rng.integers(1, 12))
    if rng.random() < 0.3:
        cfg.optim.obstacle_cost_exponent = float(rng.uniform(1.2, 3.0))
    if rng.random() < 0.3:
        cfg.optim.weight_shortest_path = float(rng.uniform(0.1, 2.0))
    if rng.random() < 0.3:
        cfg.optim.weight_velocity_obstacle_ratio = float(rng.uniform(0.5, 5.0))
        cfg.obstacles.obstacle_proximity_upper_bound = 1.0
    if rng.random() < 0.3:
        cfg.obstacles.inflation_dist = 0.4          # below min_obstacle_dist: EdgeObstacle instead of EdgeInflatedObstacle
    if rng.random() < 0.3:
        cfg.trajectory.via_points_ordered = True
    if rng.random() < 0.3:
        cfg.hcp.selection_alternative_time_cost = True
    if rng.random() < 0.3:
        cfg.recovery.divergence_detection_enable = True
    cfg.trajectory.dt_ref = float(rng.choice([0.2, 0.3, 0.45]))
    cfg.trajectory.dt_hysteresis = cfg.trajectory.dt_ref / 3
    cfg.optim.weight_adapt_factor = float(rng.choice([1.0, 2.0, 3.0]))
    # generated headings lie exactly along the segments: the non-holonomic residual of collinear stretches is then 0 or +-1e-17,
    # and the reference's Jacobian convention sign(0) = 0 (edge_kinematics.h:112-149) turns that last bit into a rank-1 change of H
    # (measured: seed 65, tools/fuzz_probe*.py). Real bands are never exactly straight; neither are these.
    batch.theta += rng.normal(0.0, 2e-3, batch.theta.shape)
    batch.prefer_rotdir[:] = rng.integers(0, 3, batch.count)
    batch.has_vel_goal[:] = rng.integers(0, 2, batch.count)
    batch.vel_start[:] = rng.uniform(-0.2, 0.3, (batch.count, 3))
    return cfg, obst, via, batch
