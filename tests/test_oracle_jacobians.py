"""The oracle's two Jacobian paths (g2o central differences, delta=1e-9 vs closed form) must agree.

This is check (1) of SURVEY.md §8c "how the oracle earns trust without goldens". H and b are compared in
the canonical index space for every edge family / footprint / obstacle type; tolerance = the noise floor
of the delta=1e-9 central difference (eps_mach/delta ~ 1e-7 relative) with head-room.
"""
import numpy as np
import pytest

from teb_local_planner_amd import scenes, _abi
from teb_local_planner_amd.config import RobotFootprintModel


def _cmp(oracle, cfg, obst, via, batch, b=0, wm=1.0, rtol=2e-6):
    cfg.jacobian_mode = _abi.JACOBIAN_ANALYTIC
    A = oracle.linearize(cfg, obst, via, batch, b, wm)
    cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    N = oracle.linearize(cfg, obst, via, batch, b, wm)
    assert A["n_edges"] == N["n_edges"] and A["n_rows"] == N["n_rows"]
    np.testing.assert_allclose(A["chi2"], N["chi2"], rtol=0, atol=0)
    sH = np.abs(N["H"]).max()
    sb = np.abs(N["b"]).max()
    assert np.abs(A["H"] - N["H"]).max() <= rtol * sH, (np.abs(A["H"] - N["H"]).max(), sH)
    assert np.abs(A["b"] - N["b"]).max() <= rtol * sb, (np.abs(A["b"] - N["b"]).max(), sb)
    # symmetric, and zero rows for the fixed start / goal pose
    np.testing.assert_array_equal(A["H"], A["H"].T)
    n = int(batch.n[b])
    for r in (0, 1, 2, 4 * (n - 1), 4 * (n - 1) + 1, 4 * (n - 1) + 2, 4 * (n - 1) + 3):
        assert not A["H"][r].any() and A["b"][r] == 0
    return A


@pytest.mark.parametrize("footprint", ["point", "circular", "two_circles", "line", "polygon"])
def test_mixed_scene_all_footprints(oracle, footprint):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint=footprint)
    for b in range(batch.count):
        A = _cmp(oracle, cfg, obst, via, batch, b)
        assert A["chi2"][0] > 0  # obstacle terms active


def test_carlike_exact_arc(oracle):
    for exact in (False, True):
        cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
        cfg.robot.min_turning_radius = 1.0
        cfg.trajectory.exact_arc_length = exact
        _cmp(oracle, cfg, obst, via, batch, 1)


def test_optional_edges(oracle):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="circular")
    cfg.optim.weight_shortest_path = 0.7
    cfg.optim.weight_velocity_obstacle_ratio = 3.0
    cfg.obstacles.obstacle_proximity_lower_bound = 0.1
    cfg.obstacles.obstacle_proximity_upper_bound = 1.2
    cfg.obstacles.obstacle_proximity_ratio_max_vel = 0.8
    cfg.optim.obstacle_cost_exponent = 1.7
    batch.prefer_rotdir[0] = _abi.ROT_LEFT
    batch.prefer_rotdir[1] = _abi.ROT_RIGHT
    for b in range(batch.count):
        _cmp(oracle, cfg, obst, via, batch, b, wm=2.0)


def test_holonomic(oracle):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="point")
    cfg.robot.max_vel_y = 0.3
    cfg.robot.max_vel_trans = 0.45
    cfg.robot.acc_lim_y = 0.4
    # make the lateral velocities non-trivial
    rng = np.random.default_rng(3)
    batch.theta += rng.uniform(-0.4, 0.4, batch.theta.shape)
    batch.dt *= 0.6
    for b in range(batch.count):
        _cmp(oracle, cfg, obst, via, batch, b)


def test_legacy_association(oracle):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="point")
    cfg.obstacles.legacy_obstacle_association = True
    cfg.obstacles.obstacle_poses_affected = 6
    _cmp(oracle, cfg, obst, via, batch, 0)


def test_full_optimize_modes_agree(oracle):
    """20 LM iterations in both Jacobian modes end at the same trajectory within the reference's own
    numeric-differentiation noise (1e-3 m / 1e-3 rad, chi2 1e-3 rel)."""
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    cfg.jacobian_mode = _abi.JACOBIAN_ANALYTIC
    a, ra = oracle.optimize_batch(cfg, obst, via, batch)
    cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    g, rg = oracle.optimize_batch(cfg, obst, via, batch)
    np.testing.assert_array_equal(a.n, g.n)
    np.testing.assert_array_equal(ra.status, rg.status)
    for b in range(batch.count):
        for u, v in zip(a.get_teb(b), g.get_teb(b)):
            assert np.abs(u - v).max() < 1e-3
    np.testing.assert_allclose(ra.chi2, rg.chi2, rtol=1e-3)
    np.testing.assert_allclose(ra.cost, rg.cost, rtol=1e-3)
