"""SURVEY 8f row f4 (arithmetic part) on the device: isTrajectoryFeasible of the resident bands against a costmap grid, through the
C-ABI, against the oracle (bit-equal to the reference's own function, tests/test_feasibility_oracle.py) and the committed reference
vectors: same verdict and the same index of the first colliding footprint test - integer / byte work, exact."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from feasibility_cases import feasibility_case  # noqa: E402
from teb_local_planner_amd import planner, scenes, _abi  # noqa: E402

pytestmark = pytest.mark.gpu


def _solver(batch):
    cfg = scenes.scene_c1()[0]
    return planner.make_solver(cfg, _abi.ObstacleTable(), [], batch)


@pytest.mark.parametrize("seed", range(60))
def test_device_matches_oracle_and_reference_vectors(oracle, seed):
    g = np.load(os.path.join(HERE, "golden", "ref_f4_feasibility.npz"))
    batch, cm, fp, inscribed, ang, look, dist = feasibility_case(seed)
    s = _solver(batch)
    s.set_costmap(cm.cells, cm.resolution, cm.origin_x, cm.origin_y)
    got = s.is_trajectory_feasible(0, fp, inscribed, ang, look, dist)
    s.close()
    assert got == oracle.is_trajectory_feasible(batch, 0, cm, fp, inscribed, ang, look, dist), seed
    k = int(np.where(g["seed"] == seed)[0][0])
    assert got == (bool(g["feasible"][k]), int(g["first_infeasible"][k])), seed


def test_whole_batch_in_one_launch_and_errors(oracle):
    cases = [feasibility_case(s) for s in (1, 4, 6, 9)]
    cm, fp, inscribed, ang = cases[0][1], cases[0][2], cases[0][3], cases[0][4]
    stride = max(int(c[0].n[0]) for c in cases)
    batch = _abi.TebBatchHost(len(cases), stride)
    for k, c in enumerate(cases):
        batch.set_teb(k, *c[0].get_teb(0))
    s = _solver(batch)
    with pytest.raises(planner.TebAmdError):
        s.is_trajectory_feasible(0, fp, inscribed)                       # no costmap yet: loud
    s.set_costmap(cm.cells, cm.resolution, cm.origin_x, cm.origin_y)
    ok, first = s.is_trajectory_feasible(-1, fp, inscribed, ang, -1, -1.0)
    for k in range(len(cases)):
        want = oracle.is_trajectory_feasible(batch, k, cm, fp, inscribed, ang, -1, -1.0)
        assert (bool(ok[k]), int(first[k])) == want
        assert s.is_trajectory_feasible(k, fp, inscribed, ang, -1, -1.0) == want
    with pytest.raises(planner.TebAmdError):
        s.is_trajectory_feasible(0, fp, 1e-12)                           # 1e12 samples per segment: refused, not attempted
    s.close()


def test_homotopy_planner_drops_infeasible_best_band(oracle):
    """HomotopyClassPlanner::isTrajectoryFeasible (src/homotopy_class_planner.cpp:686-709): free map -> the best band stands; a lethal
    blob on the best band only -> it is removed and the next best answers; lethal everywhere -> every band goes, False."""
    from oracle.oracle_py import Costmap
    cfg = scenes.scene_c1()[0]
    batch = _abi.TebBatchHost(3, 64)
    for k, amp in enumerate((1.5, 0.0, -1.5)):       # three well separated bands; equal (zero) costs: the lowest index is the best
        batch.set_teb(k, *scenes.sine_band(40, 8.0, amp, 1.0, cfg.robot.max_vel_x))
    fp = [(-0.2, -0.15), (0.3, -0.15), (0.3, 0.15), (-0.2, 0.15)]
    def planner_():
        hp = planner.HomotopyClassPlanner(cfg, _abi.ObstacleTable(), [], batch.copy())
        hp.selectBestTeb()
        return hp
    free = Costmap(np.zeros((400, 800), np.uint8), 0.05, -2.0, -10.0)
    hp = planner_()
    assert hp.best_teb_ == 0
    assert hp.isTrajectoryFeasible(free, fp, 0.15) and hp.solver.count == 3 and hp.best_teb_ == 0
    x, y, _, _ = batch.get_teb(0)
    i = int(np.argmax(np.abs(y)))                     # where band 0 is 1.5 m away from band 1
    cells = np.zeros((400, 800), np.uint8)
    mx, my = int((x[i] + 2.0) / 0.05), int((y[i] + 10.0) / 0.05)
    cells[my - 3:my + 4, mx - 3:mx + 4] = 254
    blocked = Costmap(cells, 0.05, -2.0, -10.0)
    assert not oracle.is_trajectory_feasible(batch, 0, blocked, fp, 0.15)[0]
    assert oracle.is_trajectory_feasible(batch, 1, blocked, fp, 0.15)[0]
    assert hp.isTrajectoryFeasible(blocked, fp, 0.15)
    assert hp.solver.count == 2 and hp.best_teb_ == 0
    np.testing.assert_array_equal(hp.bands()[0][1], batch.get_teb(1)[1])      # the former band 1 answers
    # the same band infeasible again, now as last tick's best band: False at once, no further candidates are tried (:703-704)
    x1, y1, _, _ = batch.get_teb(1)
    cells[:] = 0
    cells[int((y1[20] + 10.0) / 0.05) - 3:int((y1[20] + 10.0) / 0.05) + 4, int((x1[20] + 2.0) / 0.05) - 3:int((x1[20] + 2.0) / 0.05) + 4] = 254
    hp.selectBestTeb()                                # next tick: last_best_teb_ = this band
    assert not hp.isTrajectoryFeasible(Costmap(cells, 0.05, -2.0, -10.0), fp, 0.15)
    assert hp.solver.count == 1
    wall = Costmap(np.full((400, 800), 254, np.uint8), 0.05, -2.0, -10.0)
    hp2 = planner_()
    assert not hp2.isTrajectoryFeasible(wall, fp, 0.15) and hp2.solver.count == 0
    hp.solver.close(); hp2.solver.close()


def test_feasibility_and_signatures_of_a_band_beyond_512_poses(oracle):
    """Rows f3 / f4 at a pose capacity beyond 512 (round 5): a 700-pose band - isTrajectoryFeasible over its whole length (a lethal blob near
    pose 650: the first infeasible test lies beyond index 512), and both H-signatures against the oracle."""
    from oracle.oracle_py import Costmap
    n, length = 700, 175.0
    x, y, th, dt = scenes.sine_band(n, length, 0.3, 1.0, 0.4)
    batch = _abi.TebBatchHost(1, _abi.MAX_POSES)
    batch.set_teb(0, x, y, th, dt)
    res = 0.1
    ox, oy = -2.0, -6.0
    cells = np.zeros((int(12.0 / res), int((length + 4.0) / res)), np.uint8)
    fp = [(-0.3, -0.25), (0.9, -0.25), (0.9, 0.25), (-0.3, 0.25)]
    cm_free = Costmap(cells.copy(), res, ox, oy)
    blob = cells.copy()
    mx, my = int((x[650] - ox) / res), int((y[650] - oy) / res)
    blob[my - 2:my + 3, mx - 2:mx + 3] = 254
    cm_blob = Costmap(blob, res, ox, oy)
    cfg = scenes.scene_c1()[0]
    cfg.trajectory.max_samples = 900
    obst = _abi.ObstacleTable()
    rng = np.random.default_rng(5)
    for k in range(0, n, 23):
        obst.add_point(float(x[k]) + 0.1, float(y[k]) + float(rng.choice([-1.0, 1.0])) * float(rng.uniform(0.4, 1.5)))
    s = planner.make_solver(cfg, obst, [], batch)
    for cm in (cm_free, cm_blob):
        s.set_costmap(cm.cells, cm.resolution, cm.origin_x, cm.origin_y)
        got = s.is_trajectory_feasible(0, fp, 0.25, 0.3, -1, -1.0)
        want = oracle.is_trajectory_feasible(batch, 0, cm, fp, 0.25, 0.3, -1, -1.0)
        assert got == want, (got, want)
    assert want[0] is False and want[1] > 512
    for mode in (2, 3):
        cfg.obstacles.include_dynamic_obstacles = (mode == 3)
        s.set_config(cfg)
        sig = s.h_signatures(1.0)
        wsig = oracle.h_signatures(cfg, obst, batch, mode, 1.0)
        tol = 4 * np.finfo(float).eps * max(1.0, np.abs(wsig).max()) if mode == 3 else 1e-10 * max(np.abs(wsig).max(), 1e-300)
        assert np.abs(sig - wsig).max() <= tol, (mode, np.abs(sig - wsig).max())
    s.close()
