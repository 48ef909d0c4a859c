"""SURVEY 8f row f4 (arithmetic part): the oracle's isTrajectoryFeasible against the reference's own function compiled in place
(oracle/_ref, src/optimal_planner.cpp:1250-1308) - bit for bit: same verdict and the same number of footprint tests before a failure -
plus known answers of the footprint rasterisation. base_local_planner (the costmap model) is an absent dependency: both sides use the
grid restatement of oracle/grid_costmap.h, so this pins the reference's look-ahead / interpolation logic, not the navigation stack."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from feasibility_cases import feasibility_case, FOOTPRINTS  # noqa: E402
from oracle import oracle_py, ref_py  # noqa: E402
from oracle.oracle_py import Costmap  # noqa: E402

GOLDEN = os.path.join(HERE, "golden", "ref_f4_feasibility.npz")


def test_footprint_cost_known_answers():
    cells = np.zeros((40, 60), np.uint8)
    cm = Costmap(cells, 0.1, 0.0, 0.0)
    sq = [(-0.2, -0.2), (0.2, -0.2), (0.2, 0.2), (-0.2, 0.2)]
    assert oracle_py.footprint_cost(cm, 3.0, 2.0, 0.3, sq) == 0.0
    cells[20, 32] = 77                       # under the right edge of the square at theta = 0
    assert oracle_py.footprint_cost(cm, 3.0, 2.0, 0.0, sq) == 77.0
    cells[20, 32] = 254
    assert oracle_py.footprint_cost(cm, 3.0, 2.0, 0.0, sq) == -1.0
    cells[20, 32] = 255
    assert oracle_py.footprint_cost(cm, 3.0, 2.0, 0.0, sq) == -2.0
    cells[20, 32] = 254
    assert oracle_py.footprint_cost(cm, 3.0, 2.0, 0.0, [(0.0, 0.0)]) == 0.0      # a point robot only looks at its own cell
    cells[20, 30] = 253
    assert oracle_py.footprint_cost(cm, 3.0, 2.0, 0.0, [(0.0, 0.0)]) == -1.0     # ... where "inscribed" is lethal too
    assert oracle_py.footprint_cost(cm, 3.0, 2.0, 0.0, sq) == -1.0
    assert oracle_py.footprint_cost(cm, 5.95, 2.0, 0.0, sq) == -3.0              # a vertex off the map
    assert oracle_py.footprint_cost(cm, -0.1, 2.0, 0.0, sq) == -3.0
    cells[:] = 0
    cells[18:23, 28:33] = 254                # a lethal blob strictly inside the outline is NOT seen (only the outline is rasterised)
    big = [(-0.6, -0.6), (0.6, -0.6), (0.6, 0.6), (-0.6, 0.6)]
    assert oracle_py.footprint_cost(cm, 3.0, 2.0, 0.0, big) == 0.0


@pytest.mark.skipif(not ref_py.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(60))
def test_oracle_equals_reference_code(seed):
    batch, cm, fp, inscribed, ang, look, dist = feasibility_case(seed)
    got = oracle_py.is_trajectory_feasible(batch, 0, cm, fp, inscribed, ang, look, dist)
    want = ref_py.is_trajectory_feasible(batch, 0, cm, fp, inscribed, ang, look, dist)
    assert got == want, (seed, got, want)


def test_oracle_matches_committed_reference_vectors():
    g = np.load(GOLDEN)
    seen = set()
    for k, seed in enumerate(g["seed"]):
        batch, cm, fp, inscribed, ang, look, dist = feasibility_case(int(seed))
        got = oracle_py.is_trajectory_feasible(batch, 0, cm, fp, inscribed, ang, look, dist)
        assert got == (bool(g["feasible"][k]), int(g["first_infeasible"][k])), seed
        seen.add(got[0])
    assert seen == {True, False}             # the vectors hold both verdicts
