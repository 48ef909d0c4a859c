#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the ORACLE (analytic-Jacobian mode).

The reference publishes no golden vectors for optimizeTEB and cannot be built here (ROS/Eigen/g2o absent),
so these fixtures are regression pins of the oracle itself: the CPU suite checks the oracle still reproduces
them, the GPU suite checks the HIP path against the same numbers. Re-generate only on purpose:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from teb_local_planner_amd import scenes, _abi  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

CASES = {
    "c1_test_optim_node": lambda: scenes.scene_c1(),
    "c1_dynamic": lambda: scenes.scene_c1(with_velocities=True),
    "mixed_point": lambda: scenes.scene_small_mixed(footprint="point"),
    "mixed_polygon": lambda: scenes.scene_small_mixed(footprint="polygon"),
    "mixed_two_circles": lambda: scenes.scene_small_mixed(footprint="two_circles"),
    "c2_small": lambda: scenes.scene_c2(n=60, M=40, stride=128, length=8.0),
    "c5_small": lambda: scenes.scene_c5(n=50, M=40, stride=128, length=10.0),
}


def pack(batch, res):
    d = dict(n=batch.n.copy(), status=res.status.copy(), lm_iterations=res.lm_iterations.copy(),
             lm_trials=res.lm_trials.copy(), chi2=res.chi2.copy(), cost=res.cost.copy())
    for b in range(batch.count):
        x, y, th, dt = batch.get_teb(b)
        d["x%d" % b], d["y%d" % b], d["th%d" % b], d["dt%d" % b] = x, y, th, dt
    return d


def main():
    O.build()
    for name, mk in CASES.items():
        cfg, obst, via, batch = mk()
        cfg.jacobian_mode = _abi.JACOBIAN_ANALYTIC
        out, res = O.optimize_batch(cfg, obst, via, batch)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **pack(out, res))
        print(name, out.n, res.lm_trials, res.cost)


if __name__ == "__main__":
    main()
