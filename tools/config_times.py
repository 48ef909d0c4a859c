#!/usr/bin/env python3
"""Kernel time of one optimizeAllTEBs (4 outer x 5 inner, cost on) for the five BASELINE.json configurations at full size.
autoResize is on except for C4, whose bands would outgrow the 208-pose capacity that still leaves room for the 500-obstacle LDS cache."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teb_local_planner_amd import planner, scenes  # noqa: E402

CASES = [("C1 test_optim_node, 1 x 50 poses, 3 point obstacles", lambda: scenes.scene_c1()),
         ("C2 1 x 200 poses, 100 point obstacles", lambda: scenes.scene_c2(stride=232)),
         ("C3 64 x 150 poses, 200 obstacles", lambda: scenes.scene_c3(stride=208)),
         ("C4 256 x 200 poses, 500 obstacles (50 dynamic), teb_autosize off", lambda: _c4()),
         ("C5 1 x 300 poses, polygon footprint vs 300 polygon obstacles, car-like", lambda: scenes.scene_c5(stride=336))]
def _c4():
    cfg, obst, via, batch = scenes.scene_c4(stride=208)
    cfg.trajectory.teb_autosize = False
    return cfg, obst, via, batch


for name, mk in CASES:
    cfg, obst, via, batch = mk()
    s = planner.make_solver(cfg, obst, via, batch)
    s.snapshot()
    ms = []
    for rep in range(5):
        s.restore()
        s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, cfg.hcp.selection_obst_cost_scale,
                   cfg.hcp.selection_viapoint_cost_scale, cfg.hcp.selection_alternative_time_cost)
        res = s.results()
        ms.append(s.last_kernel_ms())
    out = s.download(batch.copy())
    units = int(res.lm_iterations.sum())
    print("%-78s kernel %.3f ms  units %5d  -> %9.0f units/s   poses after %d..%d  status ok %d/%d" %
          (name, np.median(ms), units, units / (np.median(ms) * 1e-3), out.n.min(), out.n.max(), int((res.status == 0).sum()), batch.count),
          flush=True)
    s.close()
