#!/usr/bin/env python3
"""GPU: deviation of both Jacobian modes from the vectors of the reference's own code (tests/golden/ref_opt_*.npz)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_ref_golden as RG  # noqa: E402
from teb_local_planner_amd import planner, _abi  # noqa: E402

for name in RG.PLANNER_CASES:
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_opt_%s.npz" % name))
    line = [name]
    for jm in (_abi.JACOBIAN_ANALYTIC, _abi.JACOBIAN_G2O_NUMERIC):
        cfg, obst, via, batch = RG.PLANNER_CASES[name]()
        cfg.jacobian_mode = jm
        s = planner.make_solver(cfg, obst, via, batch)
        s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, cfg.hcp.selection_obst_cost_scale,
                   cfg.hcp.selection_viapoint_cost_scale, cfg.hcp.selection_alternative_time_cost)
        res = s.results(); out = s.download(batch.copy()); ms = s.last_kernel_ms(); s.close()
        devs = []
        for b in range(min(batch.count, RG.MAX_TEBS)):
            n = int(g["n"][b])
            if int(out.n[b]) != n:
                devs.append("n%d/%d" % (out.n[b], n)); continue
            x, y, th, dt = out.get_teb(b)
            d = max(np.abs(x - g["state"][b, 0, :n]).max(), np.abs(y - g["state"][b, 1, :n]).max(),
                    np.abs(th - g["state"][b, 2, :n]).max(), np.abs(dt - g["state"][b, 3, :n - 1]).max())
            devs.append("%.1e(c %.1e)" % (d, abs(res.cost[b] - g["cost"][b]) / abs(g["cost"][b])))
        line.append(("analytic" if jm == 0 else "numeric") + " " + " ".join(devs) + " %.2fms" % ms)
    print(" | ".join(line), flush=True)
