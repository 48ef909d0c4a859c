"""each specialised instantiation that is a candidate for retirement, timed against the kernel the host would fall back
to (teb_amd_options_t::generic_config_path forces the generic kind of the same batch size) - same box, alternating, median kernel ms.
Retire at <= 2 % loss.   usage (GPU box): python tools/retire_candidates.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import planner, scenes, _abi


def c3_via():
    c, o, v, b = scenes.scene_c3(stride=208)
    c.optim.weight_viapoint = 1.0; v = [(4.0, 0.3), (8.0, -0.2), (12.0, 0.25)]; b.via_points_enabled[:] = 1
    return c, o, v, b

def c3_short():
    c, o, v, b = scenes.scene_c3(stride=208); c.optim.weight_shortest_path = 1.0
    return c, o, v, b

def c2_via():
    c, o, v, b = scenes.scene_c2(stride=208)
    c.optim.weight_viapoint = 1.0; v = [(5.0, 0.3), (10.0, -0.2), (15.0, 0.25)]; b.via_points_enabled[:] = 1
    return c, o, v, b

def c2_short():
    c, o, v, b = scenes.scene_c2(stride=208); c.optim.weight_shortest_path = 1.0
    return c, o, v, b

def c3_band_via():
    c, o, v, b = scenes.scene_c3(stride=288)
    c.optim.weight_viapoint = 1.0; v = [(4.0, 0.3), (8.0, -0.2), (12.0, 0.25)]; b.via_points_enabled[:] = 1
    return c, o, v, b

def c4_numeric():
    c, o, v, b = scenes.scene_c4(stride=288); c.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    return c, o, v, b

def c4fix_numeric():
    c, o, v, b = scenes.scene_c4(stride=208); c.trajectory.teb_autosize = False; c.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    return c, o, v, b

CASES = [("C3 + via-points, blocks layout: small wide kind 9 vs small generic kind 2", c3_via),
         ("C3 + shortest path, blocks layout: small light kind 11 vs small generic kind 2", c3_short),
         ("C2 + via-points: small wide kind 9 vs small generic kind 2", c2_via),
         ("C2 + shortest path: small light kind 11 vs small generic kind 2", c2_short),
         ("C3 + via-points, band layout (capacity 288): small wide kind 9 vs small generic kind 2", c3_band_via),
         ("C4 headline, numeric Jacobians: defaults kind 4 vs generic kind 0", c4_numeric),
         ("C4 fixed 200 (blocks), numeric Jacobians: defaults kind 4 vs generic kind 0", c4fix_numeric),
         ("C5 (polygons, 63 helpers): generic-shape defaults small kind 7 vs generic small kind 3", lambda: scenes.scene_c5(stride=320))]
for label, mk in CASES:
    res = {}
    insts = {}
    for rnd in range(2):
        for forced in (False, True):
            cfg, obst, via, batch = mk()
            s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(generic_config_path=forced))
            s.snapshot()
            ms = []
            for r in range(7):
                s.restore(); s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize()
                ms.append(s.last_kernel_ms())
            res.setdefault(forced, []).append(float(np.median(ms[2:])))
            insts[forced] = s.last_instantiation()
            s.close()
    a, b = np.mean(res[False]), np.mean(res[True])
    print("%-95s %s %.3f ms | fallback %s %.3f ms | retiring it would cost %+.1f %%" % (label, insts[False], a, insts[True], b, 100 * (b / a - 1)), flush=True)
