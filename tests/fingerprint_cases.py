"""Bitwise fingerprints of optimised bands: sha256 over pose counts, states, cost and chi^2 of a fixed set of scenes. A kernel change
that is meant to be a pure speed-up must leave every one of them unchanged (tests/test_gpu_bit_fingerprint.py against
tests/golden/bit_fingerprints.json; tools/bit_fingerprint.py prints them for a build given by TEB_AMD_LIB)."""
import hashlib

import numpy as np

from teb_local_planner_amd import planner, scenes

CASES = {
    "c4on": lambda: scenes.scene_c4(stride=288),
    "c2": lambda: scenes.scene_c2(stride=208),
    "c3": lambda: scenes.scene_c3(stride=208),
    "c5": lambda: scenes.scene_c5(stride=320),
    "mixed_polygon": lambda: scenes.scene_small_mixed(footprint="polygon"),
    "mixed_two_circles": lambda: scenes.scene_small_mixed(footprint="two_circles"),
}


def fingerprint(name, options=None):
    cfg, obst, via, batch = CASES[name]()
    s = planner.make_solver(cfg, obst, via, batch, options=options)
    s.optimize(5, 4, True, 100.0, 1.0, False)
    s.synchronize()
    out = s.download(batch.copy())
    r = s.results()
    h = hashlib.sha256()
    for a in (out.n, out.x, out.y, out.theta, out.dt, r.cost, r.chi2):
        h.update(np.ascontiguousarray(a).tobytes())
    ms = s.last_kernel_ms()
    s.close()
    return h.hexdigest()[:16], ms
