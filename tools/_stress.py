"""stress: the small mixed circular scene, one CU per band repeated and 40 distance helpers repeated, each against the first one-CU result"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import scenes, planner, _abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fp = sys.argv[2] if len(sys.argv) > 2 else "circular"
def run(**opt):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint=fp)
    dbg = opt.pop("dbg", 0)
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(**opt))
    if dbg:
        import ctypes as C
        planner.lib().teb_amd_debug_mcu_flags.argtypes = [C.c_void_p, C.c_int32]
        assert planner.lib().teb_amd_debug_mcu_flags(s._h, dbg) == 0
    s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale, cfg.hcp.selection_alternative_time_cost)
    out = s.download(batch.copy()); r = s.results(); s.close()
    return np.concatenate([out.x.ravel(), out.y.ravel(), out.theta.ravel(), out.dt.ravel(), r.chi2.ravel(), r.lm_trials.astype(float).ravel()]), out.n.copy()
ref, n = run(multi_cu=-1, speculative_trials=-1, generic_distance_path=True)
print("poses", n, flush=True)
for name, opt in [c for c in (("one CU", dict(multi_cu=-1)), ("40 helpers", dict(multi_cu=40)), ("7 helpers", dict(multi_cu=7)), ("20 helpers", dict(multi_cu=20)), ("80 helpers", dict(multi_cu=80)), ("80assocself helpers (master associates, helpers DIST only)", dict(multi_cu=80, dbg=1)), ("80distself helpers (helpers ASSOC only, master computes distances)", dict(multi_cu=80, dbg=2))) if not os.environ.get("ONLY") or c[0].split()[0] in os.environ["ONLY"].split(",")]:
    bad = 0
    for k in range(N):
        v, _ = run(speculative_trials=-1, generic_distance_path=True, **opt)
        if not np.array_equal(v, ref):
            bad += 1
            if bad <= 3: print("  %s run %d differs: max |diff| %.3e, bands %s" % (name, k, np.abs(v - ref).max(), np.nonzero(np.abs(v - ref).reshape(6, 3, -1).max(axis=(0, 2)) > 0)[0] if v.size % 18 == 0 else "?"), flush=True)
    print("%s %s: %d of %d runs differ from the first one-CU result" % (os.path.basename(os.environ.get("TEB_AMD_LIB", "product")), name, bad, N), flush=True)
