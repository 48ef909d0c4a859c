"""Stress of the multi-CU distance helpers (profiles/mcu_race_r06.txt, DESIGN.md section 8 "Known defect"): the 3-band small mixed scene
(circular footprint) launched N times with HELPERS distance helpers per band asked for (default 80; the host grants at most stride / 6),
each against ONE launch on one CU per band; prints the launches of which any bit differs (band, first differing LM iteration, chi2).
usage (GPU box): [HELPERS=40] [TEB_AMD_LIB=..] python tools/mcu_stress.py 2000"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import scenes, planner, _abi
N = int(sys.argv[1])
def run(**opt):
    cfg, obst, via, batch = scenes.scene_small_mixed(footprint="circular")
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(**opt))
    s.set_iteration_log(True)
    s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale, cfg.hcp.selection_alternative_time_cost)
    out = s.download(batch.copy()); r = s.results(); info = s.last_launch_info(); fl = s.debug_overflow_flags(); logs = [s.iteration_log(b) for b in range(3)]; s.close()
    return out, r, info, fl, logs
o0, r0, i0, f0, l0 = run(multi_cu=-1, speculative_trials=-1, generic_distance_path=True)
print("ref", o0.n, r0.lm_trials, r0.lm_iterations, r0.chi2, i0, flush=True)
bad = 0
for k in range(N):
    o, r, info, fl, lg = run(multi_cu=int(os.environ.get("HELPERS", "80")), speculative_trials=-1, generic_distance_path=True)
    same = all(np.array_equal(getattr(o, a), getattr(o0, a)) for a in ("x", "y", "theta", "dt")) and np.array_equal(r.chi2, r0.chi2)
    if not same:
        bad += 1
        if bad <= 8:
            db = [b for b in range(len(o.n)) if not (np.array_equal(o.x[b], o0.x[b]) and np.array_equal(o.theta[b], o0.theta[b]) and np.array_equal(o.dt[b], o0.dt[b]))]
            for b in db:
                d = [j for j in range(min(len(lg[b]), len(l0[b]))) if not np.array_equal(lg[b][j], l0[b][j])]
                print("   band %d: first differing LM iteration %s of %d; rows there %s vs %s" % (b, d[:1], len(l0[b]), lg[b][d[0]] if d else None, l0[b][d[0]] if d else None), flush=True)
            print("run %d: info %s flags %s bands %s n %s trials %s vs %s iters %s chi2 %s vs %s status %s" % (k, info, fl, db, o.n, r.lm_trials, r0.lm_trials, r.lm_iterations, r.chi2, r0.chi2, r.status), flush=True)
print("%d of %d differ" % (bad, N))
