"""Developer script: GPU path vs oracle on small scenes, prints deltas at each parity level."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import scenes, _abi, planner
from oracle import oracle_py as O


def cmp_scene(name, cfg, obst, via, batch, full=True):
    print("=== %s: B=%d n=%s M=%d" % (name, batch.count, batch.n[:4], len(obst)))
    s = planner.make_solver(cfg, obst, via, batch, max_poses=batch.stride)
    print("capacity", s.capacity())
    # distances
    rng = np.random.default_rng(0)
    nq = 200
    oi = rng.integers(0, len(obst), nq); x = rng.uniform(0, 6, nq); y = rng.uniform(-2, 2, nq); th = rng.uniform(-3, 3, nq)
    t = rng.uniform(0, 5, nq)
    for tt in (None, t):
        dg, gg = s.debug_distance(oi, x, y, th, tt)
        do = np.zeros(nq); go = np.zeros((nq, 3))
        for q in range(nq):
            do[q], go[q] = O.distance(cfg, obst, int(oi[q]), x[q], y[q], th[q], None if tt is None else tt[q])
        print(" dist max|d|", np.abs(dg - do).max(), "grad", np.abs(gg - go).max())
    for b in range(min(batch.count, 3)):
        n = int(batch.n[b])
        for wm in (1.0, 4.0):
            G = s.debug_linearize(b, n, wm)
            R = O.linearize(cfg, obst, via, batch, b, wm)
            sH = np.abs(R["H"]).max(); sb = np.abs(R["b"]).max()
            print(" lin b=%d wm=%g: dH %.3e (rel %.2e) db %.3e (rel %.2e) chi2 %s vs %s" % (
                b, wm, np.abs(G["H"] - R["H"]).max(), np.abs(G["H"] - R["H"]).max() / sH,
                np.abs(G["b"] - R["b"]).max(), np.abs(G["b"] - R["b"]).max() / sb, G["chi2"], R["chi2"]))
            ap, ao = O.associate(cfg, obst, batch, b)
            same = len(ap) == len(G["assoc_pose"]) and (ap == G["assoc_pose"]).all() and (ao == G["assoc_obst"]).all()
            print("   assoc pairs %d vs %d same=%s" % (len(G["assoc_pose"]), len(ap), same))
    if full:
        t0 = time.time()
        s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True,
                   cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale,
                   cfg.hcp.selection_alternative_time_cost)
        res = s.results()
        print(" gpu optimize wall %.3f s kernel %.3f ms" % (time.time() - t0, s.last_kernel_ms()))
        out = batch.copy()
        s.download(out)
        t0 = time.time()
        ob, ores = O.optimize_batch(cfg, obst, via, batch)
        print(" oracle wall %.3f s" % (time.time() - t0))
        print(" n", out.n[:6], ob.n[:6])
        print(" status", res.status[:6], ores.status[:6], "iters", res.lm_iterations[:6], ores.lm_iterations[:6],
              "trials", res.lm_trials[:6], ores.lm_trials[:6])
        print(" chi2", res.chi2[:4], ores.chi2[:4])
        print(" cost", res.cost[:4], ores.cost[:4])
        for b in range(min(batch.count, 4)):
            if out.n[b] == ob.n[b]:
                d = [np.abs(u - v).max() for u, v in zip(out.get_teb(b), ob.get_teb(b))]
                print("  teb %d max|dx,dy,dth,ddt| = %s" % (b, d))
        print(" overflow flags", s.debug_overflow_flags()[:8])
        print(" select", s.select_best(), O.select_best(cfg, ores.cost))
    s.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["mixed", "c1", "c2"]
    if "mixed" in which:
        for fp in ("point", "circular", "two_circles", "line", "polygon"):
            cmp_scene("mixed/" + fp, *scenes.scene_small_mixed(footprint=fp))
    if "c1" in which:
        cmp_scene("c1", *scenes.scene_c1())
    if "c2" in which:
        cmp_scene("c2", *scenes.scene_c2(stride=208))
    if "c5" in which:
        cmp_scene("c5", *scenes.scene_c5(n=120, M=80, stride=200))
