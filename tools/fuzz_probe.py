import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_gpu_parity as T
from teb_local_planner_amd import planner
from oracle import oracle_py as orc
for seed in [int(a) for a in sys.argv[1:]]:
    cfg, obst, via, batch = T._random_case(seed)
    t, r, o, ob = cfg.trajectory, cfg.robot, cfg.optim, cfg.obstacles
    print("seed", seed, "fp", cfg.robot_model.kind if hasattr(cfg.robot_model, "kind") else cfg.to_c().footprint_type, "holo", r.max_vel_y, "rmin", r.min_turning_radius, "arc", t.exact_arc_length,
          "legacy", ob.legacy_obstacle_association, ob.obstacle_poses_affected, "exp", o.obstacle_cost_exponent, "sp", o.weight_shortest_path,
          "vor", o.weight_velocity_obstacle_ratio, "infl", ob.inflation_dist, "dyn", ob.include_dynamic_obstacles, "via", len(via), t.via_points_ordered)
    s = planner.make_solver(cfg, obst, via, batch)
    for b in range(batch.count):
        G = s.debug_linearize(b, int(batch.n[b]), 2.0)
        R = orc.linearize(cfg, obst, via, batch, b, 2.0)
        print("   teb", b, "chi2 rel", np.abs(G["chi2"] - R["chi2"]).max() / max(1e-300, np.abs(R["chi2"]).max()),
              "H rel", np.abs(G["H"] - R["H"]).max() / np.abs(R["H"]).max(), "b rel", np.abs(G["b"] - R["b"]).max() / np.abs(R["b"]).max())
    s.close()
    for outer, inner in ((1, 1), (1, 3), (1, 5), (2, 1), (2, 2), (2, 3), (2, 5), (4, 5)):
        s = planner.make_solver(cfg, obst, via, batch)
        s.optimize(inner, outer, True, cfg.hcp.selection_obst_cost_scale, cfg.hcp.selection_viapoint_cost_scale, cfg.hcp.selection_alternative_time_cost)
        res = s.results(); out = s.download(batch.copy()); s.close()
        ref, rres = orc.optimize_batch(cfg, obst, via, batch, inner=inner, outer=outer)
        d = [max(np.abs(u - v).max() for u, v in zip(out.get_teb(b), ref.get_teb(b))) if out.n[b] == ref.n[b] else -1 for b in range(batch.count)]
        print("   outer", outer, "inner", inner, "state diff", d, "chi2 rel", np.abs(res.chi2 - rres.chi2) / np.abs(rres.chi2), "trials", res.lm_trials, rres.lm_trials)
    # where does H differ?
    s = planner.make_solver(cfg, obst, via, batch)
    for b in range(batch.count):
        G = s.debug_linearize(b, int(batch.n[b]), 2.0); R = orc.linearize(cfg, obst, via, batch, b, 2.0)
        D = np.abs(G["H"] - R["H"]); m = np.abs(R["H"]).max()
        idx = np.argwhere(D > 1e-9 * m)
        if len(idx):
            print("   teb", b, "n", batch.n[b], "mismatching H entries (row, col, gpu, oracle):")
            for (r_, c_) in idx[:12]:
                print("      pose %d var %d | pose %d var %d : %.10g vs %.10g" % (r_ // 4, r_ % 4, c_ // 4, c_ % 4, G["H"][r_, c_], R["H"][r_, c_]))
            Dm = G["H"] - R["H"]
            rows = sorted(set(idx[:, 0].tolist()) | set(idx[:, 1].tolist()))
            sub = Dm[np.ix_(rows, rows)]
            wv, V = np.linalg.eigh(sub)
            print("      variables involved:", [(r_ // 4, "xytd"[r_ % 4]) for r_ in rows])
            print("      eigenvalues of the difference:", np.round(wv, 6))
            k = int(np.argmax(np.abs(wv)))
            print("      dominant direction:", np.round(V[:, k] * np.sqrt(abs(wv[k])), 5), "sign", np.sign(wv[k]))
            i = int(idx[0][0] // 4)
            x, y, th, dt = batch.get_teb(b)
            print("      state around pose", i, x[i-1:i+3], y[i-1:i+3], th[i-1:i+3], dt[i-1:i+2])
    s.close()
