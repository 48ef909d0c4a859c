"""Do two builds give the same bands? usage (GPU box): python tools/ab_equal.py libA.so libB.so [case ..]
Runs the measured configurations (tools/kernel_times.py's cases) once per library, each in a process of its own (TEB_AMD_LIB), and reports
per case: identical bits or the largest state / cost difference and the bands whose pose count or LM trial count differ."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dump(case, path):
    from teb_local_planner_amd import planner, scenes
    def _c4fix():
        a = scenes.scene_c4(stride=208); a[0].trajectory.teb_autosize = False; return a
    cases = {"c4on": lambda: scenes.scene_c4(stride=288), "c4fix": _c4fix, "c2": lambda: scenes.scene_c2(stride=208),
             "c3": lambda: scenes.scene_c3(stride=208), "c5": lambda: scenes.scene_c5(stride=320)}
    cfg, obst, via, batch = cases[case]()
    s = planner.make_solver(cfg, obst, via, batch)
    s.optimize(5, 4, True, 100.0, 1.0, False)
    s.synchronize()
    out = s.download(batch.copy())
    r = s.results()
    np.savez(path, n=out.n, x=out.x, y=out.y, theta=out.theta, dt=out.dt, cost=r.cost, chi2=r.chi2, trials=r.lm_trials, status=r.status)
    s.close()


if __name__ == "__main__":
    if sys.argv[1] == "--dump":
        dump(sys.argv[2], sys.argv[3])
        sys.exit(0)
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    cases = [a for a in sys.argv[1:] if not a.endswith(".so")] or ["c4on", "c4fix", "c2", "c3", "c5"]
    tmp = tempfile.mkdtemp()
    for case in cases:
        res = []
        for k, lib in enumerate(libs):
            p = os.path.join(tmp, "%s_%d.npz" % (case, k))
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--dump", case, p], env=dict(os.environ, TEB_AMD_LIB=os.path.abspath(lib)))
            res.append(np.load(p))
        a = res[0]
        for lib, b in zip(libs[1:], res[1:]):
            same = all(np.array_equal(a[k], b[k]) for k in a.files)
            if same:
                print("%s: %s == %s bit for bit" % (case, os.path.basename(lib), os.path.basename(libs[0])), flush=True)
                continue
            dn = np.nonzero(a["n"] != b["n"])[0]
            dt_ = np.nonzero(a["trials"] != b["trials"])[0]
            eq = a["n"] == b["n"]
            err = 0.0
            for k in ("x", "y", "theta", "dt"):
                err = max(err, float(np.abs(a[k][eq] - b[k][eq]).max())) if eq.any() else err
            crel = float(np.max(np.abs(a["cost"] - b["cost"]) / np.maximum(1e-300, np.abs(a["cost"]))))
            print("%s: %s differs from %s: max state diff %.3e (bands of equal pose count), cost rel %.3e, pose counts differ on %d bands, "
                  "trial counts on %d, status equal %s" % (case, os.path.basename(lib), os.path.basename(libs[0]), err, crel, len(dn), len(dt_),
                                                            bool(np.array_equal(a["status"], b["status"]))), flush=True)
