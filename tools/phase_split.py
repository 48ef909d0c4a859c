"""Phase split of the PRODUCT kernel (teb_amd_set_phase_log, include/teb_amd_debug.h): shader cycles per phase of every band's workgroup,
measured by the kernel the bench times - not by the -DTEB_PROFILE build (tools/prof_phases.py: finer counters, 40 % slower).
usage (on the GPU box): python tools/phase_split.py [c4on c4 c3 c2 c5]        TEB_AMD_LIB selects a build"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import planner, scenes


def scene(which):
    if which == "c4":
        a = scenes.scene_c4(stride=208); a[0].trajectory.teb_autosize = False; return a
    return {"c4on": lambda: scenes.scene_c4(stride=288), "c3": lambda: scenes.scene_c3(stride=208), "c2": lambda: scenes.scene_c2(stride=208),
            "c5": lambda: scenes.scene_c5(stride=320)}[which]()


def run(which, reps=9):
    cfg, obst, via, batch = scene(which)
    s = planner.make_solver(cfg, obst, via, batch)
    s.snapshot()
    ms = {False: [], True: []}
    log = None
    for on in (False, True, False, True):
        s.set_phase_log(on)
        for r in range(reps + 1):
            s.restore()
            s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize()
            if r:
                ms[on].append(s.last_kernel_ms())
        if on:
            log = s.phase_log()
    n = s.pose_counts(); res = s.results()
    s.close()
    return log, float(np.median(ms[False])), float(np.median(ms[True])), n, res


def report(which):
    log, ms_off, ms_on, n, res = run(which)
    tot = log[:, :7].sum(axis=1)
    slow = int(np.argmax(log[:, 8]))
    out = ["== %s: kernel %.3f ms with the phase log off, %.3f ms on (x %.3f); %d bands, %d .. %d poses, LM iterations %d, trials %d" % (
        which, ms_off, ms_on, ms_on / ms_off, len(log), n.min(), n.max(), int(res.lm_iterations.sum()), int(res.lm_trials.sum()))]
    out.append("   %-20s %14s %14s %16s" % ("phase", "mean of bands", "slowest band", "cycles (slowest)"))
    share = log[:, :7] / tot[:, None]
    for k, name in enumerate(planner.TebBatchSolver.PHASES):
        out.append("   %-20s %13.1f %% %13.1f %% %16.0f" % (name, 100 * share[:, k].mean(), 100 * share[slow, k], log[slow, k]))
    out.append("   phases cover %.1f %% of the workgroup's cycles (mean); slowest band %d: %.0f cycles = %.3f ms at %.0f MHz; CU utilisation (mean / max) %.2f" % (
        100 * (tot / log[:, 8]).mean(), slow, log[slow, 8], ms_on, log[slow, 8] / (ms_on * 1e3), log[:, 8].mean() / log[:, 8].max()))
    long_ = n > 256
    if long_.any() and (~long_).any():   # what crossing 256 poses costs, phase by phase (cycles per LM trial of the band)
        tr = res.lm_trials.astype(float)
        out.append("   cycles per LM trial, bands <= 256 poses (%d) | > 256 poses (%d):" % ((~long_).sum(), long_.sum()))
        for k, name in enumerate(planner.TebBatchSolver.PHASES):
            a, b = (log[~long_, k] / tr[~long_]).mean(), (log[long_, k] / tr[long_]).mean()
            out.append("   %-20s %10.0f | %10.0f   (%+.0f)" % (name, a, b, b - a))
        a, b = (log[~long_, 8] / tr[~long_]).mean(), (log[long_, 8] / tr[long_]).mean()
        out.append("   %-20s %10.0f | %10.0f   (%+.0f = %+.1f %%)" % ("whole workgroup", a, b, b - a, 100 * (b - a) / a))
    return "\n".join(out)


if __name__ == "__main__":
    for w in sys.argv[1:] or ["c4on", "c4", "c3", "c2", "c5"]:
        print(report(w), flush=True)
