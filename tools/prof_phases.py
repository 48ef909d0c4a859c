"""Phase breakdown of teb_optimize_kernel (workgroup 0) with a -DTEB_PROFILE build of the library:
   tools/build_prof.sh && TEB_AMD_LIB=$PWD/tools/libteb_amd_prof.so python tools/prof_phases.py [c4on c4 c3 c2 c5] [--one-cu]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import planner, scenes, _abi
Lb = planner.lib()
Lb.teb_amd_debug_profile.argtypes = [C.c_void_p, _abi.p_f64]
ONE_CU = "--one-cu" in sys.argv
OPT = _abi.Options(multi_cu=-1, speculative_trials=-1) if ONE_CU else None
names = ["autoresize", "assoc+via+tdyn", "linearize", "H backup", "solve", "update+evaluate", "accept/restore"]
for which in [a for a in sys.argv[1:] if not a.startswith("--")] or ["c4", "c2", "c5"]:
    if which == "c4":
        cfg, obst, via, batch = scenes.scene_c4(); cfg.trajectory.teb_autosize = False
    elif which == "c4on":
        cfg, obst, via, batch = scenes.scene_c4(stride=288)
    elif which == "c3":
        cfg, obst, via, batch = scenes.scene_c3(stride=208)
    elif which == "c2":
        cfg, obst, via, batch = scenes.scene_c2(stride=208)
    else:
        cfg, obst, via, batch = scenes.scene_c5(stride=320)
    s = planner.make_solver(cfg, obst, via, batch, options=OPT)
    for rep in range(2):
        s.upload(batch)
        s.optimize(5, 4, True, 100.0, 1.0, False)
        res = s.results()
    cyc = np.zeros(8)
    rc = Lb.teb_amd_debug_profile(s._h, _abi._ptr(cyc, C.c_double))
    ms = s.last_kernel_ms()
    print("== %s kernel %.3f ms, TEB0: iters %d trials %d, helpers (distance, solver, repeated) %s" % (which, ms, res.lm_iterations[0], res.lm_trials[0], s.last_launch_info()))
    tot = cyc[:7].sum()
    for k, nm in enumerate(names):
        print("   %-18s %12.0f cycles  %5.1f %%   (%.1f us @100MHz-clock64?)" % (nm, cyc[k], 100 * cyc[k] / tot, cyc[k] / 100.0))
    print("   autoResize detail: %d sequential sweeps, %.0f cycles inside them" % (int(cyc[7] // 1e9), cyc[7] % 1e9))
    tot = cyc[:7].sum()
    print("   total cycles %.0f -> implied counter rate %.1f MHz" % (tot, tot / (ms * 1e3)))
    per = np.zeros(batch.count)
    Lb.teb_amd_debug_profile_bands.argtypes = [C.c_void_p, _abi.p_f64]
    if Lb.teb_amd_debug_profile_bands(s._h, _abi._ptr(per, C.c_double)) == 0 and batch.count > 1:
        n_after = s.pose_counts()
        q = np.percentile(per, [0, 25, 50, 75, 100]) / per.max()
        print("   per-band workgroup time / slowest: min %.2f  p25 %.2f  p50 %.2f  p75 %.2f  max 1.00; mean %.2f (= CU utilisation of the launch)" % (q[0], q[1], q[2], q[3], per.mean() / per.max()))
        long_ = n_after > 256
        if long_.any() and (~long_).any():
            print("   bands > 256 poses: %d, mean time %.2f of the slowest; bands <= 256 poses: mean %.2f" % (int(long_.sum()), per[long_].mean() / per.max(), per[~long_].mean() / per.max()))
        print("   correlation of time with pose count: %.2f" % np.corrcoef(per, n_after)[0, 1])
    s.close()
