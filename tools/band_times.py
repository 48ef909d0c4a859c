"""Which bands does a launch wait for? Per-band workgroup cycles of a -DTEB_PROFILE build (tools/build_prof.sh) beside pose count, LM
trials and iterations: TEB_AMD_LIB=$PWD/tools/libteb_amd_prof.so python tools/band_times.py [c4on|c3]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from teb_local_planner_amd import planner, scenes, _abi
Lb = planner.lib()
which = (sys.argv[1:] or ["c4on"])[0]
cfg, obst, via, batch = scenes.scene_c4(stride=288) if which == "c4on" else scenes.scene_c3(stride=208)
s = planner.make_solver(cfg, obst, via, batch)
for rep in range(2):
    s.upload(batch)
    s.optimize(5, 4, True, 100.0, 1.0, False)
    res = s.results()
per = np.zeros(batch.count)
Lb.teb_amd_debug_profile_bands.argtypes = [C.c_void_p, _abi.p_f64]
assert Lb.teb_amd_debug_profile_bands(s._h, _abi._ptr(per, C.c_double)) == 0
n = s.pose_counts()
order = np.argsort(-per)
print("kernel %.3f ms; slowest bands: index, time / slowest, poses after, LM trials, LM iterations" % s.last_kernel_ms())
for b in order[:12]:
    print("  %3d  %.3f  n=%3d  trials=%2d  iters=%2d" % (b, per[b] / per.max(), n[b], res.lm_trials[b], res.lm_iterations[b]))
print("fastest:")
for b in order[-4:]:
    print("  %3d  %.3f  n=%3d  trials=%2d  iters=%2d" % (b, per[b] / per.max(), n[b], res.lm_trials[b], res.lm_iterations[b]))
t = np.asarray(res.lm_trials, dtype=float)
print("correlation of time with trials %.2f, with poses %.2f; trials min/mean/max %d / %.1f / %d" % (np.corrcoef(per, t)[0, 1], np.corrcoef(per, n)[0, 1], t.min(), t.mean(), t.max()))
A = np.stack([np.ones_like(t), t, np.asarray(n, dtype=float), (np.asarray(n) > 256).astype(float)], axis=1)
coef, *_ = np.linalg.lstsq(A, per / per.max(), rcond=None)
print("least squares: time/slowest = %.3f + %.4f trials + %.5f poses + %.3f [poses > 256]" % tuple(coef))
s.close()
