"""MFMA Schur-update build (-DTEB_AMD_MFMA_SCHUR): operand-map self test of v_mfma_f64_16x16x4_f64, its issue rate on one wave, and the
kernel time of the BASELINE configurations; run once per library (TEB_AMD_LIB=...)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teb_local_planner_amd import planner, scenes, _abi
L = planner.lib()
cfg, obst, via, batch = scenes.scene_c1()
s = planner.make_solver(cfg, obst, via, batch)
rng = np.random.default_rng(1)
A = rng.standard_normal((16, 8)); B = rng.standard_normal((8, 16)); Cm = np.zeros((16, 16)); cyc = C.c_double(0)
L.teb_amd_debug_mfma_selftest.argtypes = [C.c_void_p, _abi.p_f64, _abi.p_f64, _abi.p_f64, C.c_int32, C.POINTER(C.c_double)]
rc = L.teb_amd_debug_mfma_selftest(s._h, _abi._ptr(A, C.c_double), _abi._ptr(B, C.c_double), _abi._ptr(Cm, C.c_double), 4096, C.byref(cyc))
if rc == 0:
    print("mfma self test: max |C - A B| = %.2e (asymmetric operands); %.1f clock ticks per v_mfma_f64_16x16x4_f64 on one wave" % (np.abs(Cm - A @ B).max(), cyc.value))
else:
    print("no MFMA in this build (rc %d)" % rc)
s.close()
def t(name, mk, reps=7):
    cfg, obst, via, batch = mk()
    s = planner.make_solver(cfg, obst, via, batch); s.snapshot(); ms = []
    for _ in range(reps):
        s.restore(); s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize(); ms.append(s.last_kernel_ms())
    r = s.results(); s.close()
    print("%-28s kernel %.3f ms (min %.3f)  trials %d" % (name, np.median(ms), min(ms), int(r.lm_trials.sum())))
def c4f():
    a = scenes.scene_c4(stride=208); a[0].trajectory.teb_autosize = False; return a
t("C2 1x200x100", lambda: scenes.scene_c2(stride=208))
t("C3 64x150x200", lambda: scenes.scene_c3(stride=208))
t("C4 fixed 200 (blocks in LDS)", c4f)
t("C4 headline (capacity 288)", lambda: scenes.scene_c4(stride=288))
