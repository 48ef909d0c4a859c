import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from teb_local_planner_amd import scenes, planner, _abi
fp = sys.argv[1] if len(sys.argv) > 1 else "point"
cfg, obst, via, batch = scenes.scene_small_mixed(footprint=fp)
s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(multi_cu=2, generic_distance_path=True))
s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize()
L = planner.lib()
out = np.zeros(16)
L.teb_amd_debug_mcu_verify.argtypes = [C.c_void_p, _abi.p_f64]
assert L.teb_amd_debug_mcu_verify(s._h, _abi._ptr(out, C.c_double)) == 0
print(fp, "records checked %d | evaluate: dist differs %d, grad-only %d | linearise: dist differs %d, grad-only %d" % tuple(out[:5]))
print("   first mismatch: dist rec %.17g local %.17g | g0 rec %.17g local %.17g | g2 rec %.17g local %.17g | pose %d entry %d" % tuple(out[8:16]))
s.close()
