import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from teb_local_planner_amd import scenes, planner, _abi
def run(cfg, obst, via, batch, flags=0, **opt):
    s = planner.make_solver(cfg, obst, via, batch, options=_abi.Options(**opt))
    L = planner.lib(); L.teb_amd_debug_mcu_flags.argtypes = [C.c_void_p, C.c_int32]; L.teb_amd_debug_mcu_watchdog.argtypes = [C.c_void_p, C.c_int32]
    L.teb_amd_debug_mcu_flags(s._h, flags); L.teb_amd_debug_mcu_watchdog(s._h, 3000)
    s.set_iteration_log(True)
    s.optimize(5, 4, True, 100.0, 1.0, False); s.synchronize()
    r = s.results(); out = s.download(batch.copy()); tr = [s.iteration_log(b) for b in range(batch.count)]
    info = s.last_launch_info(); s.close()
    return out, r, tr, info
fp = sys.argv[1] if len(sys.argv) > 1 else "point"
sc = scenes.scene_small_mixed(footprint=fp, with_dynamic=(len(sys.argv) < 3 or sys.argv[2] != "nodyn"), with_via=(len(sys.argv) < 4 or sys.argv[3] != "novia"))
base = run(*sc, multi_cu=-1, generic_distance_path=True)
for name, flags in (("assoc+dist by helpers", 0), ("dist only by helpers", 1), ("assoc only by helpers", 2), ("nothing by helpers", 3)):
    o, r, tr, info = run(*sc, flags=flags, multi_cu=2, generic_distance_path=True)
    same = all(np.array_equal(getattr(o, k), getattr(base[0], k)) for k in ("x", "y", "theta", "dt"))
    first = None
    for b in range(sc[3].count):
        for k in range(min(len(tr[b]), len(base[2][b]))):
            if not np.array_equal(tr[b][k], base[2][b][k]):
                first = (b, k, tr[b][k].tolist(), base[2][b][k].tolist()); break
        if first: break
    print(fp, name, "info", info, "identical", same, "first differing (band, LM iteration, mcu row, single row):", first, flush=True)
