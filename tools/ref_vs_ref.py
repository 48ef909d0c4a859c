"""Reference-vs-reference noise floor: the reference's own optimizeTEB (src/optimal_planner.cpp compiled in place)
in two builds - oracle/_ref/libteb_ref.so (strict IEEE: -O2, no contraction, libm sin / cos) and libteb_ref_alt.so (-O3, FMA
contraction, builtin sin / cos) - on every band of the measured configurations. CPU only.
    python tools/ref_vs_ref.py [c4_headline c3 c2 c5] > profiles/ref_vs_ref_r04.txt"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from teb_local_planner_amd import scenes  # noqa: E402
from oracle import ref_py, ref_alt_py, refcode_compare as RC  # noqa: E402

CASES = {
    "c4_headline": lambda: scenes.scene_c4(B=256, n=200, seed=1004, stride=288),
    "c2": lambda: scenes.scene_c2(stride=208),
    "c3": lambda: scenes.scene_c3(stride=208),
    "c5": lambda: scenes.scene_c5(stride=320),
}

if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in CASES] or list(CASES)
    th = os.cpu_count() or 1
    for name in names:
        cfg, obst, via, batch = CASES[name]()
        t0 = time.time()
        a = ref_py.optimize_batch(cfg, obst, via, batch, threads=th, trace=True)
        b = ref_alt_py.optimize_batch(cfg, obst, via, batch, threads=th, trace=True)
        rep = RC.ref_vs_ref(a[0], a[1], a[2], a[4], b[0], b[1], b[2], b[4])
        pb = rep.pop("per_band")
        best_a, best_b = RC.select_best_of_costs(a[2]), RC.select_best_of_costs(b[2])
        print("== %s: %d bands, two builds of the reference's code (%.1f s host)" % (name, batch.count, time.time() - t0))
        print("   success equal %d, pose counts equal %d, LM accept/reject sequences equal %d" % (rep["success_equal"], rep["pose_counts_equal"], rep["lm_sequences_equal"]))
        print("   state error p50 / p99 / max: %s" % rep["state_err"])
        print("   chi2 rel    p50 / p99 / max: %s ; cost rel max %s" % (rep["chi2_rel"], rep["cost_rel_max"]))
        print("   beyond T3 (1e-3 m/rad/s, chi2 1e-3 rel): %d" % rep["bands_outside_T3"])
        for o in rep["outside"]:
            print("      ", json.dumps(o))
        for o in rep["pose_count_mismatch"]:
            print("      pose count differs:", json.dumps(o))
        print("   selectBestTeb index: strict build %d, alt build %d" % (best_a, best_b))
        big = sorted(((d, i) for i, d in enumerate(pb) if d is not None), reverse=True)[:12]
        print("   largest per-band distances:", ", ".join("band %d: %.2e" % (i, d) for d, i in big))
