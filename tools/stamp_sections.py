"""Sections of the edge loops of the PRODUCT kernel, one per diagnostic build (-DTEB_AMD_STAMP=<id>, csrc/teb_kernel.hpp): the section's
cycles land in the spare slot of the phase log.   usage (GPU box): python tools/stamp_sections.py tools/libv_stamp_<id>.so ...  [c4on]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = {0: "evaluate: static obstacle edges", 1: "evaluate: dynamic obstacle edges", 2: "evaluate: between-pose terms", 3: "linearise: static obstacle edges",
         4: "linearise: dynamic obstacle edges", 5: "linearise: between-pose terms", 10: "linearize(): zero H, b", 11: "linearize(): trig", 12: "linearize(): near masks",
         13: "linearize(): edges", 14: "linearize(): slice reduction", 15: "linearize(): scatter", 16: "linearize(): fixed rows + chi2", 18: "graph: trig",
         19: "graph: association", 20: "graph: time stamps", 21: "graph: via-points"}
if os.environ.get("TEB_AMD_LIB") and len(sys.argv) >= 2 and sys.argv[1] == "--one":
    import numpy as np
    from tools.phase_split import run
    log, ms_off, ms_on, n, res = run(sys.argv[2], reps=3)
    print("%.2f %% of the workgroup cycles (mean of bands), %.0f cycles; kernel %.3f ms" % (100 * (log[:, 7] / log[:, 8]).mean(), log[:, 7].mean(), ms_off))
else:
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    case = ([a for a in sys.argv[1:] if not a.endswith(".so")] or ["c4on"])[0]
    for lib in libs:
        sid = int(os.path.basename(lib).split("_")[-1].split(".")[0])
        out = subprocess.run([sys.executable, __file__, "--one", case], env=dict(os.environ, TEB_AMD_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        print("%-40s %s" % (NAMES.get(sid, str(sid)), (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)
