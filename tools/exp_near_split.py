import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.phase_split import run
for w in sys.argv[1:] or ["c4on"]:
    log, ms_off, ms_on, n, res = run(w, reps=5)
    print("== %s kernel %.3f ms; near masks (wave 0) %.2f %% of the workgroup cycles (mean), %.0f cycles; trials %d" % (w, ms_on, 100 * (log[:, 7] / log[:, 8]).mean(), log[:, 7].mean(), int(res.lm_trials.sum())))
