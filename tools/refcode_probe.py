"""Device (both Jacobian modes) vs the reference's own code (oracle/_ref) and vs the oracle in the same mode, on the measured
configurations at full size; prints the distributions the tests in tests/test_gpu_reference_code.py assert floors on."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teb_local_planner_amd import planner, scenes, _abi
from oracle import oracle_py, ref_py, ref_alt_py, refcode_compare as RC

CASES = {
    "c4_headline": lambda: scenes.scene_c4(B=256, n=200, seed=1004, stride=288),
    "c2": lambda: scenes.scene_c2(stride=208),
    "c3": lambda: scenes.scene_c3(stride=208),
    "c5": lambda: scenes.scene_c5(stride=320),
}
T = os.cpu_count() or 1
which = sys.argv[1:] or list(CASES)
for name in which:
    cfg, obst, via, batch = CASES[name]()
    t0 = time.time()
    rout, rok, rcost, rit, rtr = ref_py.optimize_batch(cfg, obst, via, batch, threads=T, trace=True)
    t_ref = time.time() - t0
    # the reference against a second build of itself (oracle/_ref/libteb_ref_alt.so): the per-band noise floor
    aout, aok, acost, ait, atr = ref_alt_py.optimize_batch(cfg, obst, via, batch, threads=T, trace=True)
    rr = RC.ref_vs_ref(aout, aok, acost, atr, rout, rok, rcost, rtr)
    rr_band = rr.pop("per_band")
    print("== %s reference vs reference (alt build): %s" % (name, json.dumps({k: v for k, v in rr.items() if k not in ("outside", "pose_count_mismatch")})))
    for o in rr["outside"]: print("      ref-vs-ref outside T3:", o)
    best_ref = RC.select_best_of_costs(rcost)
    for mode, jm in (("analytic", _abi.JACOBIAN_ANALYTIC), ("g2o_numeric", _abi.JACOBIAN_G2O_NUMERIC)):
        cfg.jacobian_mode = jm
        out, res, tr, ms = RC.run_device_traced(planner, cfg, obst, via, batch)
        rep = RC.compare_with_reference_code(out, res, tr, rout, rok, rcost, rtr)
        oout, ores, otr = oracle_py.optimize_batch(cfg, obst, via, batch, threads=T, trace=True)
        same_n = [b for b in range(batch.count) if out.n[b] == oout.n[b]]
        d = np.array([RC.state_error(out.get_teb(b), oout.get_teb(b)) for b in same_n])
        seq = sum(int(RC.first_divergence(tr[b], otr[b]) is None or RC.first_divergence(tr[b], otr[b])[0] != "accept/reject") for b in range(batch.count))
        print("== %s %s: kernel %.3f ms, reference code %.2f s on %d threads" % (name, mode, ms, t_ref, T))
        print("   vs reference code:", json.dumps({k: v for k, v in rep.items() if k not in ("outside", "pose_count_mismatch")}))
        for o in rep["outside"][:8]: print("      outside T3:", o)
        for o in rep["pose_count_mismatch"][:8]: print("      pose count:", o)
        best_dev = res.best_index
        print("   T4 selectBestTeb: device %d, reference code %d, alt build %d" % (best_dev, best_ref, RC.select_best_of_costs(acost)))
        dd = [(RC.state_error(out.get_teb(b), rout.get_teb(b)), rr_band[b], b) for b in range(batch.count) if out.n[b] == rout.n[b] and rr_band[b] is not None]
        ratio = np.array([a / max(r, 1e-9) for a, r, b in dd])
        print("   device-to-reference / reference-to-reference per band (floor 1e-9): p50 %.2f p90 %.2f p99 %.2f max %.2f" % (
            np.median(ratio), np.percentile(ratio, 90), np.percentile(ratio, 99), ratio.max()))
        for a, r, b in sorted(dd, reverse=True)[:16]: print("      band %3d: device %.2e  ref-vs-ref %.2e  ratio %.1f" % (b, a, r, a / max(r, 1e-9)))
        worst = sorted(((a / max(r, 1e-9), a, r, b) for a, r, b in dd if a > 2e-5), reverse=True)[:10]
        for q, a, r, b in worst: print("      ratio-worst (device > 2e-5) band %3d: device %.2e  ref-vs-ref %.2e  ratio %.1f" % (b, a, r, q))
        print("   vs oracle (same mode): pose counts equal %d / %d, LM sequences equal %d, state err p50 %.2e p99 %.2e max %.2e; > 2e-5: %d, > 1e-3: %d" % (
            len(same_n), batch.count, seq, np.median(d), np.percentile(d, 99), d.max(), int((d > 2e-5).sum()), int((d > 1e-3).sum())))
    cfg.jacobian_mode = _abi.JACOBIAN_ANALYTIC
