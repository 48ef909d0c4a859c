"""The measured configurations at FULL size against the reference's own code, in both Jacobian modes.

Reference side: oracle/_ref/libteb_ref.so = the reference's src/optimal_planner.cpp, timed_elastic_band.cpp, obstacles.cpp and all
edge headers compiled in place (oracle/ref_shim/Makefile; travels to the GPU box as a prebuilt file) - B x
TebOptimalPlanner::optimizeTEB (src/optimal_planner.cpp:182-231) on the same bands, with the LM trace of its optimiser.

Configurations: the bench headline (BASELINE C4: 256 bands x 200 poses, 450 static + 50 dynamic obstacles, autoResize on, capacity
288), C2 (1 x 200 x 100), C3 (64 x 150 x 200), C5 (car-like, 300 poses, polygon footprint vs 300 polygons) - each with the device in
the closed-form mode (the benchmarked one) and in the g2o-numeric mode (the reference's own linearisation).

What is asserted, per configuration and mode (SURVEY section 8(c), T3 "vs the faithful oracle"):
  * same success flag on every band;
  * pose counts after the four autoResize / optimise rounds agree on every band (POSE_COUNT_FLOOR = 1);
  * the bands that are NOT within T3 (<= 1e-3 m / rad / s, chi^2 <= 1e-3 relative) of the reference's result are exactly bands on which
    two builds of the reference are not within T3 of each other (round 5; a floor of 97 % of the bands before: seven bands could drift
    where none or three do); every such band is PRINTED with the LM iteration at which the two runs part (the accept / reject
    sequence, or - same decisions throughout - the first chi^2 that differs by 1e-6) - and for each of them the CPU oracle's own two
    Jacobian modes must disagree as well (by more than 2e-6 on that band, i.e. tests/sensitivity.py does not call it well conditioned):
    the distance is the reference's numeric-differentiation noise acting on an ill-conditioned band, not the device;
  * device vs the CPU oracle in the SAME mode: identical pose counts and LM sequences (iteration and trial counts per iteration) on
    every band in the closed-form mode; in the numeric mode the device's libm (sin / cos differ from glibc's in the last bit)
    enters the central differences divided by 2e-9: counts and sequences are still identical on every band, the state is held to the
    per-band yardstick of tests/sensitivity.py on at least NUMERIC_STATE_MIN_INSIDE[configuration] bands (what was observed, per configuration).
The floors are what was observed on MI355X (tools/refcode_probe.py, profiles/refcode_probe_r03.txt) with margin."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import sensitivity  # noqa: E402

from teb_local_planner_amd import scenes, planner, _abi  # noqa: E402
from oracle import ref_py, ref_alt_py, refcode_compare as RC  # noqa: E402

pytestmark = pytest.mark.gpu

THREADS = os.cpu_count() or 1
POSE_COUNT_FLOOR = 1.0    # (0.97 until round 4; every band of every measured configuration ends with the reference's pose count)
# Bands of a configuration that must sit inside their per-band bound in the numeric mode (device vs oracle, same mode). Until round 5 one
# floor of 95 % for all; since round 6 what was OBSERVED per configuration on MI355X (the driver's run and this round's:
# c4_headline 253 of 256 - bands 126 and 134 sit 1 % and 17 % beyond their own bound, band 174 is the ill-conditioned one, 3e-2 -, every band
# of C2 / C3 / C5), minus ONE band of slack on the headline for the two borderline bands. bench.py records the count of its own run
# (secondary.c4_g2o_numeric_jacobians.numeric_state_inside_bound).
NUMERIC_STATE_MIN_INSIDE = {"c4_headline": 252, "c2": 1, "c3": 64, "c5": 1}

CASES = {
    "c4_headline": lambda: scenes.scene_c4(B=256, n=200, seed=1004, stride=288),   # exactly bench.py's rank-0 workload
    "c2": lambda: scenes.scene_c2(stride=208),
    "c3": lambda: scenes.scene_c3(stride=208),
    "c5": lambda: scenes.scene_c5(stride=320),
}
MODES = {"analytic": _abi.JACOBIAN_ANALYTIC, "g2o_numeric": _abi.JACOBIAN_G2O_NUMERIC}

_REF = {}


def _reference(name):
    """The reference's own optimizeTEB on every band of the configuration (cached per session): (out, ok, cost, traces)."""
    if name not in _REF:
        if not os.path.exists(ref_py.SO):
            pytest.skip("oracle/_ref/libteb_ref.so is not built (needs /root/reference once, see oracle/ref_shim/Makefile)")
        cfg, obst, via, batch = CASES[name]()
        out, ok, cost, it, tr = ref_py.optimize_batch(cfg, obst, via, batch, threads=THREADS, trace=True)
        _REF[name] = (out, ok, cost, tr)
    return _REF[name]


_ALT = {}


def _noise_floor(name):
    """The reference against a second build of itself (oracle/_ref/libteb_ref_alt.so) on every band: per-band state distance
    (None where the two builds end with different pose counts) and the alt build's selectBestTeb index."""
    if name not in _ALT:
        if not os.path.exists(ref_alt_py.SO):
            pytest.skip("oracle/_ref/libteb_ref_alt.so is not built (oracle/ref_shim/Makefile)")
        rout, rok, rcost, rtr = _reference(name)
        cfg, obst, via, batch = CASES[name]()
        aout, aok, acost, ait, atr = ref_alt_py.optimize_batch(cfg, obst, via, batch, threads=THREADS, trace=True)
        rr = RC.ref_vs_ref(aout, aok, acost, atr, rout, rok, rcost, rtr)
        _ALT[name] = (rr, RC.select_best_of_costs(acost))
    return _ALT[name]


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", list(CASES))
def test_measured_configuration_matches_reference_code(oracle, name, mode):
    rout, rok, rcost, rtr = _reference(name)
    cfg, obst, via, batch = CASES[name]()
    B = batch.count
    cfg.jacobian_mode = MODES[mode]
    out, res, tr, ms = RC.run_device_traced(planner, cfg, obst, via, batch)
    rep = RC.compare_with_reference_code(out, res, tr, rout, rok, rcost, rtr, ok_status=_abi.TEB_OK)
    print("%s / %s vs the reference's own code: %d bands, success equal %d, pose counts equal %d, LM sequences equal %d, state err %s, "
          "chi2 rel %s, outside T3: %d" % (name, mode, B, rep["success_equal"], rep["pose_counts_equal"], rep["lm_sequences_equal"],
                                           rep["state_err"], rep["chi2_rel"], rep["bands_outside_T3"]))
    for o in rep["outside"]:
        print("   outside T3:", o)
    for o in rep["pose_count_mismatch"]:
        print("   pose count differs:", o)
    assert rep["success_equal"] == B, rep
    assert rep["pose_counts_equal"] >= int(np.floor(POSE_COUNT_FLOOR * B)), rep
    # T4: selectBestTeb on the device's costs = the arg-min of the reference code's own costs (src/homotopy_class_planner.cpp:593-615)
    best_ref = RC.select_best_of_costs(rcost)
    best_dev = res.best_index   # select_best_kernel on the device-resident costs
    assert best_dev == RC.select_best_of_costs(res.cost)
    assert best_dev == best_ref, (name, mode, best_dev, best_ref)
    # Independent yardstick: a SECOND BUILD of the reference's code (-O3, FMA contraction, builtin sin / cos). In the
    # reference's own linearisation scheme (g2o_numeric: same method, only rounding differs) the device may be on EVERY band at most
    # NOISE_FLOOR_K x as far from the reference as that build is (absolute floor K x NOISE_FLOOR_ABS). With closed-form Jacobians the
    # method itself differs from the reference's delta = 1e-9 central differences (whose truncation + cancellation error is not compiler
    # noise: C5 sits 6e-4 from the reference in this mode and 2e-7 in the numeric one), so there the bound is T3 or K x the reference's
    # own noise on that band, whichever is larger. In both modes every band the device has beyond T3 must be one that the two reference
    # builds disagree on by at least a tenth of the device's distance.
    rr, best_alt = _noise_floor(name)
    per_band = rr["per_band"]
    worst = 0.0
    for b in range(B):
        if per_band[b] is None or int(out.n[b]) != int(rout.n[b]):
            continue
        d = RC.state_error(out.get_teb(b), rout.get_teb(b))
        bound = max(RC.NOISE_FLOOR_K * RC.NOISE_FLOOR_ABS if mode == "g2o_numeric" else RC.T3_STATE, RC.NOISE_FLOOR_K * per_band[b])
        worst = max(worst, d / bound)
        assert d <= bound, ("band %d: device %.3e from the reference, the reference's two builds %.3e apart" % (b, d, per_band[b]), name, mode)
    for o in rep["outside"]:
        b = o["band"]
        assert per_band[b] is None or per_band[b] >= 0.1 * o["state_err"] or per_band[b] >= RC.T3_STATE, (o, per_band[b])
    print("   reference vs its second build: state err %s, outside T3 %d (bands %s), best index %d; device / (K x that) worst %.2f" % (
        rr["state_err"], rr["bands_outside_T3"], [o["band"] for o in rr["outside"]], best_alt, worst))
    # beyond T3 only where the reference's own two builds are beyond T3 of each other (or end with different pose counts)
    noisy = {o["band"] for o in rr["outside"]} | {b for b in range(B) if per_band[b] is None}
    assert {o["band"] for o in rep["outside"]} <= noisy, ([o["band"] for o in rep["outside"]], sorted(noisy))
    # every band beyond T3 (or with another pose count) must be one on which the reference's own linearisation noise decides: the CPU
    # oracle's two Jacobian modes - neither involves the device - disagree there too
    suspects = [o["band"] for o in rep["outside"]] + [o["band"] for o in rep["pose_count_mismatch"]]
    if suspects:
        sub = _abi.TebBatchHost(len(suspects), batch.stride)
        for k, b in enumerate(suspects):
            sub.set_teb(k, *batch.get_teb(b))
            sub.has_vel_start[k] = batch.has_vel_start[b]; sub.has_vel_goal[k] = batch.has_vel_goal[b]
            sub.vel_start[k] = batch.vel_start[b]; sub.vel_goal[k] = batch.vel_goal[b]
            sub.prefer_rotdir[k] = batch.prefer_rotdir[b]; sub.via_points_enabled[k] = batch.via_points_enabled[b]
        tols = sensitivity.band_tolerances(oracle, cfg, obst, via, sub, threads=THREADS)
        for k, b in enumerate(suspects):
            assert tols[k] is None or tols[k] > sensitivity.WELL_CONDITIONED_TOL, ("band %d is beyond T3 although the oracle calls it well conditioned" % b, tols[k], rep)


@pytest.mark.parametrize("name", list(CASES))
def test_numeric_mode_matches_oracle_at_full_size(oracle, name):
    """The secondary bench number (c4_g2o_numeric_jacobians) and the other configurations in the reference's own linearisation
    scheme: device vs the CPU oracle in the SAME mode (which is bit-equal to the reference's code, tests/test_reference_pinning.py).
    Status, pose counts and the whole LM sequence (iterations, damping trials per iteration) identical on every band. The state is
    compared with the per-band yardstick of tests/sensitivity.py - 2e-5 where the oracle's own two Jacobian modes agree to 2e-6, 10 x
    their distance elsewhere (capped at 5e-3): the device's sin / cos differ from glibc's in the last bit, and the central differences
    divide that by 2e-9, i.e. the device sits as far from the oracle as a second compiler of the reference would. At least
    NUMERIC_STATE_MIN_INSIDE[name] bands must be inside their bound; the others are printed."""
    cfg, obst, via, batch = CASES[name]()
    B = batch.count
    tols = sensitivity.band_tolerances(oracle, cfg, obst, via, batch, threads=THREADS)
    cfg.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
    out, res, tr, ms = RC.run_device_traced(planner, cfg, obst, via, batch)
    oout, ores, otr = oracle.optimize_batch(cfg, obst, via, batch, threads=THREADS, trace=True)
    inside = 0; worst_well = 0.0; well = 0
    for b in range(B):
        assert int(res.status[b]) == int(ores.status[b]) and int(out.n[b]) == int(oout.n[b]), (name, b)
        div = RC.first_divergence(tr[b], otr[b])
        assert div is None or div[0] != "accept/reject", (name, b, div)
        d = RC.state_error(out.get_teb(b), oout.get_teb(b))
        if np.isfinite(ores.cost[b]) and ores.cost[b] != 0:
            d = max(d, abs(res.cost[b] - ores.cost[b]) / abs(ores.cost[b]))
        bound = tols[b] if tols[b] is not None else sensitivity.ILL_CONDITIONED_CAP
        if d <= bound:
            inside += 1
        else:
            print("   band %d: %.2e beyond its bound %.2e (first chi2 difference at LM iteration %s)" % (b, d, bound, div[1] if div else None))
        if tols[b] is not None and tols[b] <= sensitivity.WELL_CONDITIONED_TOL:
            well += 1; worst_well = max(worst_well, d)
    print("%s numeric mode vs oracle: %d bands, LM sequences identical on all, %d inside their bound, %d well conditioned (worst %.2e), "
          "kernel %.3f ms" % (name, B, inside, well, worst_well, ms))
    assert inside >= NUMERIC_STATE_MIN_INSIDE[name], (inside, B, NUMERIC_STATE_MIN_INSIDE[name])
