"""how far the closed-form Jacobian mode (the benchmarked one) lands from the reference's own code - which
differentiates numerically, delta = 1e-9 central differences - over SEEDED scene families, not one scene: 20 C5-class scenes (car-like,
polygon footprint vs convex polygon obstacles, 150 .. 300 poses) and 20 C2-class scenes (diff-drive, point robot vs point obstacles, 200
poses). Per scene: device (closed forms) vs oracle/_ref/libteb_ref.so (src/optimal_planner.cpp compiled in place), state error / T3,
pose counts, LM sequence; beside it the device in the g2o-numeric mode (same method as the reference: what is left is rounding) and the
reference's second build (libteb_ref_alt.so: the reference's own build-to-build noise on that scene). A scene beyond T3 is classified
with tests/sensitivity.py (the CPU oracle's two Jacobian modes - no device involved - disagree there too: ill-conditioned) or is a finding.
usage (GPU box): python tools/analytic_margin.py > profiles/analytic_margin_<tag>.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sensitivity
from teb_local_planner_amd import planner, scenes, _abi
from oracle import ref_py, ref_alt_py, refcode_compare as RC, oracle_py

oracle_py.build()
have_alt = os.path.exists(ref_alt_py.SO)


def families():
    rng = np.random.default_rng(606)
    for k in range(20):
        n = int(rng.integers(150, 301)); seed = 7000 + k
        yield "C5-class", "seed %d n %d" % (seed, n), (lambda n=n, seed=seed: scenes.scene_c5(n=n, M=300, seed=seed, stride=336, length=0.1 * n))
    for k in range(20):
        seed = 8000 + k
        yield "C2-class", "seed %d" % seed, (lambda seed=seed: scenes.scene_c2(n=200, M=100, seed=seed, stride=232))


def device(cfg, obst, via, batch, mode):
    cfg.jacobian_mode = mode
    out, res, tr, ms = RC.run_device_traced(planner, cfg, obst, via, batch)
    cfg.jacobian_mode = _abi.JACOBIAN_ANALYTIC
    return out, res, tr


def mixed_family():
    """20 randomised small scenes with every option of the path toggled at random (tests/random_cases.py: footprint kind, holonomic,
    car-like, exact arc length, legacy association, cost exponent, shortest path, velocity-obstacle ratio, ordered via-points, ..), three
    bands each: per band the same comparison."""
    import random_cases
    out = []
    for seed in range(20):
        cfg, obst, via, batch = random_cases.random_case(seed)
        rout, rok, rcost, rit, rtr = ref_py.optimize_batch(cfg, obst, via, batch, threads=batch.count, trace=True)
        oa, ra, ta = device(cfg, obst, via, batch, _abi.JACOBIAN_ANALYTIC)
        on, rn, tn = device(cfg, obst, via, batch, _abi.JACOBIAN_G2O_NUMERIC)
        aout = None
        if have_alt:
            aout = ref_alt_py.optimize_batch(cfg, obst, via, batch, threads=batch.count, trace=True)[0]
        tols = None
        for b in range(batch.count):
            same_n = int(oa.n[b]) == int(rout.n[b])
            da = RC.state_error(oa.get_teb(b), rout.get_teb(b)) if same_n else float("nan")
            dn = RC.state_error(on.get_teb(b), rout.get_teb(b)) if int(on.n[b]) == int(rout.n[b]) else float("nan")
            dalt = RC.state_error(aout.get_teb(b), rout.get_teb(b)) if (aout is not None and int(aout.n[b]) == int(rout.n[b])) else float("nan")
            div = RC.first_divergence(ta[b], rtr[b])
            note = ""
            if not same_n or not (da <= RC.T3_STATE):
                if tols is None:
                    tols = sensitivity.band_tolerances(oracle_py, cfg, obst, via, batch, threads=batch.count)
                tol = tols[b]
                note = ("ILL-CONDITIONED by tests/sensitivity.py (tolerance %s)" % tol if (tol is None or tol > sensitivity.WELL_CONDITIONED_TOL)
                        else "FINDING: beyond T3 on a band the oracle calls well conditioned")
            label = "seed %d band %d" % (seed, b)
            out.append((label, int(rout.n[b]), int(oa.n[b]), da, dn, dalt, div, bool((int(ra.status[b]) == _abi.TEB_OK) == bool(rok[b])), note))
            print("mixed-class %-16s poses ref %3d dev %3d | closed forms %.2e = %.3f T3 | numeric mode %.2e | reference's 2nd build %.2e | LM sequence %s%s" % (
                label, int(rout.n[b]), int(oa.n[b]), da, da / RC.T3_STATE, dn, dalt,
                "equal" if (div is None or div[0] != "accept/reject") else "parts at iteration %d" % div[1], (" | " + note) if note else ""), flush=True)
    return out


rows = {}
if "--mixed" in sys.argv:
    rows["mixed-class"] = mixed_family()
for fam, label, mk in (families() if "--mixed-only" not in sys.argv else ()):
    cfg, obst, via, batch = mk()
    rout, rok, rcost, rit, rtr = ref_py.optimize_batch(cfg, obst, via, batch, threads=1, trace=True)
    oa, ra, ta = device(cfg, obst, via, batch, _abi.JACOBIAN_ANALYTIC)
    on, rn, tn = device(cfg, obst, via, batch, _abi.JACOBIAN_G2O_NUMERIC)
    same_n = int(oa.n[0]) == int(rout.n[0])
    da = RC.state_error(oa.get_teb(0), rout.get_teb(0)) if same_n else float("nan")
    dn = RC.state_error(on.get_teb(0), rout.get_teb(0)) if int(on.n[0]) == int(rout.n[0]) else float("nan")
    dalt = float("nan")
    if have_alt:
        aout, aok, acost, ait, atr = ref_alt_py.optimize_batch(cfg, obst, via, batch, threads=1, trace=True)
        dalt = RC.state_error(aout.get_teb(0), rout.get_teb(0)) if int(aout.n[0]) == int(rout.n[0]) else float("nan")
    div = RC.first_divergence(ta[0], rtr[0])
    note = ""
    if not same_n or not (da <= RC.T3_STATE):
        tol = sensitivity.band_tolerances(oracle_py, cfg, obst, via, batch, threads=1)[0]
        note = ("ILL-CONDITIONED by tests/sensitivity.py (the oracle's two Jacobian modes differ there too: tolerance %s)" % tol
                if (tol is None or tol > sensitivity.WELL_CONDITIONED_TOL) else "FINDING: beyond T3 on a band the oracle calls well conditioned")
    rows.setdefault(fam, []).append((label, int(rout.n[0]), int(oa.n[0]), da, dn, dalt, div, bool((int(ra.status[0]) == _abi.TEB_OK) == bool(rok[0])), note))
    print("%s %-18s poses ref %3d dev %3d | closed forms %.2e = %.3f T3 | numeric mode %.2e | reference's 2nd build %.2e | LM sequence %s%s" % (
        fam, label, int(rout.n[0]), int(oa.n[0]), da, da / RC.T3_STATE, dn, dalt,
        "equal" if (div is None or div[0] != "accept/reject") else "parts at iteration %d" % div[1], (" | " + note) if note else ""), flush=True)

print()
for fam, rs in rows.items():
    d = np.array([r[3] for r in rs]); ok = np.isfinite(d)
    dn = np.array([r[4] for r in rs]); dl = np.array([r[5] for r in rs])
    print("== %s: %d scenes, pose counts equal on %d, success flags equal on %d, LM accept / reject sequences equal on %d" % (
        fam, len(rs), int(sum(r[1] == r[2] for r in rs)), int(sum(r[7] for r in rs)), int(sum(r[6] is None or r[6][0] != "accept/reject" for r in rs))))
    q = lambda a: "p50 %.2e  p90 %.2e  max %.2e" % (np.nanpercentile(a, 50), np.nanpercentile(a, 90), np.nanmax(a))
    print("   closed forms vs reference code, state error     : %s   (/ T3: p50 %.3f  p90 %.3f  max %.3f)" % (
        q(d), np.nanpercentile(d, 50) / RC.T3_STATE, np.nanpercentile(d, 90) / RC.T3_STATE, np.nanmax(d) / RC.T3_STATE))
    print("   g2o-numeric mode vs reference code              : %s" % q(dn))
    print("   the reference's second build vs the reference   : %s" % q(dl))
    beyond = [r for r in rs if not (r[3] <= RC.T3_STATE)]
    print("   beyond T3 (or other pose count): %d%s" % (len(beyond), "".join("\n      %s: %s" % (r[0], r[8]) for r in beyond)))
