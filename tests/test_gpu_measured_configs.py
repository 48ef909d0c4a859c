"""Oracle parity on exactly the configurations bench.py quotes numbers on ("Next round" item 1):

  * the headline: BASELINE C4, 256 bands x 200 poses at the start, 450 static + 50 dynamic point obstacles, TebConfig defaults with
    teb_autosize ON, pose capacity 288 (band-form normal matrix in LDS + cyclic reduction on HBM blocks + 500-obstacle LDS cache,
    up to 100 autoResize sweeps per outer iteration) - every one of the 256 bands against the oracle, thread per TEB;
  * BASELINE C2 at full size (1 x 200 poses x 100 obstacles);
  * BASELINE C3 at full size (64 x 150 poses x 200 obstacles);
  * the secondary number: C4 with teb_autosize off (blocks in LDS), all 256 bands.

Same closed-form Jacobian mode on both sides. Stated tolerance (fp64), on EVERY band, no skip rule: status, pose count, LM iteration
and trial counts identical; poses / time differences <= 1e-7 (m, rad, s); cost and chi^2 rel 1e-7; same selectBestTeb index.
Observed on MI355X (tools/parity_probe.py): 211 of the 256 headline bands below 1e-12, the worst 3e-9."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import sensitivity  # noqa: E402

from teb_local_planner_amd import scenes, planner, _abi  # noqa: E402

pytestmark = pytest.mark.gpu

THREADS = os.cpu_count() or 1


def _run_gpu(cfg, obst, via, batch):
    s = planner.make_solver(cfg, obst, via, batch)
    s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, cfg.hcp.selection_obst_cost_scale,
               cfg.hcp.selection_viapoint_cost_scale, cfg.hcp.selection_alternative_time_cost)
    res = s.results()
    out = s.download(batch.copy())
    best = s.select_best()
    flags = s.debug_overflow_flags()
    s.close()
    assert not flags.any()
    return out, res, best


def _check(oracle, cfg, obst, via, batch, label):
    out, res, best = _run_gpu(cfg, obst, via, batch)
    ref, rres = oracle.optimize_batch(cfg, obst, via, batch, threads=THREADS)
    rep = sensitivity.compare_bands(out, res, ref, rres, None)
    print("%s: %d bands, status equal %d, all counts equal %d, state/cost checked on %d (skipped: %d), max state err %.2e, "
          "max cost rel %.2e" % (label, rep["bands"], rep["status_equal"], rep["counts_equal"], rep["checked"], rep["skipped"],
                                 rep["max_state_err"], rep["max_cost_rel"]))
    assert rep["status_equal"] == rep["bands"] and rep["counts_equal"] == rep["bands"], rep
    assert rep["checked"] == rep["bands"] and rep["skipped"] == 0, rep
    assert rep["max_state_err"] <= 1e-7 and rep["max_cost_rel"] <= 1e-7, rep
    np.testing.assert_allclose(res.chi2, rres.chi2, rtol=1e-7)
    assert best[0] == oracle.select_best(cfg, rres.cost)[0]
    return rep, out, res


def test_headline_c4_autosize_on_all_256_bands(oracle):
    cfg, obst, via, batch = scenes.scene_c4(B=256, n=200, seed=1004, stride=288)     # exactly bench.py's rank-0 workload
    assert cfg.trajectory.teb_autosize
    rep, out, res = _check(oracle, cfg, obst, via, batch, "C4 headline (autosize on, capacity 288)")
    assert (res.status == _abi.TEB_OK).all() and out.n.max() > 256        # the long-band path (two poses per lane) is exercised


def test_c4_fixed_200_all_256_bands(oracle):
    cfg, obst, via, batch = scenes.scene_c4(B=256, n=200, seed=1004, stride=208)
    cfg.trajectory.teb_autosize = False
    _check(oracle, cfg, obst, via, batch, "C4 fixed 200 poses (blocks in LDS)")


def test_c2_full_size(oracle):
    cfg, obst, via, batch = scenes.scene_c2(stride=208)
    assert batch.count == 1 and batch.n[0] == 200 and len(obst) == 100
    _check(oracle, cfg, obst, via, batch, "C2 1 x 200 x 100")


def test_c3_full_size_64_bands(oracle):
    cfg, obst, via, batch = scenes.scene_c3(stride=208)
    assert batch.count == 64 and len(obst) == 200
    _check(oracle, cfg, obst, via, batch, "C3 64 x 150 x 200")


def test_c4_strong_shard_of_8_32_bands(oracle):
    """bench.py's secondary.c4_strong_shard_of_8: bands 0 .. 31 of the headline batch = rank 0's shard of BASELINE config 4 (256
    candidates over 8 GPUs, parallel.shard_range(256, 0, 8)). 32 bands leave CUs idle: the launch runs with solver helpers, and its
    bands must still be the oracle's - and, bit for bit, the bands 0 .. 31 of the whole batch launched at once."""
    from teb_local_planner_amd import parallel
    lo, hi = parallel.shard_range(256, 0, 8)
    assert (lo, hi) == (0, 32)
    cfg, obst, via, full = scenes.scene_c4(B=256, n=200, seed=1004, stride=288)
    shard = _abi.TebBatchHost(hi - lo, 288)
    for k, b in enumerate(range(lo, hi)):
        shard.set_teb(k, *full.get_teb(b))
        shard.has_vel_goal[k] = full.has_vel_goal[b]
    s = planner.make_solver(cfg, obst, via, shard)
    s.optimize(cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations, True, cfg.hcp.selection_obst_cost_scale,
               cfg.hcp.selection_viapoint_cost_scale, cfg.hcp.selection_alternative_time_cost)
    helpers = s.last_launch_info()
    s.close()
    assert helpers[1] >= 1 and not helpers[2], helpers      # the shard's launch has solver helpers (the mode the number is quoted on)
    rep, out, res = _check(oracle, cfg, obst, via, shard, "C4 strong shard of 8 (bands 0..31, solver helpers)")
    whole, wres, _ = _run_gpu(cfg, obst, via, full)
    for k, b in enumerate(range(lo, hi)):
        assert out.n[k] == whole.n[b]
        for u, v in zip(out.get_teb(k), whole.get_teb(b)):
            assert np.array_equal(u, v), (k, b)
    assert np.array_equal(res.cost, wres.cost[lo:hi]) and np.array_equal(res.lm_trials, wres.lm_trials[lo:hi])
