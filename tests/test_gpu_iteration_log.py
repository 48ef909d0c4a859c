"""Opt-in per-iteration log of the LM loop (teb_amd_set_iteration_log / teb_amd_get_iteration_log): the data of the line g2o prints
per iteration when the reference switches SparseOptimizer::setVerbose on (src/optimal_planner.cpp:384)."""
import numpy as np
import pytest

from teb_local_planner_amd import scenes, planner, _abi
from oracle import refcode_compare as RC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene", ["c2", "mixed_polygon", "c3_small"])
def test_log_is_consistent_with_results_and_equals_the_oracle_trace(oracle, scene):
    if scene == "c2":
        cfg, obst, via, batch = scenes.scene_c2(stride=208)
    elif scene == "mixed_polygon":
        cfg, obst, via, batch = scenes.scene_small_mixed(footprint="polygon")
    else:
        cfg, obst, via, batch = scenes.scene_c3(B=6, n=60, M=40, stride=208)
    out, res, tr, ms = RC.run_device_traced(planner, cfg, obst, via, batch)
    oout, ores, otr = oracle.optimize_batch(cfg, obst, via, batch, trace=True)
    for b in range(batch.count):
        t = tr[b]
        assert len(t) == res.lm_iterations[b] == len(otr[b])
        assert int(t[:, 2].sum()) == res.lm_trials[b]                      # damping trials add up to the reported total
        assert t[-1, 0] == res.chi2[b] and t[-1, 1] == res.lambda_[b]      # last row = the results of the band
        assert t[-1, 3] == out.n[b]
        np.testing.assert_array_equal(t[:, 2:], otr[b][:, 2:])             # same accept / reject sequence, same pose counts
        np.testing.assert_allclose(t[:, 0], otr[b][:, 0], rtol=1e-7)       # chi2 after every iteration
        np.testing.assert_allclose(t[:, 1], otr[b][:, 1], rtol=1e-6)       # lambda after every iteration


def test_log_is_off_by_default_and_fails_loudly():
    cfg, obst, via, batch = scenes.scene_small_mixed()
    s = planner.make_solver(cfg, obst, via, batch)
    s.optimize(2, 1)
    with pytest.raises(planner.TebAmdError) as e:
        s.iteration_log(0)
    assert e.value.code == _abi.ERR_INVALID_ARG
    s.set_iteration_log(True)
    s.upload(batch)
    s.optimize(2, 1)
    assert len(s.iteration_log(0)) == s.results().lm_iterations[0]
    s.close()


def test_pose_count_beyond_the_capacity_is_refused_by_the_kernel():
    """Memory safety does not rest on the host's cached upper bound of the pose counts: a count written behind the host's
    back that exceeds the LDS strips of the launch makes that band fail; it does not run off the LDS, and its neighbours are untouched.
    Both launch paths: the handle's own layout (small capacity) and the optimistic per-launch layout of a handle made for long bands."""
    import ctypes as C
    for stride in (96, 400):
        cfg, obst, via, batch = scenes.scene_small_mixed(stride=stride)
        ref = planner.make_solver(cfg, obst, via, batch)
        ref.optimize(5, 4, True)
        want = ref.download(batch.copy())
        ref.close()
        s = planner.make_solver(cfg, obst, via, batch)
        L = planner.lib()
        L.teb_amd_debug_poke_pose_count.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        assert L.teb_amd_debug_poke_pose_count(s._h, 1, stride + 7) == 0
        s.optimize(5, 4, True)
        res = s.results()
        assert res.status[1] == _abi.TEB_FAILED and res.status[0] == _abi.TEB_OK and res.status[2] == _abi.TEB_OK, res.status
        assert s.debug_overflow_flags()[1] & 2
        assert L.teb_amd_debug_poke_pose_count(s._h, 1, int(batch.n[1])) == 0   # (download wants a sane count)
        got = s.download(batch.copy())
        for b in (0, 2):
            for u, v in zip(got.get_teb(b), want.get_teb(b)):
                np.testing.assert_array_equal(u, v)
        s.close()


def test_pose_count_beyond_the_optimistic_layout_repeats_the_launch():
    """A count the handle's capacity holds but the per-launch (optimistic) layout does not: the first launch flags the band, the host
    repeats the batch in the handle's own layout; the other bands end exactly as without the poke."""
    import ctypes as C
    cfg, obst, via, batch = scenes.scene_small_mixed(stride=400)
    ref = planner.make_solver(cfg, obst, via, batch)
    ref.optimize(5, 4, True)
    want = ref.download(batch.copy())
    ref.close()
    s = planner.make_solver(cfg, obst, via, batch)
    L = planner.lib()
    L.teb_amd_debug_poke_pose_count.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    assert L.teb_amd_debug_poke_pose_count(s._h, 1, 300) == 0   # host still believes max n = 24 -> blocks-in-LDS layout of 238 poses
    s.optimize(5, 4, True)
    res = s.results()
    assert res.status[0] == _abi.TEB_OK and res.status[2] == _abi.TEB_OK, res.status
    assert L.teb_amd_debug_poke_pose_count(s._h, 1, int(batch.n[1])) == 0
    got = s.download(batch.copy())
    for b in (0, 2):
        for u, v in zip(got.get_teb(b), want.get_teb(b)):
            np.testing.assert_allclose(u, v, rtol=0, atol=1e-9)   # (another layout: same arithmetic up to the order of the block reduction)
    s.close()
