"""SURVEY section 8(f) row f3, arithmetic core: H-signatures of the device-resident bands (teb_hsig.hpp) through the C-ABI, against
the CPU oracle (bit-equal to the reference's h_signature.h, tests/test_reference_pinning.py) and the vectors of the reference code.

Tolerances: 3-D signature - only + - * / sqrt in the reference's order: expected bit-equal; asserted <= 4 ulp (the reference takes
pose differences in long double before rounding to double, a second rounding the device does not have). 2-D signature - the
reference works in complex<long double>, the device in fp64 with an explicit exponent: relative 1e-10 of the largest |value| of
the batch. Class decisions (keep / valid / reasonable): identical."""
import os
import sys
import time

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_ref_golden as RG  # noqa: E402

from teb_local_planner_amd import scenes, planner, _abi  # noqa: E402

pytestmark = pytest.mark.gpu


def _solver(cfg, obst, batch, mode, kern=None):
    cfg.obstacles.include_dynamic_obstacles = (mode == 3)   # HomotopyClassPlanner::calculateEquivalenceClass picks the class by this flag
    return planner.make_solver(cfg, obst, [], batch, options=_abi.Options(hsig3d_kernel=kern) if kern else None)   # kern pins one 3-D kernel


@pytest.mark.parametrize("mode", [2, 3])
def test_h_signatures_match_oracle_and_reference_vectors(oracle, mode):
    g = np.load(os.path.join(HERE, "golden", "ref_f3_hsig_%dd.npz" % mode))
    for cname, cfg, obst, batch in RG.h_signature_cases():
        s = _solver(cfg, obst, batch, mode)
        sig = s.h_signatures(1.0)
        want = oracle.h_signatures(cfg, obst, batch, mode, 1.0)
        np.testing.assert_array_equal(want, g[cname + "_sig"])
        if mode == 3:
            assert np.abs(sig - want).max() <= 4 * np.finfo(float).eps * max(1.0, np.abs(want).max()), (cname, np.abs(sig - want).max())
        else:
            assert np.abs(sig - want).max() <= 1e-10 * max(np.abs(want).max(), 1e-300), (cname, sig, want)
        keep, valid, reas = s.filter_equivalence_classes(0.1, -1, 1)
        okeep, ovalid, oreas = oracle.filter_equivalence_classes(mode, want, 0.1, -1, 1)
        np.testing.assert_array_equal(keep, okeep); np.testing.assert_array_equal(valid, ovalid); np.testing.assert_array_equal(reas, oreas)
        np.testing.assert_array_equal(valid, g[cname + "_valid"]); np.testing.assert_array_equal(reas, g[cname + "_reasonable"])
        s.close()


@pytest.mark.parametrize("mode", [2, 3])
def test_best_band_first_and_plans_in_current_class(oracle, mode):
    cname, cfg, obst, batch = RG.h_signature_cases()[2]
    s = _solver(cfg, obst, batch, mode)
    sig = s.h_signatures(1.0)
    for best, maxp in ((-1, 1), (3, 1), (3, 2), (5, 3), (0, 8)):
        got = s.filter_equivalence_classes(0.1, best, maxp)
        want = oracle.filter_equivalence_classes(mode, sig, 0.1, best, maxp)
        for u, v in zip(got, want):
            np.testing.assert_array_equal(u, v)
        if best >= 0:
            assert got[0][best] == 1      # the last best band always opens the list
    s.close()


def test_h_signature_3d_full_c4_batch(oracle):
    """256 candidates x 200 poses x 500 obstacles (50 moving): 2.56e8 Biot-Savart steps in one launch."""
    cfg, obst, via, batch = scenes.scene_c4()
    s = _solver(cfg, obst, batch, 3)
    s.h_signatures(1.0)                      # warm-up (first launch)
    t0 = time.perf_counter()
    sig = s.h_signatures(1.0)
    dt = time.perf_counter() - t0
    assert sig.shape == (256, 500) and np.isfinite(sig).all()
    sub = _abi.TebBatchHost(4, batch.stride)
    for k, b in enumerate((0, 17, 128, 255)):
        sub.set_teb(k, *batch.get_teb(b))
    want = oracle.h_signatures(cfg, obst, sub, 3, 1.0)
    for k, b in enumerate((0, 17, 128, 255)):
        assert np.abs(sig[b] - want[k]).max() <= 4 * np.finfo(float).eps
    keep, valid, reas = s.filter_equivalence_classes(0.1, -1, 1)
    assert valid.all() and keep[0] == 1 and 1 <= keep.sum() <= 256
    print("3-D H-signatures of the C4 batch incl. download: %.2f ms, %d classes" % (1e3 * dt, int(keep.sum())))
    s.close()


def test_h_signature_2d_large_obstacle_count_stays_in_range(oracle):
    """A few hundred obstacles push A_l = f0 * prod 1/(o_l - o_j) out of the fp64 range (the reference uses long double):
    the device keeps an explicit exponent and still agrees."""
    cfg, obst, via, batch = scenes.scene_c3(B=4, n=60, M=400, stride=64)
    s = _solver(cfg, obst, batch, 2)
    sig = s.h_signatures(1.0)
    want = oracle.h_signatures(cfg, obst, batch, 2, 1.0)
    assert np.isfinite(sig).all()
    assert np.abs(sig - want).max() <= 1e-10 * max(np.abs(want).max(), 1e-300) + 1e-300
    s.close()


def test_3d_signature_kernels_return_identical_bits(oracle):
    """hsig3d_kernel (one lane per band x obstacle) and hsig3d_small_kernel (lanes over obstacle x segment, sequential sum by one lane
    per obstacle) perform the same operations in the same order: identical results, on every case incl. ragged band lengths."""
    for cname, cfg, obst, batch in RG.h_signature_cases():
        res = []
        for kern in ("wide", "small"):
            s = _solver(cfg, obst, batch, 3, kern)
            res.append(s.h_signatures(1.0).copy())
            s.close()
        np.testing.assert_array_equal(res[0], res[1])
        assert np.isfinite(res[0]).all() and np.abs(res[0]).max() > 0


@pytest.mark.parametrize("seed", range(0, 80, 4))
@pytest.mark.parametrize("mode", [2, 3])
def test_randomized_h_signatures_match_oracle(oracle, mode, seed):
    """Random scenes (tests/random_cases.py; the oracle is bit-equal to the reference's code on them): signatures and class decisions on
    the device, both 3-D kernels."""
    from random_cases import random_case
    cfg, obst, via, batch = random_case(seed)
    want = oracle.h_signatures(cfg, obst, batch, mode, 1.0)
    for kern in (("wide", "small") if mode == 3 else (None,)):
        s = _solver(cfg, obst, batch, mode, kern)
        sig = s.h_signatures(1.0)
        if mode == 3:
            assert np.abs(sig - want).max(initial=0) <= 4 * np.finfo(float).eps * max(1.0, np.abs(want).max(initial=0)), (kern, np.abs(sig - want).max())
        else:
            assert np.abs(sig - want).max() <= 1e-10 * max(np.abs(want).max(), 1e-300)
        got = s.filter_equivalence_classes(0.1, -1, 1)
        for u, v in zip(got, oracle.filter_equivalence_classes(mode, want, 0.1, -1, 1)):
            np.testing.assert_array_equal(u, v)
        s.close()
