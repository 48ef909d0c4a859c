"""The chain-parallel autoResize sweep (csrc/teb_autoresize_chain.hpp, the device's source compiled for the host) against the oracle's
sequential sweep (src/timed_elastic_band.cpp:227-286): same time differences, same poses, on random bands of every flavour."""
import ctypes as C
import math
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import oracle_py

HERE = os.path.dirname(os.path.abspath(__file__))
NEW = 1024
SEEDS = {"near": 1, "long": 2, "short": 3, "creep": 4, "exact": 5}


@pytest.fixture(scope="module")
def host():
    so = os.path.join(tempfile.mkdtemp(prefix="teb_ar_"), "libar_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           os.path.join(HERE, "host", "autoresize_chains_host.cpp"), "-o", so])
    L = C.CDLL(so)
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.teb_host_autoresize_chains.restype = C.c_int
    L.teb_host_autoresize_chains.argtypes = [pd, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, pd, pi, pi, pd, pi]
    return L


def sweep(L, x, y, th, dt, dt_ref, hyst, min_samples, max_samples, cap=1024):
    """one sweep through the chains; returns (x, y, th, dt, info) or None when the chains decline"""
    Tin = len(dt)
    d = np.ascontiguousarray(dt, dtype=np.float64)
    odt = np.zeros(cap); desc = np.zeros(cap + 1, dtype=np.int32); rec = np.zeros(cap, dtype=np.int32)
    tail = np.zeros(1); info = np.zeros(10, dtype=np.int32)
    as_d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    as_i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    rc = L.teb_host_autoresize_chains(as_d(d), Tin, dt_ref, hyst, min_samples, max_samples, cap, as_d(odt), as_i(desc), as_i(rec), as_d(tail), as_i(info))
    assert rc == 0
    if info[5]:
        return None, info
    K, NN, tail_k = int(info[0]), int(info[1]), int(info[4])
    if tail_k >= 0:
        odt[tail_k] += tail[0]
    nx, ny, nth = np.zeros(NN), np.zeros(NN), np.zeros(NN)

    def pose(p):
        return (x[p], y[p], th[p]) if p < NEW else (nx[p - NEW], ny[p - NEW], nth[p - NEW])
    for depth in range(1, int(info[3]) + 1):
        for q in range(NN):
            r = int(rec[q])
            if (r >> 22) != depth:
                continue
            a, b = pose(r & 0x7ff), pose((r >> 11) & 0x7ff)
            nx[q], ny[q] = (a[0] + b[0]) / 2, (a[1] + b[1]) / 2
            sx, sy = math.cos(a[2]) + math.cos(b[2]), math.sin(a[2]) + math.sin(b[2])
            nth[q] = 0.0 if (sx == 0 and sy == 0) else math.atan2(sy, sx)
    P = [pose(int(desc[k])) for k in range(K + 1)]
    return (np.array([p[0] for p in P]), np.array([p[1] for p in P]), np.array([p[2] for p in P]), odt[:K].copy()), info


def random_band(rng, n, kind):
    x = np.cumsum(rng.uniform(0.05, 0.3, n)); y = np.cumsum(rng.normal(0, 0.05, n)); th = rng.uniform(-3.1, 3.1, n)
    if kind == "near":        # most intervals inside or close to the dead band
        dt = rng.normal(0.3, 0.07, n - 1).clip(0.01, None)
    elif kind == "long":      # repeated splits
        dt = rng.uniform(0.05, 2.5, n - 1)
    elif kind == "short":     # merge cascades
        dt = rng.uniform(0.005, 0.25, n - 1)
    elif kind == "creep":     # every interval slightly too long: one chain over the whole band
        dt = np.full(n - 1, 0.41) + rng.uniform(0, 0.01, n - 1)
    else:                     # mixture with exact boundary values
        dt = rng.choice([0.2, 0.4, 0.3, 0.6, 0.19999, 0.40001, 0.1, 0.75, 1.3], n - 1)
    return x, y, th, dt


@pytest.mark.parametrize("kind", ["near", "long", "short", "creep", "exact"])
def test_chain_sweep_equals_the_sequential_sweep(host, kind):
    rng = np.random.default_rng(SEEDS[kind] + 11)
    compared = declined = 0
    for case in range(120):
        n = int(rng.integers(3, 60 if kind == "long" else 300))   # (long intervals quadruple the band; sizeTimeDiffs() has to stay below max_samples)
        x, y, th, dt = random_band(rng, n, kind)
        min_s, max_s = (3, 500) if case % 4 else (int(rng.integers(3, n + 5)), int(rng.integers(max(4, n - 20), n + 60)))
        for sw in range(6):   # a few sweeps in a row: the later ones see the output of the earlier ones
            got, info = sweep(host, x, y, th, dt, 0.3, 0.1, min_s, max_s)
            X, Y, T, D = oracle_py.autoresize(x, y, th, dt, 0.3, 0.1, min_s, max_s, True)
            if got is None:
                declined += 1 if case % 4 else 0   # (the cases with random guards are meant to be declined now and then)
            else:
                compared += 1
                assert len(got[3]) == len(D), (kind, case, sw, info)
                np.testing.assert_array_equal(got[3], D)
                np.testing.assert_array_equal(got[0], X)
                np.testing.assert_array_equal(got[1], Y)
                np.testing.assert_allclose(got[2], T, rtol=0, atol=1e-15)
            if len(D) == len(dt) and np.array_equal(D, dt):
                break
            x, y, th, dt = X, Y, T, D
            if len(dt) < 2:
                break
    assert compared > 100, (compared, declined)
    if kind in ("near", "long", "short", "exact"):
        assert compared > 3 * declined, (compared, declined)


@pytest.mark.parametrize("kind", ["near", "long", "short", "exact"])
def test_modified_flag_drives_the_same_number_of_sweeps(host, kind):
    """the whole autoResize (sweeps until one modifies nothing, at most 100) through the chains' `modified` flag = the reference's loop"""
    rng = np.random.default_rng(SEEDS[kind] + 77)
    full = 0
    for case in range(60):
        n = int(rng.integers(3, 60 if kind == "long" else 250))
        x0, y0, th0, dt0 = random_band(rng, n, kind)
        x, y, th, dt = x0, y0, th0, dt0
        ok = True
        for sw in range(100):
            got, info = sweep(host, x, y, th, dt, 0.3, 0.1, 3, 500)
            if got is None:
                ok = False
                break
            x, y, th, dt = got
            if not info[2] or len(dt) < 1:
                break
        if not ok:
            continue
        full += 1
        X, Y, T, D = oracle_py.autoresize(x0, y0, th0, dt0, 0.3, 0.1, 3, 500, False)
        assert len(D) == len(dt), (kind, case)
        np.testing.assert_array_equal(dt, D)
        np.testing.assert_array_equal(x, X)
        np.testing.assert_allclose(th, T, rtol=0, atol=1e-14)
    assert full >= 30, full


def test_guards_make_the_chains_decline(host):
    """a sweep in which sizeTimeDiffs() reaches max_samples (or min_samples) belongs to the sequential machine"""
    rng = np.random.default_rng(5)
    x, y, th, dt = random_band(rng, 60, "long")
    got, info = sweep(host, x, y, th, dt, 0.3, 0.1, 3, 70)
    assert got is None and info[5] == 2
    x, y, th, dt = random_band(rng, 60, "short")
    got, info = sweep(host, x, y, th, dt, 0.3, 0.1, 50, 500)
    assert got is None and info[5] == 2


def test_long_chains_are_handed_to_the_sequential_machine(host):
    rng = np.random.default_rng(6)
    x, y, th, dt = random_band(rng, 200, "near")
    dt = np.full(199, 0.3); dt[0] = 0.41   # the excess of the first interval is handed on from interval to interval: one chain over the whole band
    got, info = sweep(host, x, y, th, dt, 0.3, 0.1, 3, 500)
    assert got is None and info[5] == 1
    dt[70] = 0.25                          # .. until an interval absorbs it: three chains of fewer than 64 evaluations would do, this one has 70
    got, info = sweep(host, x, y, th, dt, 0.3, 0.1, 3, 500)
    assert got is None and info[5] == 1
    dt[40] = 0.25                          # now every chain is short enough
    got, info = sweep(host, x, y, th, dt, 0.3, 0.1, 3, 500)
    X, Y, T, D = oracle_py.autoresize(x, y, th, dt, 0.3, 0.1, 3, 500, True)
    assert got is not None and info[7] <= 64
    np.testing.assert_array_equal(got[3], D)
