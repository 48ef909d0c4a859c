"""Multi-GPU path on the GPU box. The box has ONE GPU, and RCCL refuses two ranks on one device, so what can be
exercised here is: the C-ABI exchange on a world-size-1 communicator (every RCCL call of the path runs: ncclCommInitRank,
ncclAllGather, ncclBroadcast), its agreement with the single-GPU selection incl. the hysteresis / prefer-initial multipliers on global
indices, the winner-strip broadcast, and that `bench.py --gpus N` either runs N real ranks or fails loudly - never a silent single rank.
World sizes 2 / 3 of the same reduction run on CPU (gloo) in tests/test_distributed_selection.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from teb_local_planner_amd import scenes, planner, parallel, _abi  # noqa: E402

pytestmark = pytest.mark.gpu


def test_c_abi_exchange_on_a_world_of_one(oracle):
    cfg, obst, via, batch = scenes.scene_c3(B=8, n=60, M=40, stride=128)
    cfg.hcp.selection_cost_hysteresis = 0.5
    cfg.hcp.selection_prefer_initial_plan = 0.7
    s = planner.make_solver(cfg, obst, via, batch)
    s.optimize(5, 4, True, 100.0, 1.0, False)
    res = s.results()
    comm = parallel.RcclComm(parallel.RcclComm.unique_id(), 0, 1, 0)
    offset = 1000                                   # this rank's candidates are the global indices 1000 .. 1007
    for lb, ip in ((-1, -1), (3, -1), (-1, 5), (2, 2), (7, 0)):
        want_i, want_c = oracle.select_best(cfg, res.cost, lb, ip)
        g, c, owner = s.select_best_distributed(comm, offset, offset + lb if lb >= 0 else -1, offset + ip if ip >= 0 else -1)
        assert (g, owner) == (offset + want_i, 0) and c == want_c
        assert s.select_best(lb, ip) == (want_i, want_c)
    # multipliers addressed to candidates of OTHER ranks leave this rank's costs alone
    g, c, _ = s.select_best_distributed(comm, offset, 3, 5)
    assert (g - offset, c) == oracle.select_best(cfg, res.cost, -1, -1)
    # winner strip: what every rank receives is the owner's resident band
    best = g - offset
    x, y, th, dt = s.broadcast_band(comm, 0, best, 128)
    out = s.download(batch.copy())
    for u, v in zip((x, y, th, dt), out.get_teb(best)):
        np.testing.assert_array_equal(u, v)
    with pytest.raises(planner.TebAmdError):
        s.broadcast_band(comm, 0, best, 8)          # capacity below the winner's pose count: loud
    comm.close()
    s.close()


def test_a_rank_with_a_bad_handle_still_enters_the_all_gather():
    """whatever is wrong on ONE rank - a null handle, no output pointer - it contributes the unusable record
    (DBL_MAX, -1) to the all-gather and returns its error afterwards; it never leaves its peers waiting. On a world of one the proof is
    that the call returns an error AND the communicator is still usable (a collective that had been skipped by one of two ranks would
    have hung; here the collective demonstrably ran: the next call on the same communicator gives the right answer)."""
    import ctypes as C
    cfg, obst, via, batch = scenes.scene_c3(B=8, n=60, M=40, stride=128)
    s = planner.make_solver(cfg, obst, via, batch)
    s.optimize(5, 4, True, 100.0, 1.0, False)
    comm = parallel.RcclComm(parallel.RcclComm.unique_id(), 0, 1, 0)
    L = planner.lib()
    best = C.c_int32(-7); cost = C.c_double(0); owner = C.c_int32(-7)
    rc = L.teb_amd_select_best_distributed(None, comm._c, 0, -1, -1, C.byref(best), C.byref(cost), C.byref(owner))
    assert rc != 0 and b"unusable record" in L.teb_amd_last_error()
    rc = L.teb_amd_select_best_distributed(s._h, comm._c, 0, -1, -1, None, None, None)
    assert rc != 0 and b"unusable record" in L.teb_amd_last_error()
    g, c, o = s.select_best_distributed(comm, 10)
    assert g >= 10 and (g - 10, c) == s.select_best(-1, -1) and o == 0
    comm.close()
    s.close()


def _bench(*extra, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                           "--no-parity-check", "--latency-reps", "0"] + list(extra), env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_flag_is_real():
    import torch
    have = torch.cuda.device_count()
    r = _bench("--gpus", str(have + 1))
    assert r.returncode != 0 and "refusing to fake a multi-GPU run" in (r.stderr + r.stdout)
    r = _bench("--gpus", "1", "--scaling", "strong")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["config"]["tebs_total"] == 256
    if have >= 2:                                    # a multi-GPU box: N real ranks, the exchange inside libteb_amd.so
        r = _bench("--gpus", "2")
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and "libteb_amd.so" in line["config"]["exchange"]
