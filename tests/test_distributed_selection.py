"""N > 1 path on CPU: two gloo ranks shard a candidate batch, optimise their shards with the ORACLE standing in
for the GPU kernel (the exchange step is what is under test), and agree on the same best trajectory as a
single-process selectBestTeb over the whole batch — including ties and hysteresis / prefer-initial multipliers."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from teb_local_planner_amd import parallel, scenes, _abi  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if case == "oracle_shards":
            from oracle import oracle_py
            cfg, obst, via, batch = scenes.scene_small_mixed(B=6, stride=192)
            lo, hi = parallel.shard_range(batch.count, rank, world)
            sub = _abi.TebBatchHost(hi - lo, batch.stride)
            for k, b in enumerate(range(lo, hi)):
                sub.set_teb(k, *batch.get_teb(b))
                sub.has_vel_goal[k] = batch.has_vel_goal[b]
                sub.has_vel_start[k] = batch.has_vel_start[b]
                sub.vel_start[k] = batch.vel_start[b]
            _, res = oracle_py.optimize_batch(cfg, obst, via, sub)
            c, i = parallel.local_best(res.cost, offset=lo)
            gc, gi = parallel.select_best_distributed(c, i)
            q.put((rank, gc, gi, res.cost.tolist(), lo))
        elif case == "one_rank_unusable":
            # gloo-side mirror of teb_amd_select_best_distributed's error path (csrc/teb_amd.hip): the rank on which something went wrong
            # (bad handle, device mismatch, failed launch) still ENTERS the all-gather with the unusable record (DBL_MAX, -1) and reports
            # its error afterwards; its peers pick among the other ranks' candidates and nobody waits for ever
            costs = np.array([5.0, 1.0, 9.0, 3.0, 7.0, 4.0])
            lo, hi = parallel.shard_range(len(costs), rank, world)
            failed = (rank == 1)                       # rank 1 owns the overall minimum (index 1 or 2 .. depending on world) and fails
            c, i = parallel.UNUSABLE_RECORD if failed else parallel.local_best(costs[lo:hi], offset=lo)
            gc, gi = parallel.select_best_distributed(c, i)
            q.put((rank, gc, gi, failed, lo, hi))
        elif isinstance(case, str) and case.startswith("sharded_plan"):
            # HomotopyClassPlannerAmd::plan() sharded (host/teb_amd_hcp_backend.cpp): selection all-gather -> winner broadcast, with rank 1
            # failing (a) before the selection (exploration / optimise error), (b) inside the selection call after its record went out
            costs = np.array([5.0, 1.0, 9.0, 3.0, 7.0, 4.0])
            lo, hi = parallel.shard_range(len(costs), rank, world)
            bands = {g: (np.arange(4.0) + g, np.arange(4.0) * 2 + g, np.zeros(4) + 0.1 * g, np.ones(3) * (1 + g), g % 2 == 0, 10.0 * g) for g in range(lo, hi)}
            owner_of = lambda g: next(r for r in range(world) if parallel.shard_range(len(costs), r, world)[0] <= g < parallel.shard_range(len(costs), r, world)[1])
            failing = rank == 1
            mode = case.split(":")[1]
            if mode == "nobody_has_a_candidate":
                out = parallel.sharded_plan_exchange(True, parallel.UNUSABLE_RECORD, {}, owner_of, 8)
            elif mode == "between":          # rank 1 fails after the selection, before the broadcast: it says so in the broadcast's first round
                out = parallel.sharded_plan_exchange(True, parallel.local_best(costs[lo:hi], offset=lo), bands, owner_of, 8, fail_before_broadcast=failing)
            elif mode == "owner_band_too_long":   # the winner's band does not fit the message (capacity 3 < 4 poses): the OWNER finds out, everyone hears
                out = parallel.sharded_plan_exchange(True, parallel.local_best(costs[lo:hi], offset=lo), bands, owner_of, 3)
            else:
                out = parallel.sharded_plan_exchange(not (failing and mode == "before"), parallel.local_best(costs[lo:hi], offset=lo), bands, owner_of, 8,
                                                     fail_inside_selection=failing and mode == "inside")
            ok, gi, band = out
            q.put((rank, ok, gi, None if band is None else [np.asarray(a).tolist() if hasattr(a, "__len__") else a for a in band], lo, hi))
        else:
            res = []
            for cs in case:
                costs = np.array(cs["costs"], dtype=np.float64)
                lo, hi = parallel.shard_range(len(costs), rank, world)
                c, i = parallel.local_best(costs[lo:hi], offset=lo, last_best=cs["last_best"],
                                           initial_plan=cs["initial_plan"], hysteresis=cs["hyst"],
                                           prefer_initial=cs["prefer"])
                res.append(parallel.select_best_distributed(c, i))
            q.put((rank, res))
    finally:
        dist.destroy_process_group()


def _run(world, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(outs)


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_selection_matches_single_process(world):
    from teb_local_planner_amd.config import TebConfig
    from oracle import oracle_py
    cases = [
        dict(costs=[5.0, 3.0, 9.0, 3.0, 7.0], last_best=-1, initial_plan=-1, hyst=1.0, prefer=1.0),   # tie across ranks
        dict(costs=[5.0, 3.0, 9.0, 3.0, 7.0], last_best=4, initial_plan=-1, hyst=0.4, prefer=1.0),    # hysteresis wins
        dict(costs=[5.0, 3.0, 9.0, 3.0, 7.0], last_best=-1, initial_plan=2, hyst=1.0, prefer=0.3),    # prefer-initial wins
        dict(costs=[4.0], last_best=-1, initial_plan=-1, hyst=1.0, prefer=1.0),                        # some ranks own nothing
    ]
    outs = _run(world, cases)
    for k, case in enumerate(cases):
        cfg = TebConfig()
        cfg.hcp.selection_cost_hysteresis = case["hyst"]
        cfg.hcp.selection_prefer_initial_plan = case["prefer"]
        ref_i, ref_c = oracle_py.select_best(cfg, case["costs"], case["last_best"], case["initial_plan"])
        for rank, res in outs:
            gc, gi = res[k]
            assert gi == ref_i and gc == ref_c, (case, rank, gc, gi, ref_c, ref_i)


def test_two_ranks_optimise_shards_and_agree():
    outs = _run(2, "oracle_shards")
    from oracle import oracle_py
    cfg, obst, via, batch = scenes.scene_small_mixed(B=6, stride=192)
    _, res = oracle_py.optimize_batch(cfg, obst, via, batch)
    ref_i, ref_c = oracle_py.select_best(cfg, res.cost)
    allc = {}
    for rank, gc, gi, costs, lo in outs:
        assert gi == ref_i and gc == ref_c
        for k, c in enumerate(costs):
            allc[lo + k] = c
    np.testing.assert_array_equal(np.array([allc[k] for k in range(6)]), res.cost)   # sharding does not change results


@pytest.mark.parametrize("world", [2, 3])
def test_a_failing_rank_sends_the_unusable_record_and_nobody_hangs(world):
    outs = _run(world, "one_rank_unusable")
    costs = np.array([5.0, 1.0, 9.0, 3.0, 7.0, 4.0])
    lo1, hi1 = parallel.shard_range(len(costs), 1, world)
    masked = costs.copy(); masked[lo1:hi1] = np.inf      # the failing rank's candidates take no part
    want = int(np.argmin(masked))
    for rank, gc, gi, failed, lo, hi in outs:
        assert gi == want and gc == costs[want], (rank, gc, gi, want)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["before", "inside", "nobody_has_a_candidate", "between", "owner_band_too_long"])
def test_sharded_plan_tick_survives_a_failing_rank(world, mode):
    """The collective sequence of the sharded HomotopyClassPlannerAmd::plan() (parallel.sharded_plan_exchange mirrors
    host/teb_amd_hcp_backend.cpp): rank 1 fails before the selection (its candidates take no part) or inside the selection
    call after its record went out (its candidate may still win: the peers saw a valid record) - either way every rank passes the
    all-gather AND the broadcast, the healthy ranks hold the winner's band with its batch statistics, the failed rank reports False."""
    outs = _run(world, "sharded_plan:" + mode)
    costs = np.array([5.0, 1.0, 9.0, 3.0, 7.0, 4.0])
    lo1, hi1 = parallel.shard_range(len(costs), 1, world)
    if mode == "nobody_has_a_candidate":
        assert all(ok and gi == -1 and band is None for _, ok, gi, band, _, _ in outs)
        return
    if mode in ("between", "owner_band_too_long"):
        # no rank waits for a strip that never comes: all of them return not-ok with the agreed index and no band
        assert all((not ok) and gi == 1 and band is None for _, ok, gi, band, _, _ in outs), outs
        return
    masked = costs.copy()
    if mode == "before":
        masked[lo1:hi1] = np.inf
    want = int(np.argmin(masked))
    for rank, ok, gi, band, lo, hi in outs:
        assert gi == want, (rank, gi, want)
        assert ok == (rank != 1)
        assert band is not None and band[0] == (np.arange(4.0) + want).tolist() and band[3] == [1.0 + want] * 3
        assert band[4] == (want % 2 == 0) and band[5] == 10.0 * want            # the statistics hasDiverged reads came with the band
