#!/usr/bin/env python3
"""bench.py — throughput of the TEB hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

One "step" = one pass of the hot path over one batch: B x optimizeTEB (4 outer x 5 inner LM iterations,
association, cost) = HomotopyClassPlanner::optimizeAllTEBs on the resident batch, from the same initial
state every step (device-to-device restore inside the timed region), with inputs already in HBM.
Workload at N=1: BASELINE config C4 — 256 candidate TEBs x 200 poses, 500 point obstacles incl. 50 dynamic
(teb_autosize off so n stays 200). N>1: every rank runs its own 256 candidates (weak scaling),
no data-path collective; one 16-byte all-gather per step performs the best-trajectory selection.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from teb_local_planner_amd import scenes, planner, parallel, _abi  # noqa: E402

# ALGORITHMIC bytes per TEB.LM-iteration (restated in DESIGN.md §Measurement):
#   64*n (state read+write) + R_obst + 4*E_assoc + 32  + (32*n + R_obst + 4*E_assoc)/inner  [association amortised]
def alg_bytes_per_unit(n, M, e_assoc, inner):
    r_obst = 32 * M
    return 64 * n + r_obst + 4 * e_assoc + 32 + (32 * n + r_obst + 4 * e_assoc) / inner


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tebs", type=int, default=256, help="candidate TEBs per GPU")
    ap.add_argument("--poses", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=256, help="TEBs in the CPU-oracle sample")
    ap.add_argument("--latency-reps", type=int, default=20)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:   # plan latencies, secondary configurations and the CPU baseline are N = 1 items (contract: rank 0 at N = 1 only)
        args.latency_reps = 0
        args.no_cpu_baseline = True
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    B, n = args.tebs, args.poses
    # every rank owns its own candidates (different seed -> different bands), the scene is replicated
    # TebConfig defaults throughout (teb_autosize on: the bands are resized on the device every outer iteration and end with
    # 190 .. 290 poses); capacity 288 poses per band = band-form normal matrix in LDS + the 500-obstacle LDS cache
    STRIDE = max(288, n)
    cfg, obst, via, batch = scenes.scene_c4(B=B, n=n, seed=1004 + 7919 * rank, stride=STRIDE)
    inner, outer = cfg.optim.no_inner_iterations, cfg.optim.no_outer_iterations
    hp = planner.HomotopyClassPlanner(cfg, obst, via, batch, device=local_rank)
    s = hp.solver
    s.snapshot()
    def step():
        s.restore()
        hp.optimizeAllTEBs(inner, outer)
        best, cost = s.select_best(-1, -1)          # K9 on the resident costs; synchronises the stream (16-byte D2H)
        if distributed:                             # the path's only exchange: one (cost, global index) record per rank
            return parallel.select_best_distributed(cost, rank * B + best, device="cuda")[1]
        return best

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(s.last_kernel_ms())        # HIP events on the launch stream
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    res = s.results()
    units_step = int(res.lm_iterations.sum())
    n_after = s.pose_counts()
    tebs_ok = int((res.status == 0).sum())
    tt = torch.tensor([elapsed, float(units_step)], dtype=torch.float64, device="cuda")
    if distributed:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max, units_all = float(tmax[0]), float(tsum[1])
    else:
        elapsed_max, units_all = elapsed, float(units_step)

    out = None
    if rank == 0:
        value = units_all * args.steps / elapsed_max
        kms = float(np.mean(kernel_ms))
        M = len(obst)
        # association list size for the algorithmic-byte model: measured on TEB 0 of this rank
        s.restore()                                   # association list size of the initial band of TEB 0
        dbg = s.debug_linearize(0, n, 1.0)
        e_assoc = len(dbg["assoc_pose"])
        n_eff = float(n_after.mean())                 # pose count the iterations of this step actually worked on
        abu = alg_bytes_per_unit(n_eff, M, e_assoc, inner)
        alg_bytes_launch = abu * units_step
        achieved = alg_bytes_launch / (kms * 1e-3) / 1e9
        # HBM-side traffic per launch from the committed rocprofv3 PMC passes of THIS command (FETCH_SIZE / WRITE_SIZE in
        # separate runs, scaled by the factors calibrated on a known 1 GiB stream; profiles/rocprof_*_summary.json)
        traffic, traffic_src, fp64 = None, None, None
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "rocprof_r*_summary.json")))
            if cands and B == 256 and n == 200:
                pj = json.load(open(cands[-1]))
                cal = pj.get("calibration", {})
                fr = cal.get("FETCH_SIZE_bytes_per_counted_KB", 2048.0)
                fw = cal.get("WRITE_SIZE_bytes_per_counted_KB", 1024.0)
                traffic = pj["FETCH_SIZE_KB_per_launch"] * fr + pj["WRITE_SIZE_KB_per_launch"] * fw
                traffic_src = os.path.basename(cands[-1])
                if pj.get("fp64_flop_per_launch"):   # PMC pass SQ_INSTS_VALU_*_F64 of this command (instruction counts x 64 lanes)
                    fp64 = {"flop_per_launch": pj["fp64_flop_per_launch"], "achieved": pj["fp64_flop_per_launch"] / (kms * 1e-3) / 1e12,
                            "peak": 78.6, "unit": "TFLOP/s", "source": traffic_src,
                            "note": "vector fp64 (no MFMA on this path); peak = AMD spec, half the 157.3 TFLOP/s fp32 vector rate; "
                                    "the build uses -ffp-contract=off, so mul+add pairs issue as two instructions (ceiling ~39 TFLOP/s)"}
                    fp64["frac"] = fp64["achieved"] / fp64["peak"]
        except Exception:
            traffic = None
        out = {
            "metric": "TEB LM iterations/sec (whole node)", "value": value, "unit": "TEB.LM-iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4: %d candidate TEBs/GPU x %d poses at the start (teb_autosize on, the reference default: "
                                   "%d..%d poses, mean %.0f, after the step), %d point obstacles (%d dynamic), diff-drive, point "
                                   "footprint, TebConfig defaults, 4 outer x 5 inner" %
                                   (B, n, int(n_after.min()), int(n_after.max()), n_eff, M, int(np.sum(obst.dynamic))),
                       "tebs_per_gpu": B, "poses": n, "poses_after": [int(n_after.min()), int(n_after.max())],
                       "pose_capacity": STRIDE, "tebs_ok": tebs_ok, "obstacles": M, "units_per_step_per_gpu": units_step,
                       "lm_trials_per_step_per_gpu": int(res.lm_trials.sum())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "kernel": "teb_optimize_kernel", "kernel_ms": kms,
                         "alg_bytes_per_unit": abu, "alg_bytes_per_launch": alg_bytes_launch},
        }
        if fp64:
            out["roofline"]["valu_fp64"] = fp64
        # ---- p50 plan()-equivalent latency on the 200-pose band: upload -> 4x5 iterations incl. autoResize,
        #      association, cost -> select -> download (single TEB, config C2, and the C4 batch)
        lat = {}
        for name, (c2, o2, v2, b2) in ((("c2_single_teb", scenes.scene_c2(stride=208)),
                                        ("c4_batch", scenes.scene_c4(B=B, n=n, stride=STRIDE))) if args.latency_reps > 0 else ()):
            s2 = planner.make_solver(c2, o2, v2, b2)
            ts = []
            for _ in range(args.latency_reps):
                hb = b2.copy()
                t1 = time.perf_counter()
                s2.upload(hb)
                s2.optimize(inner, outer, True, c2.hcp.selection_obst_cost_scale, c2.hcp.selection_viapoint_cost_scale,
                            c2.hcp.selection_alternative_time_cost)
                s2.select_best(-1, -1)
                s2.download(hb)
                ts.append(time.perf_counter() - t1)
            lat[name + "_p50_ms"] = 1e3 * float(np.median(ts))
            # the same tick with the bands resident in HBM: warm start on the device, optimise,
            # select, velocity command of the winner; only a start pose, a goal pose and two twists cross PCIe
            s2.upload(b2)
            s2.snapshot()
            x0, y0, th0, _ = b2.get_teb(0)
            new_start = [float(x0[0]), float(y0[0]), float(th0[0])]
            goal = [float(x0[-1]), float(y0[-1]), float(th0[-1])]
            ts = []
            for _ in range(args.latency_reps):
                s2.restore()
                t1 = time.perf_counter()
                s2.update_and_prune(new_start, goal, c2.trajectory.min_samples)
                s2.set_velocity_start([0.0, 0.0, 0.0])
                s2.optimize(inner, outer, True, c2.hcp.selection_obst_cost_scale, c2.hcp.selection_viapoint_cost_scale,
                            c2.hcp.selection_alternative_time_cost)
                best, _ = s2.select_best(-1, -1)
                s2.velocity_command(best, 1, 0)
                ts.append(time.perf_counter() - t1)
            lat[name + "_device_resident_p50_ms"] = 1e3 * float(np.median(ts))
            s2.close()
        if args.latency_reps > 0:
            # whole HomotopyClassPlanner::plan() tick on device-resident bands: updateAllTEBs, H-signatures +
            # class filter + detour deletion + compaction, roadmap graph (15 samples) with all-pairs collision tests, depth-first
            # candidate paths -> band init + signatures, optimizeAllTEBs (4x5, autosize), selectBestTeb, velocity command
            hc = scenes.scene_c4(B=1, n=n)[0]
            hc.hcp.max_number_classes = 5
            rng = np.random.default_rng(5)
            hob = _abi.ObstacleTable()
            for _ in range(12):
                hob.add_point(rng.uniform(1.5, 14.5), rng.uniform(-2.5, 2.5))
            ticks = max(8, args.latency_reps)
            starts = [[0.05 * k, 0.0, 0.0] for k in range(ticks)]
            goals = [[16.0, 0.0, 0.0]] * ticks
            hp = planner.HomotopyClassPlanner(hc, hob, [], None, max_tebs=8, max_poses=224)   # <= 238: normal matrix as blocks in LDS
            ts, nb = [], []
            for k in range(ticks):
                t1 = time.perf_counter()
                hp.plan(starts[k], goals[k], [0.3, 0.0, 0.0])
                hp.getVelocityCommand()
                ts.append(time.perf_counter() - t1)
                nb.append(hp.solver.count)
            hp.solver.close()
            lat["hcp_plan_tick_p50_ms"] = 1e3 * float(np.median(ts[1:]))
            lat["hcp_plan_tick"] = {"workload": "HomotopyClassPlanner::plan() ticks on one planner: 16 m straight task, 12 point obstacles, "
                                                "roadmap graph (15 samples), max_number_classes 5, 4x5 iterations, teb_autosize on, pose capacity 224",
                                    "ticks": ticks, "first_tick_ms": 1e3 * ts[0], "bands_per_tick": [int(min(nb)), int(max(nb))]}
            try:
                from oracle import ref_py
                if os.path.exists(ref_py.SO):
                    t1 = time.perf_counter()
                    ref_py.hcp_plan_ticks(hc, hob, starts, goals, [[0.3, 0.0, 0.0]] * ticks)
                    lat["hcp_plan_tick"]["reference_code_cpu_ms_per_tick"] = 1e3 * (time.perf_counter() - t1) / ticks   # oracle/_ref, one thread
            except Exception as e:
                lat["hcp_plan_tick"]["reference_code_cpu_ms_per_tick"] = str(e)[:120]
        out["plan_latency"] = lat
        # ---- secondary numbers: C4 with autoResize switched off (every band keeps exactly 200 poses, block-form normal matrix in
        #      LDS - the fastest configuration of the kernel) and BASELINE configs[2] (64 x 150 poses, 200 obstacles, defaults)
        if args.latency_reps > 0:
            c3, o3, v3, b3 = scenes.scene_c3(stride=208)
            s3 = planner.make_solver(c3, o3, v3, b3)
            s3.snapshot()
            ms3, t3 = [], []
            for _ in range(max(3, args.latency_reps // 4)):
                s3.restore()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                s3.optimize(inner, outer, True, c3.hcp.selection_obst_cost_scale, c3.hcp.selection_viapoint_cost_scale,
                            c3.hcp.selection_alternative_time_cost)
                r3 = s3.results()
                t3.append(time.perf_counter() - t1)
                ms3.append(s3.last_kernel_ms())
            n3 = s3.pose_counts()
            u3 = int(r3.lm_iterations.sum())
            c4f, o4f, v4f, b4f = scenes.scene_c4(B=B, n=n, stride=max(n, 208))
            c4f.trajectory.teb_autosize = False
            s4 = planner.make_solver(c4f, o4f, v4f, b4f)
            s4.snapshot()
            ms4, t4 = [], []
            for _ in range(max(3, args.latency_reps // 2)):
                s4.restore()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                s4.optimize(inner, outer, True, c4f.hcp.selection_obst_cost_scale, c4f.hcp.selection_viapoint_cost_scale,
                            c4f.hcp.selection_alternative_time_cost)
                r4 = s4.results()
                t4.append(time.perf_counter() - t1)
                ms4.append(s4.last_kernel_ms())
            u4 = int(r4.lm_iterations.sum())
            s4.close()
            out["secondary"] = {"c4_fixed_200_poses": {
                "workload": "C4 with teb_autosize off: every band keeps exactly %d poses; normal matrix as 8x8 blocks in LDS" % n,
                "kernel_ms": float(np.median(ms4)), "ms_per_step": 1e3 * float(np.median(t4)), "units_per_step": u4,
                "value": u4 / float(np.median(t4)), "unit": "TEB.LM-iterations/s", "tebs_ok": int((r4.status == 0).sum())},
                                "c3_autosize_on": {
                "workload": "C3: 64 candidate TEBs x 150 poses, 200 point obstacles, teb_autosize on, 4 outer x 5 inner",
                "kernel_ms": float(np.median(ms3)), "ms_per_step": 1e3 * float(np.median(t3)), "units_per_step": u3,
                "value": u3 / float(np.median(t3)), "unit": "TEB.LM-iterations/s",
                "poses_after": [int(n3.min()), int(n3.max())], "tebs_ok": int((r3.status == 0).sum())}}
            s3.close()
        # ---- CPU baseline: the oracle in the reference-faithful mode (g2o central differences), thread per TEB
        if not args.no_cpu_baseline:
            from oracle import oracle_py
            oracle_py.build()
            cores = os.cpu_count() or 1
            ks = min(args.cpu_sample, B)
            cb = _abi.TebBatchHost(ks, max(batch.stride, 320))
            for b in range(ks):
                cb.set_teb(b, *batch.get_teb(b))
            cb.has_vel_goal[:] = batch.has_vel_goal[:ks]
            cfg_cpu = scenes.scene_c4(B=1, n=n)[0]
            cfg_cpu.jacobian_mode = _abi.JACOBIAN_G2O_NUMERIC
            # thread-per-TEB like the reference's optimizeAllTEBs; the thread count that gives the CPU its best rate is searched
            # (a shared 256-thread host is not fastest with 256 busy threads) and reported as `cores`
            try:
                avail = len(os.sched_getaffinity(0))
            except Exception:
                avail = cores
            cand = sorted({t for t in (avail, avail // 2, avail // 4, 64, 32, 16) if 1 <= t <= max(avail, 1)}, reverse=True)
            cpu_t, cores = float("inf"), avail
            for th in cand:
                for _rep in range(2):
                    t1 = time.perf_counter()
                    _, cres = oracle_py.optimize_batch(cfg_cpu, obst, via, cb, threads=th)
                    dtc = time.perf_counter() - t1
                    if dtc < cpu_t:
                        cpu_t, cores = dtc, th
            out["cpu_baseline"] = {
                "value": float(cres.lm_iterations.sum()) / cpu_t, "unit": "TEB.LM-iterations/s", "cores": cores,
                "kind": "port",
                "sample": "%d of the %d C4 candidates, one optimizeTEB each (4x5, teb_autosize on), g2o-numeric Jacobians, "
                          "one std::thread per TEB capped at %d (best of the thread counts tried, 2 runs each), %.1f s wall; the port is bit-identical to the "
                          "reference's src/optimal_planner.cpp on the pinned bands (tests/test_reference_pinning.py)" % (ks, B, cores, cpu_t)}
            # the reference's OWN code on the same sample (oracle/_ref: src/optimal_planner.cpp + edge classes compiled in place; only
            # the LM iteration / banded Cholesky inside is a stand-in for the absent libg2o). Same results bit for bit; slower than the
            # port because of the g2o-style virtual edge interface. Reported beside the port, which stays the (faster) baseline value.
            try:
                from oracle import ref_py
                if os.path.exists(ref_py.SO):
                    rt = float("inf")
                    for _rep in range(2):
                        t1 = time.perf_counter()
                        _, rok, _, rit = ref_py.optimize_batch(cfg_cpu, obst, via, cb, threads=cores)
                        rt = min(rt, time.perf_counter() - t1)
                    out["cpu_baseline"]["reference_code"] = {
                        "value": float(rit.sum()) / rt, "unit": "TEB.LM-iterations/s", "cores": cores,
                        "sample": "same %d candidates through oracle/_ref/libteb_ref.so, best of 2 runs, %.1f s wall" % (ks, rt)}
            except Exception as e:   # the checker library is optional on the bench box
                out["cpu_baseline"]["reference_code"] = {"error": str(e)[:200]}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
