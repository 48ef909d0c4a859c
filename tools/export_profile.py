#!/usr/bin/env python3
"""Turns the rocprofv3 rocpd sqlite outputs under gpurun_out/prof/ into the text summaries kept in profiles/.

usage: python tools/export_profile.py <round-tag>      (e.g. r01)
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", os.environ.get("PROF_DIR", "prof"))
DST = os.path.join(ROOT, "profiles")


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return list(con.execute(sql))
    finally:
        con.close()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(DST, exist_ok=True)
    lines = []
    tr = os.path.join(SRC, "trace", "trace_results.db")
    lines.append("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --latency-reps 0")
    lines.append("# (MI355X, gfx950; durations in microseconds; the single 1-workgroup debug launch of bench.py's association probe is listed separately)")
    lines.append("%-70s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in q(tr, "select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append("%-70s %8d %14.3f %12.3f %8.3f" % (name[:70], calls, total, avg, pct))
    rows = q(tr, "select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count, "
                 "count(*), avg(duration), min(duration), max(duration) from kernels where name like '%teb_optimize%' "
                 "group by name, grid_x")
    lines.append("")
    lines.append("# teb_optimize_kernel dispatches by grid size (duration ns)")
    full = None
    for r in rows:
        lines.append("grid=%d wg=%d lds=%d scratch=%d vgpr=%d agpr=%d sgpr=%d calls=%d avg_ns=%.0f min_ns=%.0f max_ns=%.0f" % r[1:])
        if full is None or r[1] > full[1]:
            full = r
    sys.path.insert(0, ROOT)
    import bench
    try:   # the hash the loaded BINARY carries (teb_amd_debug_build_info): bench.py attaches this summary only to the same binary
        from teb_local_planner_amd import planner as _pl
        binary_hash = _pl.TebBatchSolver.build_info()[0]
    except Exception as e:   # noqa: BLE001
        binary_hash = "unreadable: %s" % str(e)[:60]
    summary = {"tag": tag, "source_hash": bench.kernel_source_hash(), "binary_hash": binary_hash, "kernel": "teb_optimize_kernel", "grid": full[1], "workgroup": full[2], "calls": full[8],
               "avg_ms": full[9] / 1e6, "lds_bytes": full[3], "scratch_bytes_per_lane": full[4], "vgpr": full[5], "agpr": full[6]}
    grid = full[1]
    for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        db = os.path.join(SRC, "pmc_" + sub, sub + "_results.db")
        r = q(db, "select count(*), avg(value) from counters_collection where kernel_name like '%%teb_optimize%%' "
                  "and counter_name='%s' and grid_size_x=%d" % (cname, grid))
        lines.append("")
        lines.append("# rocprofv3 --pmc %s --kernel-trace (separate pass): mean per %d-workgroup launch over %d launches" % (cname, grid // full[2], r[0][0]))
        lines.append("%s_KB_per_launch %.3f" % (cname, r[0][1]))
        summary[cname + "_KB_per_launch"] = r[0][1]
    db = os.path.join(SRC, "pmc_sq", "sq_results.db")
    if os.path.exists(db):
        lines.append("")
        lines.append("# rocprofv3 --pmc SQ_* --kernel-trace (separate pass): mean per full launch")
        for cn, cnt, avg in q(db, "select counter_name, count(*), avg(value) from counters_collection where kernel_name like "
                                  "'%%teb_optimize%%' and grid_size_x=%d group by counter_name" % grid):
            lines.append("%-24s %.6g" % (cn, avg))
            summary[cn] = avg
    db = os.path.join(SRC, "pmc_sq2", "sq2_results.db")
    if os.path.exists(db):
        for cn, cnt, avg in q(db, "select counter_name, count(*), avg(value) from counters_collection where kernel_name like "
                                  "'%%teb_optimize%%' and grid_size_x=%d group by counter_name" % grid):
            lines.append("%-24s %.6g" % (cn, avg))
            summary[cn] = avg
    if "SQ_WAVE_CYCLES" in summary:
        wc = summary["SQ_WAVE_CYCLES"]
        att = {k: summary[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS") if k in summary}
        summary["wave_cycle_shares"] = att
        lines.append("")
        lines.append("# shares of SQ_WAVE_CYCLES (WAIT_ANY = parked at s_waitcnt / s_barrier, WAIT_INST_ANY = issue stalls, ACTIVE_INST_ANY = issuing)")
        lines.append("  ".join("%s %.1f %%" % (k, 100 * v) for k, v in att.items()))
    # matrix-core counters of the -DTEB_AMD_MFMA_SCHUR build (profiled with TEB_AMD_LIB=teb_local_planner_amd/libteb_amd_mfma.so)
    db = os.path.join(SRC, "pmc_mfma", "mfma_results.db")
    mt = os.path.join(SRC, "mfma_trace", "mtrace_results.db")
    if os.path.exists(db) and os.path.exists(mt):
        mf = {cn: avg for cn, cnt, avg in q(db, "select counter_name, count(*), avg(value) from counters_collection where kernel_name like "
                                                "'%%teb_optimize%%' and grid_size_x=%d group by counter_name" % grid)}
        r = q(mt, "select avg(duration) from kernels where name like '%%teb_optimize%%' and grid_x=%d" % grid)
        mfma_ms = r[0][0] / 1e6
        lines.append("")
        lines.append("# -DTEB_AMD_MFMA_SCHUR build (Schur update of the cyclic reduction on v_mfma_f64_16x16x4_f64), same command: mean per full launch")
        for k, v in sorted(mf.items()):
            lines.append("%-28s %.6g" % (k, v))
        lines.append("kernel avg %.3f ms (vector build: %.3f ms)" % (mfma_ms, summary["avg_ms"]))
        n_mfma = mf.get("SQ_INSTS_MFMA", 0.0)
        flop = n_mfma * 2048.0                         # 16 x 16 x 4 x 2 per instruction
        busy = mf.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        summary["mfma"] = {"build": "-DTEB_AMD_MFMA_SCHUR (off by default: measured slower)", "instruction": "v_mfma_f64_16x16x4_f64",
                           "kernel_ms": mfma_ms, "kernel_ms_vector_build": summary["avg_ms"],
                           "insts_mfma_per_launch": n_mfma, "mops_f64_per_launch": mf.get("SQ_INSTS_VALU_MFMA_MOPS_F64"),
                           "mfma_busy_cycles_per_launch": busy, "flop_per_launch": flop,
                           "achieved": flop / (mfma_ms * 1e-3) / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                           "frac": flop / (mfma_ms * 1e-3) / 1e12 / 78.6,
                           "busy_share_of_cu_time": busy / (4.0 * 256 * mfma_ms * 1e-3 * 2.4e9) if busy else None,
                           "note": "fp64 matrix rate on gfx950 = fp64 vector rate (64 cycles per 16x16x4 issue, measured: tools/mfma_probe.py); "
                                   "the update is 2 issues per elimination with 75 % useful MACs; the matrix build is slower end to end, so "
                                   "the product keeps the vector code"}
        lines.append("mfma: %.3g FLOP per launch on the matrix cores = %.3f TFLOP/s = %.2f %% of 78.6" % (flop, summary["mfma"]["achieved"], 100 * summary["mfma"]["frac"]))
        pr = os.path.join(SRC, "mfma_probe.txt")
        if os.path.exists(pr):
            lines.append(open(pr).read().strip())
    db = os.path.join(SRC, "pmc_f64", "f64_results.db")
    if os.path.exists(db):
        lines.append("")
        lines.append("# rocprofv3 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 --kernel-trace (separate pass): wave-level instruction counts,")
        lines.append("# mean per full launch; fp64 FLOP/launch = (ADD + MUL + TRANS + 2 FMA) x 64 lanes (upper bound: ignores the exec mask)")
        f64 = {}
        for cn, cnt, avg in q(db, "select counter_name, count(*), avg(value) from counters_collection where kernel_name like "
                                  "'%%teb_optimize%%' and grid_size_x=%d group by counter_name" % grid):
            lines.append("%-28s %.6g" % (cn, avg))
            summary[cn] = avg
            f64[cn] = avg
        if len(f64) == 4:
            flop = 64.0 * (f64["SQ_INSTS_VALU_ADD_F64"] + f64["SQ_INSTS_VALU_MUL_F64"] + f64["SQ_INSTS_VALU_TRANS_F64"]
                           + 2.0 * f64["SQ_INSTS_VALU_FMA_F64"])
            summary["fp64_flop_per_launch"] = flop
            lines.append("fp64_flop_per_launch %.6g  (%.3f TFLOP/s at the traced average duration)" % (flop, flop / (summary["avg_ms"] * 1e-3) / 1e12))
    # the edge phases alone: FETCH / WRITE of the -DTEB_AMD_DIAG_EDGE_ONLY build (tools/libteb_amd_edge.so) running the same command, with
    # the units (LM iterations) and trials of THAT run (its LM loop takes Jacobi steps instead of solves, so its counts are its own)
    ef, ew = os.path.join(SRC, "edge_fetch", "efetch_results.db"), os.path.join(SRC, "edge_write", "ewrite_results.db")
    if os.path.exists(ef) and os.path.exists(ew):
        eo = {}
        for db, cname in ((ef, "FETCH_SIZE"), (ew, "WRITE_SIZE")):
            r = q(db, "select count(*), avg(value) from counters_collection where kernel_name like '%%teb_optimize%%' "
                      "and counter_name='%s' and grid_size_x=%d" % (cname, grid))
            eo[cname + "_KB_per_launch"] = r[0][1]
        try:
            bj = json.loads(open(os.path.join(SRC, "bench_edge_fetch.json")).read().strip().splitlines()[-1])
            eo["units_per_launch"] = bj["config"]["units_per_step_per_gpu"]
            eo["lm_trials_per_launch"] = bj["config"]["lm_trials_per_step_per_gpu"]
            eo["kernel_ms"] = bj["roofline"]["kernel_ms"]
        except Exception as e:   # noqa: BLE001
            eo["error"] = str(e)[:120]
        summary["edge_only"] = eo
        lines.append("")
        lines.append("# the edge phases alone: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of the -DTEB_AMD_DIAG_EDGE_ONLY build (no damped solve, no band copy), same command")
        lines.append(json.dumps(eo))
    cal = os.path.join(SRC, "calib.json")
    if os.path.exists(cal):
        summary["calibration"] = json.load(open(cal))
        lines.append("")
        lines.append("# calibration of FETCH_SIZE / WRITE_SIZE on a known byte count with the kernel's 8 B/lane coalesced pattern")
        lines.append(json.dumps(summary["calibration"]))
    open(os.path.join(DST, "rocprof_%s_summary.txt" % tag), "w").write("\n".join(lines) + "\n")
    json.dump(summary, open(os.path.join(DST, "rocprof_%s_summary.json" % tag), "w"), indent=1)
    # the phase split and the per-band times of the SAME sources (-DTEB_PROFILE build run by tools/profile.sh), stamped with the hash this
    # script computes itself - one profile per round, nothing stamped by hand
    for src_name, dst_name, what in (("phases.txt", "phases_%s.txt" % tag, "tools/prof_phases.py c4on c3 c2 c5 (workgroup 0, clock64 section counters)"),
                                     ("band_times.txt", "band_times_%s.txt" % tag, "tools/band_times.py c4on (which bands the launch waits for)")):
        p = os.path.join(SRC, src_name)
        if os.path.exists(p):
            open(os.path.join(DST, dst_name), "w").write("# %s\n# source_hash %s (device + host sources of the library, bench.kernel_source_hash)\n%s" % (
                what, summary["source_hash"], open(p).read()))
    for name in ("bench_trace.json", "bench_fetch.json", "bench_write.json"):
        p = os.path.join(SRC, name)
        if os.path.exists(p):
            open(os.path.join(DST, "%s_%s" % (tag, name)), "w").write(open(p).read())
    print("\n".join(lines))


if __name__ == "__main__":
    main()
