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
    summary = {"tag": tag, "source_hash": bench.kernel_source_hash(), "kernel": "teb_optimize_kernel", "grid": full[1], "workgroup": full[2], "calls": full[8],
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
    cal = os.path.join(SRC, "calib.json")
    if os.path.exists(cal):
        summary["calibration"] = json.load(open(cal))
        lines.append("")
        lines.append("# calibration of FETCH_SIZE / WRITE_SIZE on a known byte count with the kernel's 8 B/lane coalesced pattern")
        lines.append(json.dumps(summary["calibration"]))
    open(os.path.join(DST, "rocprof_%s_summary.txt" % tag), "w").write("\n".join(lines) + "\n")
    json.dump(summary, open(os.path.join(DST, "rocprof_%s_summary.json" % tag), "w"), indent=1)
    for name in ("bench_trace.json", "bench_fetch.json", "bench_write.json"):
        p = os.path.join(SRC, name)
        if os.path.exists(p):
            open(os.path.join(DST, "%s_%s" % (tag, name)), "w").write(open(p).read())
    print("\n".join(lines))


if __name__ == "__main__":
    main()
