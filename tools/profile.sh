#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command.
# Outputs land in gpurun_out/prof/ ; copy the summaries you want to keep into profiles/.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --latency-reps 0"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- $CMD > $OUT/bench_write.json 2> $OUT/write.log
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/pmc_sq -o sq -- $CMD > $OUT/bench_sq.json 2> $OUT/sq.log
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace -d $OUT/pmc_f64 -o f64 -- $CMD > $OUT/bench_f64.json 2> $OUT/f64.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cal_fetch -o calf -- python $ROOT/tools/calib_stream.py > /dev/null 2> $OUT/calf.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cal_write -o calw -- python $ROOT/tools/calib_stream.py > /dev/null 2> $OUT/calw.log
cd $ROOT
python - <<'PY'
import sqlite3, json, os
out = os.path.join(os.getcwd(), "gpurun_out", "prof")
res = {"true_bytes_read_per_launch": 1 << 30, "true_bytes_written_per_launch": 1 << 30, "pattern": "8 B/lane coalesced fp64 stream, 4 launches"}
for sub, db, cn in (("fetch", "cal_fetch/calf_results.db", "FETCH_SIZE"), ("write", "cal_write/calw_results.db", "WRITE_SIZE")):
    con = sqlite3.connect(os.path.join(out, db))
    r = list(con.execute("select count(*), avg(value) from counters_collection where kernel_name like '%stream_kernel%' and counter_name='" + cn + "'"))
    res[cn + "_KB_per_launch"] = r[0][1]
    res[cn + "_bytes_per_counted_KB"] = (1 << 30) / r[0][1] if r[0][1] else None
json.dump(res, open(os.path.join(out, "calib.json"), "w"), indent=1)
print("calibration", res)
PY
find $OUT -name "*.csv" | head -50
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -20 $f; done
python - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.getcwd(), "gpurun_out", "prof")
for tag in ("fetch", "write", "sq", "f64"):
    for f in glob.glob(os.path.join(out, "pmc_" + tag, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
                agg[k][0] += float(row.get("Counter_Value", 0)); agg[k][1] += 1
        print("==", f)
        for (kn, cn), (v, c) in sorted(agg.items()):
            print("  %-62s %-24s sum=%.6g n=%d mean=%.6g" % (kn, cn, v, c, v / max(c, 1)))
PY
