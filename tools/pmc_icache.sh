set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_icache; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-check --latency-reps 0"
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -d $OUT/ic -o ic -- $CMD > /dev/null 2> $OUT/ic.log
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $OUT/if -o ifetch -- $CMD > /dev/null 2> $OUT/if.log
cd $ROOT
python - <<PY
import sqlite3, glob
for sub, db in (("ic", "ic_results.db"), ("if", "ifetch_results.db")):
    try:
        con = sqlite3.connect("gpurun_out/prof_icache/%s/%s" % (sub, db))
        for r in con.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%teb_optimize%' and grid_size_x=65536 group by counter_name"):
            print(r)
    except Exception as e:
        print(sub, "failed", e)
PY
tail -3 $OUT/ic.log | cut -c1-200
