#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/icache; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQ_INSTS_|SQ_WAIT|SQ_BUSY|SQ_ACTIVE" | head -60 > $OUT/counters.txt
CMD="python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --latency-reps 0"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --kernel-trace -d $OUT/ic -o ic -- $CMD > /dev/null 2> $OUT/ic.log
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $OUT/sq2 -o sq2 -- $CMD > /dev/null 2> $OUT/sq2.log
cd $ROOT
python - <<'PY'
import sqlite3, glob, os
for db in glob.glob("gpurun_out/icache/*/*_results.db"):
    con = sqlite3.connect(db)
    try:
        for r in con.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%teb_optimize%' and grid_size_x=65536 group by counter_name"):
            print(db.split('/')[-1], r)
    except Exception as e:
        print(db, "ERR", e)
PY
cat $OUT/counters.txt | head -40
tail -3 $OUT/ic.log
