#!/bin/bash
# GPU call 1 of round 2: phase profile + wait attribution counters of the headline bench
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02a; mkdir -p $OUT; export TMPDIR=/tmp
python tools/prof_phases.py c4on c4 c2 c3 > $OUT/phases.txt 2>&1
CMD="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --latency-reps 0"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/sqa -o sqa -- $CMD > $OUT/bench_sqa.json 2> $OUT/sqa.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH --kernel-trace -d $OUT/sqb -o sqb -- $CMD > $OUT/bench_sqb.json 2> $OUT/sqb.log
cd $ROOT
python - <<'PY'
import sqlite3, glob
for db in sorted(glob.glob("gpurun_out/r02a/sq*/**/*_results.db", recursive=True)):
    con = sqlite3.connect(db)
    try:
        for r in con.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%teb_optimize%' and grid_size_x=65536 group by counter_name"):
            print(db.split('/')[-1], r)
    except Exception as e:
        print(db, "ERR", e)
PY
