set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcv; mkdir -p $OUT; export TMPDIR=/tmp
export TEB_AMD_LIB=$ROOT/tools/variants/libteb_amd_inl.so
CMD="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --latency-reps 0"
$CMD | tail -1 | cut -c1-200
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -o f -- $CMD > /dev/null 2> $OUT/f.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/w -o w -- $CMD > /dev/null 2> $OUT/w.log
cd $ROOT
python - <<'PY'
import sqlite3, os
out = os.path.join(os.getcwd(), "gpurun_out", "pmcv")
for sub, cn, f in (("f", "FETCH_SIZE", 2048), ("w", "WRITE_SIZE", 1024)):
    con = sqlite3.connect(os.path.join(out, sub, sub + "_results.db"))
    r = list(con.execute("select count(*), avg(value) from counters_collection where kernel_name like '%teb_optimize%' and counter_name='" + cn + "' and grid_size_x=65536"))
    print(cn, r, "GB/launch", r[0][1] * f / 1e9)
PY
