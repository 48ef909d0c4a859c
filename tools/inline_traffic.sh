set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_inl; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-check --latency-reps 0"
cd /tmp
export TEB_AMD_LIB=$ROOT/tools/libteb_amd_inl.so
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- $CMD > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- $CMD > /dev/null 2> $OUT/write.log
cd $ROOT
python - <<PY
import sqlite3
for sub, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    con = sqlite3.connect("gpurun_out/prof_inl/pmc_%s/%s_results.db" % (sub, sub))
    print(cn, list(con.execute("select count(*), avg(value) from counters_collection where kernel_name like '%teb_optimize%' and counter_name='" + cn + "' and grid_size_x=65536")))
con = sqlite3.connect("gpurun_out/prof_inl/trace/trace_results.db")
print(list(con.execute("select count(*), avg(duration), scratch_size from kernels where name like '%teb_optimize%' and grid_x=65536")))
PY
