#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command (each counter group in its own run, with
# --kernel-trace only, as gpurun requires). Outputs land in gpurun_out/$PROF_DIR (default prof); tools/export_profile.py <tag> turns them
# into profiles/rocprof_<tag>_summary.{txt,json}. The last two passes profile the -DTEB_AMD_MFMA_SCHUR build (teb_local_planner_amd/libteb_amd_mfma.so,
# if present) for the matrix-core counters of the Schur update.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${PROF_DIR:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity-check --latency-reps 0 --sustain-seconds 0"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- $CMD > $OUT/bench_write.json 2> $OUT/write.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/pmc_sq -o sq -- $CMD > $OUT/bench_sq.json 2> $OUT/sq.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq2 -o sq2 -- $CMD > /dev/null 2> $OUT/sq2.log
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace -d $OUT/pmc_f64 -o f64 -- $CMD > $OUT/bench_f64.json 2> $OUT/f64.log
if [ -f $ROOT/teb_local_planner_amd/libteb_amd_mfma.so ]; then
  TEB_AMD_LIB=$ROOT/teb_local_planner_amd/libteb_amd_mfma.so rocprofv3 --kernel-trace --stats -d $OUT/mfma_trace -o mtrace -- $CMD > $OUT/bench_mfma_trace.json 2> $OUT/mfma_trace.log
  TEB_AMD_LIB=$ROOT/teb_local_planner_amd/libteb_amd_mfma.so rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_mfma -o mfma -- $CMD > /dev/null 2> $OUT/mfma.log
  TEB_AMD_LIB=$ROOT/teb_local_planner_amd/libteb_amd_mfma.so python $ROOT/tools/mfma_probe.py > $OUT/mfma_probe.txt 2>&1
fi
# the edge phases alone (diagnostic build without the damped solve, tools/build_prof.sh): their HBM traffic for roofline.edge_evaluation
if [ -f $ROOT/tools/libteb_amd_edge.so ]; then
  TEB_AMD_LIB=$ROOT/tools/libteb_amd_edge.so rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/edge_fetch -o efetch -- $CMD > $OUT/bench_edge_fetch.json 2> $OUT/edge_fetch.log
  TEB_AMD_LIB=$ROOT/tools/libteb_amd_edge.so rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/edge_write -o ewrite -- $CMD > $OUT/bench_edge_write.json 2> $OUT/edge_write.log
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cal_fetch -o calf -- python $ROOT/tools/calib_stream.py > /dev/null 2> $OUT/calf.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cal_write -o calw -- python $ROOT/tools/calib_stream.py > /dev/null 2> $OUT/calw.log
cd $ROOT
# phase split of the same binary's sources (-DTEB_PROFILE build, tools/build_prof.sh) and which bands the launch waits for
if [ -f $ROOT/tools/libteb_amd_prof.so ]; then
  : > $OUT/phases.txt
  for cfgname in c4on c3 c2 c5; do   # (one process per configuration)
    TEB_AMD_LIB=$ROOT/tools/libteb_amd_prof.so python $ROOT/tools/prof_phases.py $cfgname >> $OUT/phases.txt 2>&1
  done
  TEB_AMD_LIB=$ROOT/tools/libteb_amd_prof.so python $ROOT/tools/band_times.py c4on > $OUT/band_times.txt 2>&1
fi
python - <<PY
import sqlite3, json, os
out = "$OUT"
res = {"true_bytes_read_per_launch": 1 << 30, "true_bytes_written_per_launch": 1 << 30, "pattern": "8 B/lane coalesced fp64 stream, 4 launches"}
for sub, db, cn in (("fetch", "cal_fetch/calf_results.db", "FETCH_SIZE"), ("write", "cal_write/calw_results.db", "WRITE_SIZE")):
    con = sqlite3.connect(os.path.join(out, db))
    r = list(con.execute("select count(*), avg(value) from counters_collection where kernel_name like '%stream_kernel%' and counter_name='" + cn + "'"))
    res[cn + "_KB_per_launch"] = r[0][1]
    res[cn + "_bytes_per_counted_KB"] = (1 << 30) / r[0][1] if r[0][1] else None
json.dump(res, open(os.path.join(out, "calib.json"), "w"), indent=1)
print("calibration", res)
PY
tail -2 $OUT/bench_trace.json | cut -c1-600
