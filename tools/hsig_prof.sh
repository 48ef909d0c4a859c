set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/hsig; mkdir -p $OUT; export TMPDIR=/tmp
python tools/hsig_bench.py
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/t -o t -- python $ROOT/tools/hsig_bench.py > /dev/null 2> $OUT/t.log
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace -d $OUT/f -o f -- python $ROOT/tools/hsig_bench.py > /dev/null 2> $OUT/f.log
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -d $OUT/s -o s -- python $ROOT/tools/hsig_bench.py > /dev/null 2> $OUT/s.log
cd $ROOT
python - <<'PY'
import sqlite3, os
out = os.path.join(os.getcwd(), "gpurun_out", "hsig")
con = sqlite3.connect(os.path.join(out, "t", "t_results.db"))
for r in con.execute("select name, count(*), avg(duration), min(duration), vgpr_count, scratch_size, lds_size from kernels where name like '%hsig%' group by name"): print(r)
for sub in ("f", "s"):
    con = sqlite3.connect(os.path.join(out, sub, sub + "_results.db"))
    for r in con.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%hsig3d%' group by counter_name"): print(r)
PY
