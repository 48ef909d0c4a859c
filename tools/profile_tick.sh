#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace of the device-resident HomotopyClassPlanner ticks (tools/hcp_tick_profile.py)
# -> gpurun_out/prof_tick/ ; the per-kernel table is printed from the trace database.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_tick
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o tick -- python $ROOT/tools/hcp_tick_profile.py > $OUT/tick_stdout.txt 2> $OUT/tick.log
cd $ROOT
python - <<'PY'
import sqlite3, glob, os
dbs = glob.glob(os.path.join("gpurun_out", "prof_tick", "trace", "**", "*.db"), recursive=True)
con = sqlite3.connect(dbs[0])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(con.execute("select s.kernel_name, count(*), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), sum(d.end - d.start) "
                        "from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 6 desc" % (kd, ks)))
tot = sum(r[5] for r in rows)
print("%-64s %6s %10s %10s %10s %6s" % ("kernel", "calls", "avg us", "min us", "max us", "%"))
for name, n, avg, mn, mx, sm in rows:
    print("%-64s %6d %10.1f %10.1f %10.1f %6.1f" % (name.split("(")[0][:64], n, avg / 1e3, mn / 1e3, mx / 1e3, 100 * sm / tot))
PY
