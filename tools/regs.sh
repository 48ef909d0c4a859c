#!/bin/bash
# Resource usage (VGPR / AGPR / scratch / spills) of teb_optimize_kernel and its out-of-line callees for a set of -D flags.
# usage: tools/regs.sh [-DFLAG ...]   (cross-compiles for gfx950; no GPU needed)
cd "$(dirname "$0")/../teb_local_planner_amd/csrc"
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DTEB_AMD_ANALYTIC_ONLY -ldl "$@" \
    -save-temps=obj teb_amd.hip -o $T/t.so 2> $T/err.log || { tail -20 $T/err.log; exit 1; }
S=$T/teb_amd-hip-amdgcn-amd-amdhsa-gfx950.s
python3 - "$S" <<'PY'
import re, sys
name = None
info = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(_ZN\S+):", line)
    if m: name = m.group(1); continue
    m = re.match(r"^; (codeLenInByte|NumVgprs|NumAgprs|ScratchSize|Occupancy|SGPRSpill|VGPRSpill|NumSgprs)\s*[:=]\s*(\S+)", line)
    if m and name: info.setdefault(name, {})[m.group(1)] = m.group(2)
    m = re.match(r"^\s+\.(sgpr_spill_count|vgpr_spill_count):\s+(\d+)", line)
for k, v in info.items():
    if any(s in k for s in ("teb_optimize_kernel", "cr_solve", "autoresize_sweep", "lin_", "solve")):
        short = re.sub(r"E?v?14teb_amd_config.*", "", k)[:60]
        print("%-62s %s" % (short, " ".join("%s=%s" % kv for kv in v.items())))
PY
grep -E "^\s+\.(name|sgpr_spill_count|vgpr_spill_count):" $S | grep -A2 teb_optimize | grep -v "^--" | paste - - - | sed 's/ENS_8SceneDev.*LdsPlanE//' 
rm -rf $T
