#!/bin/bash
# Builds tools/libteb_amd_prof.so: the product kernel with clock64() phase counters (-DTEB_PROFILE), closed-form Jacobian
# instantiations only (-DTEB_AMD_ANALYTIC_ONLY, 35 s instead of 3 min). Used by tools/prof_phases.py on the GPU box.
set -e
cd "$(dirname "$0")/../teb_local_planner_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DTEB_PROFILE -DTEB_AMD_ANALYTIC_ONLY -DTEB_AMD_SINGLE_TU -ldl \
    teb_amd.hip -o ../../tools/libteb_amd_prof.so
echo built tools/libteb_amd_prof.so
# tools/libteb_amd_edge.so: the headline's kernel with the damped solve left out (-DTEB_AMD_DIAG_EDGE_ONLY, csrc/teb_kernel.hpp): tools/profile.sh
# takes the FETCH_SIZE / WRITE_SIZE of the edge phases alone from it (roofline.edge_evaluation.traffic of the bench line)
cd ../..
python -c "
from teb_local_planner_amd import build as b
print('built', b.build(variant='exp', extra_defines=['-DTEB_AMD_DIAG_EDGE_ONLY'], out='tools/libteb_amd_edge.so', only=('opt_0_0_4.o', 'opt_0_0_0.o')))"
