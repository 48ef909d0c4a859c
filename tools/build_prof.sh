#!/bin/bash
# Builds tools/libteb_amd_prof.so: the product kernel with clock64() phase counters (-DTEB_PROFILE), closed-form Jacobian
# instantiations only (-DTEB_AMD_ANALYTIC_ONLY, 35 s instead of 3 min). Used by tools/prof_phases.py on the GPU box.
set -e
cd "$(dirname "$0")/../teb_local_planner_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DTEB_PROFILE -DTEB_AMD_ANALYTIC_ONLY -DTEB_AMD_SINGLE_TU -ldl \
    teb_amd.hip -o ../../tools/libteb_amd_prof.so
echo built tools/libteb_amd_prof.so
