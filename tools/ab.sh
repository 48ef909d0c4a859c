#!/bin/bash
# usage (on the GPU box): tools/ab.sh [-r rounds] lib1.so lib2.so ...   - kernel times of the builds, alternately (ABAB..), same box, same call
R=2
if [ "$1" = "-r" ]; then R=$2; shift 2; fi
for i in $(seq $R); do for L in "$@"; do TEB_AMD_LIB=$PWD/$L python tools/kernel_times.py $CASES; done; done
