#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04g
mkdir -p $OUT
cd $R
for wl in c3d4 c3d10 c3d4_8m; do timeout 500 python tools/r04_ab.py footprint $wl 2>&1 | grep -v amdgpu.ids >> $OUT/ab_footprint.txt; done
cat $OUT/ab_footprint.txt
