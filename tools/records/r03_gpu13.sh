export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03m; mkdir -p $OUT
cd $R
for l in libfemcy_hip libfemcy_hip_nt; do FEMCY_HIP_LIB=$R/femcy_amd/$l.so timeout 100 python tools/asm_probe.py c3d10 8 20 2>&1 | grep mode; done
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_nt.so timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/p_$ctr -o pmc -- python $R/tools/asm_probe.py c3d10 8 5 > $OUT/p_$ctr.log 2>&1
  python $R/tools/rocprof_summary.py pmc $(find $OUT/p_$ctr -name "*.db" | head -1) $ctr 2>&1 | grep -E "k_assemble|^kernel"
  rm -rf $OUT/p_$ctr
done
