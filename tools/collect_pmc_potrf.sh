#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) and MFMA-busy of the trailing-update launches inside ONE N = 32768 factorisation:
# only the gemm_f64_pipe_kernel dispatches are profiled (--kernel-include-regex), the panel kernels beside them run as usual.
# Prints a heartbeat so that the long counter passes are not taken for a hang.  Output: gpurun_out/r02b/pmc_potrf_*/
set -o pipefail
OUT=$PWD/gpurun_out/r02b
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( while true; do sleep 45; echo "[heartbeat] $(date +%T)"; done ) &
HB=$!
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "gemm_f64_pipe_kernel" --output-format csv -d $OUT/pmc_potrf_$tag -o pmc -- python3 $R/tools/potrf_only.py 32x32x32 1 > $OUT/pmc_potrf_$tag.log 2>&1
  echo "pmc potrf [$grp] rc=$?"
done
kill $HB
find $OUT -name "*counter*.csv" | head
