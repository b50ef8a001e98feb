#!/bin/bash
# HBM-side traffic and MFMA-busy of the MERGED trailing-update kernel alone (MGPI_SYRK_MERGED_CW), separate --pmc passes
set -o pipefail
OUT=$PWD/gpurun_out/r02b
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "16384 1024 8" "16384 512 4" "32256 1024 8"; do
  set -- $cfg
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=n$1_k$2_$(echo $grp | tr ' ' '_' | cut -c1-24)
    MGPI_SYRK_MERGED_CW=$3 timeout -k 10 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_msyrk_$tag -o pmc -- python3 $R/tools/syrk_only.py $1 2 $2 > $OUT/pmc_msyrk_$tag.log 2>&1
    echo "pmc merged syrk $cfg [$grp] rc=$?"
  done
done
# the factorisation itself under counter collection falls back to the separate column update (and must not hang)
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "gemm_f64_pipe_kernel" --output-format csv -d $OUT/pmc_potrf_auto -o pmc -- python3 $R/tools/potrf_only.py 16x16x16 1 > $OUT/pmc_potrf_auto.log 2>&1
echo "potrf under --pmc rc=$?"
