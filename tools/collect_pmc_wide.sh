#!/bin/bash
# round 3: the 256 x 128 tile (MGPI_GEMM_WIDE=1) against the shipped 128 x 128 one, lower syrk at n = 32256, K = 1024:
# HBM-side traffic, L2 hit rate and MFMA-busy, each counter group in its own --pmc pass
set -o pipefail
OUT=$PWD/gpurun_out/r03_wide
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for wide in 0 1; do
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_LDS"; do
    tag=wide${wide}_$(echo $grp | tr ' ' '_' | cut -c1-24)
    MGPI_GEMM_WIDE=$wide timeout -k 10 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$tag -o pmc -- python3 $R/tools/syrk_only.py 32256 2 1024 > $OUT/pmc_$tag.log 2>&1
    echo "pmc wide=$wide [$grp] rc=$?"
  done
done
