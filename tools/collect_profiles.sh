#!/bin/bash
# Round-2 profile collection (run on the MI355X box through gpurun, from the repo root):
#   kernel-trace summary of the default bench (C3) and of the C5 / C2 workloads, then PMC passes - each counter group in
#   its own rocprofv3 run, FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots) - for
#   the trailing-update kernel, the kernel-matrix build and the fused predict-mean kernel.
# Everything lands under gpurun_out/r02/; the summaries worth keeping are copied to profiles/ afterwards.
set -o pipefail
OUT=$PWD/gpurun_out/r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o r02_c3 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/r02_c3_bench_under_rocprof.json 2> $OUT/c3.err
echo "c3 trace rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -o r02_c5 -- python3 $R/bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/r02_c5_bench_under_rocprof.json 2> $OUT/c5.err
echo "c5 trace rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -o r02_c2 -- python3 $R/bench.py --workload c2 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/r02_c2_bench_under_rocprof.json 2> $OUT/c2.err
echo "c2 trace rc=$?"
for grp in "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_valu_$tag -o pmc -- python3 $R/tools/predict_probe.py > $OUT/pmc_valu_$tag.log 2>&1
  echo "pmc predict_probe [$grp] rc=$?"
done
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_syrk_$tag -o pmc -- python3 $R/tools/syrk_only.py 16384 2 1024 > $OUT/pmc_syrk_$tag.log 2>&1
  echo "pmc syrk [$grp] rc=$?"
done
find $OUT -name "*.csv" | head -40
