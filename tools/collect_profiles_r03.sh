#!/bin/bash
# Round-3 profile collection, run on the MI355X box through gpurun from the repo root: un-profiled bench lines of every workload
# (C3 with the CPU leg), the block-cyclic driver on a 1 x 1 grid beside the single-GPU path at the same N, kernel-trace summaries
# of C3 and C2, the loglikelihood table.  Everything lands under gpurun_out/r03/.
set -o pipefail
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/r03_c3_bench.json 2> $OUT/c3_bench.err; echo "c3 bench rc=$?"
python3 $R/bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r03_c2_bench.json 2> $OUT/c2_bench.err; echo "c2 bench rc=$?"
python3 $R/bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r03_c4_bench.json 2> $OUT/c4_bench.err; echo "c4 bench rc=$?"
python3 $R/bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/r03_c5_bench.json 2> $OUT/c5_bench.err; echo "c5 bench rc=$?"
MGPI_BENCH_FORCE_DIST=1 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r03_dist_1x1_n40960_bench.json 2> $OUT/dist.err; echo "dist 1x1 rc=$?"
MGPI_BENCH_SIDE=32x32x40 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r03_single_n40960_bench.json 2> $OUT/single.err; echo "single n40960 rc=$?"
MGPI_BENCH_FORCE_DIST=1 MGPI_BENCH_SIDE=16x16x16x16 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/r03_dist_1x1_c4_bench.json 2>> $OUT/dist.err; echo "dist 1x1 c4 rc=$?"
python3 $R/tools/loglike_timing.py > $OUT/r03_loglike_timing.txt 2>&1; echo "loglike rc=$?"
python3 $R/tools/run_configs.py > $OUT/r03_run_configs.txt 2>&1; echo "run_configs rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o r03_c3 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/r03_c3_bench_under_rocprof.json 2> $OUT/c3.err
echo "c3 trace rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -o r03_c2 -- python3 $R/bench.py --workload c2 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/r03_c2_bench_under_rocprof.json 2> $OUT/c2.err
echo "c2 trace rc=$?"
find $OUT -name "*kernel_stats.csv" | head
