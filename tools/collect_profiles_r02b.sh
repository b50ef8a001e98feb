#!/bin/bash
# Round-2 (second half) profile collection, run on the MI355X box through gpurun from the repo root:
#   kernel-trace summary of the default bench (C3) with the merged trailing update, un-profiled bench lines of every
#   workload.  (Counters of the merged kernel: tools/collect_pmc_syrk.sh.)  Everything lands under gpurun_out/r02b/.
set -o pipefail
OUT=$PWD/gpurun_out/r02b
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/r02b_c3_bench.json 2> $OUT/c3_bench.err; echo "c3 bench rc=$?"
python3 $R/bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r02b_c2_bench.json 2> $OUT/c2_bench.err; echo "c2 bench rc=$?"
python3 $R/bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r02b_c4_bench.json 2> $OUT/c4_bench.err; echo "c4 bench rc=$?"
python3 $R/bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/r02b_c5_bench.json 2> $OUT/c5_bench.err; echo "c5 bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o r02b_c3 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/r02b_c3_bench_under_rocprof.json 2> $OUT/c3.err
echo "c3 trace rc=$?"
find $OUT -name "*.csv" | head -40
