#!/bin/bash
# HBM-side read traffic of the backward sweep of the single-rhs solve (trsv_bwd_dinv_kernel), rocprofv3 --pmc, C3 factor
set -o pipefail
OUT=$PWD/gpurun_out/r03_trsv
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-20)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "trsv_bwd" --output-format csv -d $OUT/pmc_$tag -o pmc -- python3 $R/tools/potrs_only.py c3 > $OUT/pmc_$tag.log 2>&1
  echo "pmc trsv [$grp] rc=$?"
done
python3 - <<PY
import csv, glob, collections
for tag in ('FETCH_SIZE', 'WRITE_SIZE', 'TCC_HIT_sum_TCC_MISS_'):
    f = glob.glob('$OUT/pmc_%s*/**/*counter_collection.csv' % tag, recursive=True)
    if not f:
        print(tag, 'no file'); continue
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        tot[(r['Kernel_Name'][:24], r['Counter_Name'])] += float(r['Counter_Value']); n[(r['Kernel_Name'][:24], r['Counter_Name'])] += 1
    for k in sorted(tot):
        print(k, 'sum %.4g over %d dispatches' % (tot[k], n[k]))
PY
