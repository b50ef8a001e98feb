cd $GRAFT_REPO_ROOT
for mm in 4096 6144 8192 12288; do
  echo "merged_min=$mm"
  for n in 8192 10240 12288 16384 20480; do MGPI_MERGED_MIN=$mm python3 tools/potrf_trace.py $n 4 | tail -1; done
  MGPI_MERGED_MIN=$mm python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
  MGPI_MERGED_MIN=$mm python bench.py --workload c2 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', round(d['ms_per_step'],2))"
done
