cd $GRAFT_REPO_ROOT
for tr in 1024 2048 3072 4096; do
  echo "tail_rows=$tr"
  for n in 4096 8192 12288; do MGPI_TAIL_ROWS=$tr python3 tools/potrf_trace.py $n 4 | tail -1; done
  MGPI_TAIL_ROWS=$tr python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
  MGPI_TAIL_ROWS=$tr python bench.py --workload c2 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', round(d['ms_per_step'],2))"
done
