cd $GRAFT_REPO_ROOT
for ir in 4096 8192 16384; do MGPI_PANEL_INNER_ROWS=$ir python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inner_rows=$ir c3', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"; done
for mm in 6144 8192 12288; do MGPI_DIST_MERGED_MIN=$mm MGPI_BENCH_HOST_STAMPS=1 MGPI_BENCH_FORCE_DIST=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep host | tail -1 | sed "s/^/dist merged_min=$mm /"; done
for sm in 10240 14336 18432; do MGPI_DIST_SUPER_MIN=$sm MGPI_BENCH_HOST_STAMPS=1 MGPI_BENCH_FORCE_DIST=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep host | tail -1 | sed "s/^/dist super_min=$sm /"; done
