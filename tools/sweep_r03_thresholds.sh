# threshold sweep behind the round-3 defaults of MGPI_PANEL_STEP_ROWS and MGPI_GEMM_SKINNY_TILES (run on the GPU box)
cd $GRAFT_REPO_ROOT
run() { # workload steps tag envs...
  w=$1; s=$2; tag=$3; shift 3
  env "$@" python bench.py --workload $w --steps $s --warmup 2 --no-cpu-baseline > gpurun_out/r4f_${w}_${tag}.json 2>&1
}
for sr in 8192 16384 32768; do
  for sk in 448 900 1300 2000; do
    run c2 20 ${sr}_${sk} MGPI_PANEL_STEP_ROWS=$sr MGPI_GEMM_SKINNY_TILES=$sk
    run c4 6 ${sr}_${sk} MGPI_PANEL_STEP_ROWS=$sr MGPI_GEMM_SKINNY_TILES=$sk
    run c3 6 ${sr}_${sk} MGPI_PANEL_STEP_ROWS=$sr MGPI_GEMM_SKINNY_TILES=$sk
  done
done
