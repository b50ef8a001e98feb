cd $GRAFT_REPO_ROOT
for inner in 128 256 512; do for sr in 4096 8192 16384; do
  MGPI_PANEL_INNER=$inner MGPI_PANEL_STEP_ROWS=$sr python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r5b_c3_${inner}_${sr}.json 2>&1
done; done
