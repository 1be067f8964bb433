# dynamic instruction counts, phase by phase, of a run over gridded surfaces (cartesian model + DEM term)
bash tools/phase_counts.sh --dem gridded > gpurun_out/r4j48_counts_cart_grid.txt 2>&1
tail -30 gpurun_out/r4j48_counts_cart_grid.txt
