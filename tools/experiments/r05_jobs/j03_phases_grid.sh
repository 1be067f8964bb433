# round 5, job 3: where a run over rasters spends its time (s_memtime phase stamps), four configurations
mkdir -p gpurun_out
for cfg in "cartesian constant" "cartesian gridded" "tangent_cartesian constant" "tangent_cartesian gridded"; do
  set -- $cfg
  echo "=== motion $1 dem $2"
  GLH_MOTION=$1 GLH_DEM=$2 python tools/phase_probe.py C3 4096 5000 12 2>&1 | grep -v "^  slowest\|block start\|percentiles"
done > gpurun_out/r5j03_phases_grid.txt 2>&1
cat gpurun_out/r5j03_phases_grid.txt
