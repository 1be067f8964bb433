# round 5, job 48: phase stamps of the general code over constants and over rasters on the final build, steady state
mkdir -p gpurun_out
for cfg in "cartesian constant" "tangent_cartesian constant" "tangent_cartesian gridded" "cartesian gridded"; do
  set -- $cfg
  echo "=== motion $1 dem $2"
  GLH_MOTION=$1 GLH_DEM=$2 python tools/phase_probe.py C3 4096 5000 60 2>&1 | grep -v "^  slowest\|block start\|percentiles"
done > gpurun_out/r5j48_phases_grid_final.txt 2>&1
cat gpurun_out/r5j48_phases_grid_final.txt
