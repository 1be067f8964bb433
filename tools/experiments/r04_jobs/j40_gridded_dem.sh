# gridded surfaces (DEM + DEM uncertainty rasters) under the tangent and the Cartesian model, against the constants
export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
for cfg in "--motion tangent_cartesian" "--motion tangent_cartesian --dem gridded" "--dem gridded" "--workload C5 --points 2048 --dem gridded"; do
  echo "--- $cfg"
  python bench.py --no-cpu-baseline --no-api --no-secondary $cfg 2> gpurun_out/r4j40_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],4), 'frac', round(r['frac'],3), d['health'], d['config'].get('dem'))" || tail -5 gpurun_out/r4j40_err.txt
done > gpurun_out/r4j40_gridded.txt 2>&1
cat gpurun_out/r4j40_gridded.txt
