# round 5, job 10: tangent models over rasters: the gather samples the surface again (zp0.so) instead of reading the parked height
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
for cfg in "--motion tangent_cartesian --dem gridded" "--motion tangent_cylindrical --dem gridded"; do
  echo "--- $cfg"
  AB_ENVS="zp0.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null | grep -v "^base"
done > gpurun_out/r5j10_ab_zpark.txt 2>&1
cat gpurun_out/r5j10_ab_zpark.txt
GLH_LIB=$PWD/glimpse_amd/lib/zp0.so timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "raster or tangent" 2>&1 | tail -3
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r5j10_api.json 2> gpurun_out/r5j10_api.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5j10_api.json').read().splitlines()[0])
print({k:v for k,v in d['api_parallel_2'].items() if k!='note'})
PY
grep -i "traceback\|KeyError" gpurun_out/r5j10_api.err | head
