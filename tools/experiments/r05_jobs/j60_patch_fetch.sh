# round 5, job 60: the windows of both surfaces requested together ahead of the prologue's barrier (was: two loads in a row
# behind it): tests over rasters, A/B against HEAD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_pinned.py -x -q -m gpu -k "raster or gridded or surface or tangent or dem or viewshed or motion" 2>&1 | tail -3
{
echo "== C3 tangent_cartesian over a gridded DEM"; bash tools/ab.sh --no-secondary --motion tangent_cartesian --dem gridded
echo "== C3 cartesian over a gridded DEM"; bash tools/ab.sh --no-secondary --dem gridded
echo "== C5 over rasters"; bash tools/ab.sh --workload C5 --points 2048 --dem gridded
} > gpurun_out/r5j60_patch_fetch.txt 2>&1
cat gpurun_out/r5j60_patch_fetch.txt
