# round 5, job 15: ablation -- the window samples replaced by a constant expression (abl.so): what they cost in place
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base_r04.so
for cfg in "--motion tangent_cartesian --dem gridded" "--dem gridded" "--motion tangent_cartesian"; do
  echo "--- $cfg"
  AB_ENVS="abl.so" bash tools/ab.sh --no-secondary $cfg 2>/dev/null
done > gpurun_out/r5j15_ablate.txt 2>&1
cat gpurun_out/r5j15_ablate.txt
for lib in libglimpse_hip abl; do
  echo "=== $lib tangent gridded"
  GLH_LIB=$PWD/glimpse_amd/lib/$lib.so GLH_MOTION=tangent_cartesian GLH_DEM=gridded python tools/phase_probe.py C3 4096 5000 12 2>&1 | grep "A split\|A evolve\|E gather\|last step"
done
