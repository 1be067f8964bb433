# round 5, job 41: workgroups with large search tiles raise their wave priority (-DGLH_PT_PRIO=1, prio.so)
mkdir -p gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base.so.off   # (base == new here: two variants are enough)
{
echo "== C5 share (512 points)"; AB_ENVS="prio.so" bash tools/ab.sh --workload C5
echo "== C2"; AB_ENVS="prio.so" bash tools/ab.sh --workload C2
echo "== C5 (2048 points)"; AB_ENVS="prio.so" bash tools/ab.sh --workload C5 --points 2048
echo "== C3"; AB_ENVS="prio.so" bash tools/ab.sh --no-secondary
} > gpurun_out/r5j41_prio.txt 2>&1
mv glimpse_amd/lib/base.so.off glimpse_amd/lib/base.so
cat gpurun_out/r5j41_prio.txt
