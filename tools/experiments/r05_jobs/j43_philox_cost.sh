# round 5, job 43: what do Philox's rounds cost?  (ph5.so: 5 rounds instead of 7 -- timing only)
mkdir -p gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base.so.off
{
echo "== C3"; AB_ENVS="ph5.so" bash tools/ab.sh --no-secondary
echo "== C2"; AB_ENVS="ph5.so" bash tools/ab.sh --workload C2
} > gpurun_out/r5j43_philox_cost.txt 2>&1
mv glimpse_amd/lib/base.so.off glimpse_amd/lib/base.so
cat gpurun_out/r5j43_philox_cost.txt
