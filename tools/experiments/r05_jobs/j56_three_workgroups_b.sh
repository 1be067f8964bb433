# round 5, job 56: the same two builds at N = 2500 with and without the LDS bound: w6.so under the bound runs three
# workgroups per CU, without it two -- same code, same spills: the difference is what the third workgroup is worth
mkdir -p gpurun_out
mv glimpse_amd/lib/base.so glimpse_amd/lib/base.so.off
{
echo "== no bound (two workgroups per CU, both builds)"; AB_ENVS="w6.so" bash tools/ab.sh --no-secondary --particles 2500
export GLH_PT_LDS_HALF=45056
echo "== dynamic LDS <= 44 KB (new: two per CU by registers; w6: three)"; AB_ENVS="w6.so" bash tools/ab.sh --no-secondary --particles 2500
unset GLH_PT_LDS_HALF
echo "== N = 5000, no bound"; AB_ENVS="w6.so" bash tools/ab.sh --no-secondary
} > gpurun_out/r5j56_three_workgroups_b.txt 2>&1
mv glimpse_amd/lib/base.so.off glimpse_amd/lib/base.so
cat gpurun_out/r5j56_three_workgroups_b.txt
