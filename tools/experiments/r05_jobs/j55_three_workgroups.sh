# round 5, job 55: would a THIRD resident workgroup per CU pay?  N = 3000 / 2500 particles (c[N] small enough for three
# workgroups' LDS), dynamic LDS bounded to 48 KB for both builds: `new` (128 VGPRs: two workgroups per CU) against w6.so
# (-DPT_MINW=6: 80 VGPRs, 192 B of scratch, three per CU); `base` = HEAD with its usual LDS plan
mkdir -p gpurun_out
export GLH_PT_LDS_HALF=49152
{
for n in 3000 2500; do
  echo "== C3 x $n particles, two streams"; AB_ENVS="w6.so" bash tools/ab.sh --no-secondary --particles $n
  echo "== C3 x $n particles, one stream"; AB_ENVS="w6.so" bash tools/ab.sh --no-secondary --particles $n --streams 1
done
} > gpurun_out/r5j55_three_workgroups.txt 2>&1
cat gpurun_out/r5j55_three_workgroups.txt
