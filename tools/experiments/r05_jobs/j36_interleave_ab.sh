# round 5, job 36: base (HEAD: contiguous halves, no stride argument) / new in block mode / new interleaved; then the point
# orders again with "evens, odds" (the interleaved mixture with each half contiguous in memory)
mkdir -p gpurun_out
{
echo "== C5"; AB_ENVS="GLH_TRACK_SPLIT=block" bash tools/ab.sh --workload C5
echo "== C3"; AB_ENVS="GLH_TRACK_SPLIT=block" bash tools/ab.sh
GLH_TRACK_SPLIT=block python tools/experiments/lpt_order.py C5 2048 60
GLH_TRACK_SPLIT=block python tools/experiments/lpt_order.py C3 4096 60
} > gpurun_out/r5j36_interleave_ab.txt 2>&1
cat gpurun_out/r5j36_interleave_ab.txt
