# round 5, job 37: which phases make the slow points of C5's per-GPU share (and of C5, C2 on the final build) slow
mkdir -p gpurun_out
{
python tools/experiments/slow_points.py C5 512 5000 40
python tools/experiments/slow_points.py C5 2048 5000 40
python tools/experiments/slow_points.py C2 256 2000 40
} > gpurun_out/r5j37_slow_points.txt 2>&1
cat gpurun_out/r5j37_slow_points.txt
