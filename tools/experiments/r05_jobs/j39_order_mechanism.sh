# round 5, job 39: why does the order of the points matter at C5?  each half alone; neighbours dealt in blocks; sorted, not dealt
mkdir -p gpurun_out
{
python tools/experiments/lpt_order.py C5 2048 60
python tools/experiments/lpt_order.py C3 4096 60
} > gpurun_out/r5j39_order_mechanism.txt 2>&1
cat gpurun_out/r5j39_order_mechanism.txt
