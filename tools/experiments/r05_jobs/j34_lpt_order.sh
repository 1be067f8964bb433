# round 5, job 34: does the order of the points in a launch matter (slowest first)?
mkdir -p gpurun_out
{
python tools/experiments/lpt_order.py C5 2048 60
python tools/experiments/lpt_order.py C3 4096 60
python tools/experiments/lpt_order.py C5 512 60
} > gpurun_out/r5j34_lpt_order.txt 2>&1
cat gpurun_out/r5j34_lpt_order.txt
