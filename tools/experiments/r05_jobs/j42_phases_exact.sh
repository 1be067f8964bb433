# round 5, job 42: phase stamps of exact arithmetic (the parity path) beside fast, steady state
mkdir -p gpurun_out
{
GLH_MATH=exact python tools/phase_probe.py C3 4096 5000 60
GLH_MATH=fast python tools/phase_probe.py C3 4096 5000 60
} 2>&1 | grep -v "^  slowest\|block start\|percentiles" > gpurun_out/r5j42_phases_exact.txt
cat gpurun_out/r5j42_phases_exact.txt
