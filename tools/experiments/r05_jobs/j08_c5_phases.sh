# round 5, job 8: phase stamps of C5 (two observers) and C3 for comparison
mkdir -p gpurun_out
{
python tools/phase_probe.py C5 2048 5000 12
python tools/phase_probe.py C3 4096 5000 12
} 2>&1 | grep -v "^  slowest\|block start" > gpurun_out/r5j08_phases.txt
cat gpurun_out/r5j08_phases.txt
