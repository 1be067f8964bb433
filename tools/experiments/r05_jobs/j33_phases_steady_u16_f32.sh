# round 5, job 33: phase stamps in the steady state (frame 60) -- 8-bit general code, uint16, float32 frames
mkdir -p gpurun_out
{
GLH_MOTION=tangent_cartesian python tools/phase_probe.py C3 4096 5000 60
GLH_BITS=16 python tools/phase_probe.py C3 4096 5000 60
GLH_BITS=32 python tools/phase_probe.py C3 4096 5000 60
} 2>&1 | grep -v "^  slowest\|block start\|percentiles" > gpurun_out/r5j33_phases_steady.txt
cat gpurun_out/r5j33_phases_steady.txt
