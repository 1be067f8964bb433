# round 5, job 19: phase stamps on uint16 and float32 frames
mkdir -p gpurun_out
{
GLH_BITS=16 python tools/phase_probe.py C3 4096 5000 10
GLH_BITS=32 python tools/phase_probe.py C3 4096 5000 10
} 2>&1 | grep -v "^  slowest\|block start\|percentiles" > gpurun_out/r5j19_phases_u16_f32.txt
cat gpurun_out/r5j19_phases_u16_f32.txt
