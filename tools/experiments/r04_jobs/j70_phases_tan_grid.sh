GLH_MOTION=tangent_cartesian GLH_DEM=gridded timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j70_phase_tan_grid.txt 2>&1
GLH_MOTION=tangent_cartesian timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j70_phase_tan.txt 2>&1
grep -n "A split\|point_step\|median  " gpurun_out/r4j70_phase_tan_grid.txt gpurun_out/r4j70_phase_tan.txt
