# where a run over gridded surfaces spends its time
GLH_DEM=gridded timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j46_phase_cart_grid.txt 2>&1
GLH_MOTION=tangent_cartesian GLH_DEM=gridded timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j46_phase_tan_grid.txt 2>&1
GLH_HP=5 timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j46_phase_cart_general.txt 2>&1
grep -n "A split\|point_step\|median  " gpurun_out/r4j46_phase_*.txt
