# where the general instantiation (TangentCartesianMotion) spends its 18 % over the common one
GLH_MOTION=tangent_cartesian timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j38_phase_tangent.txt 2>&1
timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j38_phase_common.txt 2>&1
grep -n "split\|point_step\|median  " gpurun_out/r4j38_phase_tangent.txt gpurun_out/r4j38_phase_common.txt
