GLH_CHANNELS=3 timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j65_phase_rgb.txt 2>&1
timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j65_phase_gray.txt 2>&1
grep -n "tile_prep split\|point_step\|B tile_prep" gpurun_out/r4j65_phase_rgb.txt gpurun_out/r4j65_phase_gray.txt
