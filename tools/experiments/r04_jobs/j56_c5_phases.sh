timeout 300 python tools/phase_probe.py C5 2048 5000 12 > gpurun_out/r4j56_phase_c5.txt 2>&1
cat gpurun_out/r4j56_phase_c5.txt | head -30
