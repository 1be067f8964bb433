GLH_BITS=32 timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j31_phase_b32.txt 2>&1
grep -n "split" gpurun_out/r4j31_phase_b32.txt
