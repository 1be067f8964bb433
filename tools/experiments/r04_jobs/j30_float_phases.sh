set -x
for b in 32 16; do GLH_BITS=$b timeout 300 python tools/phase_probe.py C3 4096 5000 12 > gpurun_out/r4j30_phase_b$b.txt 2>&1; done
tail -25 gpurun_out/r4j30_phase_b32.txt
