bash tools/phase_counts.sh --motion tangent_cartesian > gpurun_out/r4j58_counts_tangent.txt 2>&1
tail -22 gpurun_out/r4j58_counts_tangent.txt
