# round 5, job 1: the frame loop as one hipGraph (small batches), 1 024-thread workgroups on the wide first frames,
# and the whole of C4 on one GPU
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
{
python tools/experiments/switch_probe.py C2 256 50 GLH_TRACK_GRAPH=1
python tools/experiments/switch_probe.py C5 512 40 GLH_TRACK_GRAPH=1
python tools/experiments/switch_probe.py C3 4096 30 GLH_PT_BIG_FRAMES=6 GLH_PT_BIG_FRAMES=10
python tools/experiments/switch_probe.py C5 2048 30 GLH_PT_BIG_FRAMES=6
} > gpurun_out/r5j01_switches.txt 2>&1
python bench.py --workload C4 --split strong --gpus 1 --no-secondary --no-api --no-cpu-baseline > gpurun_out/r5j01_C4_full.json 2> gpurun_out/r5j01_C4_full.err
cat gpurun_out/r5j01_switches.txt; tail -c 3000 gpurun_out/r5j01_C4_full.json; tail -5 gpurun_out/r5j01_C4_full.err
