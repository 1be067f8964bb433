# round 5, job 51: the two streams' points dealt in chunks of 64 (GLH_TRACK_SPLIT=chunks) against contiguous halves
mkdir -p gpurun_out
{
python tools/experiments/switch_probe.py C5 2048 60 GLH_TRACK_SPLIT=chunks
python tools/experiments/switch_probe.py C3 4096 60 GLH_TRACK_SPLIT=chunks
python tools/experiments/switch_probe.py C5 512 60 GLH_TRACK_SPLIT=chunks
python tools/experiments/switch_probe.py C4 1250 40 GLH_TRACK_SPLIT=chunks
python tools/experiments/switch_probe.py C2 512 60 GLH_TRACK_SPLIT=chunks
} > gpurun_out/r5j51_chunks.txt 2>&1
cat gpurun_out/r5j51_chunks.txt
