# round 5, job 35: the two streams' points interleaved (default now) against contiguous halves (GLH_TRACK_SPLIT=block)
mkdir -p gpurun_out
{
python tools/experiments/switch_probe.py C5 2048 60 GLH_TRACK_SPLIT=block
python tools/experiments/switch_probe.py C3 4096 60 GLH_TRACK_SPLIT=block
python tools/experiments/switch_probe.py C5 512 60 GLH_TRACK_SPLIT=block
python tools/experiments/switch_probe.py C4 1250 40 GLH_TRACK_SPLIT=block
python tools/experiments/switch_probe.py C2 512 60 GLH_TRACK_SPLIT=block
} > gpurun_out/r5j35_interleave.txt 2>&1
cat gpurun_out/r5j35_interleave.txt
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_pinned.py -x -q -m gpu 2>&1 | tail -3
