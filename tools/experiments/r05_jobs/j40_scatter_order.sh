# round 5, job 40: the dispatch order of a launch scattered over its points (default now) against GLH_TRACK_ORDER=linear
mkdir -p gpurun_out
{
python tools/experiments/switch_probe.py C5 2048 60 GLH_TRACK_ORDER=linear
python tools/experiments/switch_probe.py C3 4096 60 GLH_TRACK_ORDER=linear
python tools/experiments/switch_probe.py C5 512 60 GLH_TRACK_ORDER=linear
python tools/experiments/switch_probe.py C4 1250 40 GLH_TRACK_ORDER=linear
python tools/experiments/switch_probe.py C2 256 60 GLH_TRACK_ORDER=linear
python tools/experiments/switch_probe.py C2 512 60 GLH_TRACK_ORDER=linear
echo "== bench C3: base (HEAD) / new / new linear"; AB_ENVS="GLH_TRACK_ORDER=linear" bash tools/ab.sh
} > gpurun_out/r5j40_scatter_order.txt 2>&1
cat gpurun_out/r5j40_scatter_order.txt
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_pinned.py -x -q -m gpu 2>&1 | tail -3
