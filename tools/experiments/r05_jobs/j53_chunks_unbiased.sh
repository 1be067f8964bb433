# round 5, job 53: the probe without its order bias (every measurement is the second of two back-to-back runs): a no-op switch,
# the chunked split of the two streams
mkdir -p gpurun_out
{
python tools/experiments/switch_probe.py C3 4096 60 GLH_NOOP=1 GLH_TRACK_SPLIT=chunks GLH_NOOP2=1
python tools/experiments/switch_probe.py C5 2048 60 GLH_NOOP=1 GLH_TRACK_SPLIT=chunks GLH_NOOP2=1
python tools/experiments/switch_probe.py C5 512 60 GLH_TRACK_SPLIT=chunks GLH_NOOP=1
python tools/experiments/switch_probe.py C4 1250 40 GLH_TRACK_SPLIT=chunks GLH_NOOP=1
} > gpurun_out/r5j53_chunks_unbiased.txt 2>&1
cat gpurun_out/r5j53_chunks_unbiased.txt
echo "== bench C3 / C5: base (HEAD) / new / new with chunks"
AB_ENVS="GLH_TRACK_SPLIT=chunks" bash tools/ab.sh --no-secondary
AB_ENVS="GLH_TRACK_SPLIT=chunks" bash tools/ab.sh --workload C5 --points 2048
