# what does RCCL do with two ranks on one device?  (bounded: the duplicate-GPU check either fails fast or we stop it)
cd $GRAFT_REPO_ROOT
export GLH_BENCH_DEVICE=0
timeout 180 python bench.py --gpus 2 --workload C2 --points 16 --particles 600 --steps 3 --warmup 1 --no-cpu-baseline --no-api > gpurun_out/two_rccl.json 2> gpurun_out/two_rccl.err
echo "two-rank default transport exit $?" >> gpurun_out/two_rccl.err
