# round 5, job 21: does rocprofv3's PC sampling (beta) work here?  One short run, under its own timeout.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc; mkdir -p $GLH_FRAME_CACHE gpurun_out
python3 bench.py --workload C3 --points 1024 --no-cpu-baseline --no-api --no-secondary --burn-in 6 --steps 6 --warmup 1 --repeats 1 --streams 1 > /dev/null 2>&1
timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 100 --kernel-trace -d gpurun_out/pcs -o p --output-format csv -- python3 bench.py --workload C3 --points 1024 --no-cpu-baseline --no-api --no-secondary --burn-in 6 --steps 6 --warmup 1 --repeats 1 --streams 1 > gpurun_out/pcs.log 2>&1
echo "rc=$?"; tail -5 gpurun_out/pcs.log; ls -la gpurun_out/pcs 2>/dev/null | head; find gpurun_out/pcs -type f | head; 
for f in $(find gpurun_out/pcs -name "*pc_sampling*" | head -3); do echo "== $f"; head -5 $f; wc -l $f; done
