#!/bin/bash
# Round profiles (run on the GPU box through gpurun): for every BASELINE configuration that fits one GPU the bench line,
# the rocprofv3 kernel trace + stats of the same command, and the HBM traffic counters (separate --pmc passes, as
# MI355X_MICROARCH.md prescribes); SQ counters for C3.  usage: tools/profile_round.sh r03
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
out=gpurun_out/prof_$tag
mkdir -p $out
for cfg in "C3" "C2" "C4" "C5 --points 2048"; do
  set -- $cfg
  w=$1
  args="--workload $cfg --no-secondary"
  python3 bench.py $args --no-cpu-baseline --no-api > $out/${w}_bench.json 2> $out/${w}_bench.err
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/${w}_trace -o t --output-format csv -- python3 bench.py $args --no-cpu-baseline --no-api > $out/${w}_trace.json 2> $out/${w}_trace.log
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${w}_fetch -o f --output-format csv -- python3 bench.py $args --no-cpu-baseline --no-api > /dev/null 2> $out/${w}_fetch.log
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${w}_write -o w --output-format csv -- python3 bench.py $args --no-cpu-baseline --no-api > /dev/null 2> $out/${w}_write.log
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/C3_sq -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-api --no-secondary > /dev/null 2> $out/C3_sq.log
# the headline configuration with ONE launch per frame (the per-launch roofline of rounds 1-3), bench line + kernel trace
python3 bench.py --no-secondary --no-cpu-baseline --no-api --streams 1 > $out/C3s1_bench.json 2> $out/C3s1_bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $out/C3s1_trace -o t --output-format csv -- python3 bench.py --no-secondary --no-cpu-baseline --no-api --streams 1 > $out/C3s1_trace.json 2> $out/C3s1_trace.log
# the full default line (CPU baselines, API leg) last
python3 bench.py > $out/C3_full.json 2> $out/C3_full.err
ls -la $out | head -40
du -sh $out
