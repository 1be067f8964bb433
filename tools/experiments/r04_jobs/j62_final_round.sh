bash tools/profile_round.sh r04d > gpurun_out/r4j62_round.log 2>&1; tail -2 gpurun_out/r4j62_round.log
python bench.py --steps 20 --warmup 5 --no-secondary > gpurun_out/prof_r04d/C3_driver.json 2> gpurun_out/prof_r04d/C3_driver.err
