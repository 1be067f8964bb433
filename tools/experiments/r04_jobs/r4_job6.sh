#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GLH_FRAME_CACHE=/tmp/glhfc
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_pinned.py tests/test_gpu_streams.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
echo "--- C3 (auto streams)"; AB_ENVS="prev.so" tools/ab.sh --no-secondary 2>&1 | grep -v "^base" | tee gpurun_out/r4j6_ab_C3.txt
echo "--- C3 one stream"; AB_ENVS="prev.so" tools/ab.sh --no-secondary --streams 1 2>&1 | grep -v "^base" | tee gpurun_out/r4j6_ab_C3s1.txt
echo "--- C4"; AB_ENVS="prev.so" tools/ab.sh --workload C4 --no-secondary 2>&1 | grep -v "^base" | tee gpurun_out/r4j6_ab_C4.txt
python tools/api_profile.py C3 2>&1 | head -45 | tee gpurun_out/r4j6_api_profile.txt
