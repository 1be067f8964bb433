#!/bin/bash
# A/B on one GPU box: GLH_LIB=<base> vs the in-tree build, interleaved.  usage: tools/ab.sh [bench args]
for i in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export GLH_LIB=$PWD/glimpse_amd/lib/base.so; else unset GLH_LIB; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"
  done
done
