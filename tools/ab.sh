#!/bin/bash
# A/B on one GPU box, interleaved: `base` = glimpse_amd/lib/base.so (tools/mkbase.sh: HEAD), `new` = the in-tree
# build, plus one run of `new` per extra VAR=VALUE given in AB_ENVS (space separated).
# usage: [AB_ENVS="GLH_PT_RENOISE=1"] tools/ab.sh [bench args]
for i in 1 2 3; do
  for v in base new $AB_ENVS; do
    (
      if [ $v = base ]; then export GLH_LIB=$PWD/glimpse_amd/lib/base.so; elif [ $v != new ]; then export $v; fi
      python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"
    )
  done
done
