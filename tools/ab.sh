#!/bin/bash
# A/B on one GPU box, interleaved: `base` = glimpse_amd/lib/base.so (tools/mkbase.sh: HEAD), `new` = the in-tree
# build; extra variants in AB_ENVS (space separated): VAR=VALUE runs `new` with that environment, NAME.so runs
# glimpse_amd/lib/NAME.so, --flag=value runs `new` with that extra bench argument.
# usage: [AB_ENVS="GLH_PT_ONE_BLOCK=1 nt.so --frames-per-call=1"] tools/ab.sh [bench args]
for i in 1 2 3; do
  for v in base new $AB_ENVS; do
    (
      extra=""
      case $v in
        base) export GLH_LIB=$PWD/glimpse_amd/lib/base.so ;;
        new) ;;
        *.so) export GLH_LIB=$PWD/glimpse_amd/lib/$v ;;
        --*) extra=$v ;;
        *) export $v ;;
      esac
      python bench.py --no-cpu-baseline --no-api "$@" $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms']*d['roofline']['launches_per_step'],4))"
    )
  done
done
