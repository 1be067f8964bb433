#!/bin/bash
# A/B on one GPU box, interleaved: `base` = glimpse_amd/lib/base.so (tools/mkbase.sh: HEAD), `new` = the in-tree
# build; extra variants in AB_ENVS (space separated): VAR=VALUE runs `new` with that environment, NAME.so runs
# glimpse_amd/lib/NAME.so, --flag=value runs `new` with that extra bench argument.
# prints: ms per step (wall), sum of the launch durations per step, GPU span per step (two streams: launches overlap)
# usage: [AB_ENVS="GLH_PT_ONE_BLOCK=1 nt.so --frames-per-call=1"] tools/ab.sh [bench args]
BASE=base; [ -f glimpse_amd/lib/base.so ] || BASE=""   # (no base.so: only `new` and the AB_ENVS variants run)
for i in 1 2 3; do
  for v in $BASE new $AB_ENVS; do
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
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', round(d['ms_per_step'],4), round(r['avg_launch_ms']*r['launches_per_step'],4), 'span', round(r.get('gpu_span_ms_per_frame', 0)*d['config']['frame_updates_per_step'],4), 'streams', r.get('concurrent_launches', 1))"
    )
  done
done
