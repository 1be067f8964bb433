export GLH_FRAME_CACHE=/tmp/glh_frames; mkdir -p $GLH_FRAME_CACHE
python bench.py --no-cpu-baseline --no-api --no-secondary --motion tangent_cartesian --dem gridded 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('main', d['ms_per_frame'], d['roofline']['avg_launch_ms'], d['config'].get('track_streams'), d.get('first_steps_ms'))"
python bench.py --no-cpu-baseline --no-api 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']['C3_tangent_dem']; print('leg', s)"
