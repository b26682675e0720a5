# headline step over extract chunk size x extraction lanes (one box, back to back): bash tools/gpu_chunk_lane_sweep.sh
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "512 2" "256 2" "1024 1" "512 1" "256 4" "512 3" "256 3" "342 3" "128 4" "128 2"; do set -- $cfg
python bench.py --no-c2 --no-cpu-baseline --extract-chunk $1 --extract-lanes $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $1 lanes $2: %.1f keyframes/s  %.3f ms per step  (%s W mean)' % (d['value'], d['ms_per_step'], d['board'].get('power_W_mean')))"
done; done
