#!/bin/bash
# round 3, call AO: the workspace budget of the bench (chunk size = reads per fill launch): 64 GiB (default, 200 k reads per launch),
# 128 and 200 GiB
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
for W in 64 128 200 64 128; do
  python bench.py --workspace-gib $W --steps 6 --warmup 2 --no-cpu-baseline --stream-batches 0 --sites-steps 0 --collective off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ws $W GiB', round(d['value']/1e6,3), 'launches', r['launches'], 'avg ms', round(r['avg_launch_ms'],3), 'ms/step', round(d['ms_per_step'],2))"
done
