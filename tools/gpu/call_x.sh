#!/bin/bash
# headline vs workspace size (reads per fill launch)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/x
for ws in 64 128 200 64 128 200; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --sites-steps 0 --stream-batches 0 --workspace-gib $ws 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print($ws, 'GiB:', round(d['value']/1e6,2), 'M reads/s', round(d['ms_per_step'],2), 'ms/step; launches', r['launches'], 'avg', round(r['avg_launch_ms'],2), 'ms frac', round(r['frac'],3))"
done
