#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6fused; mkdir -p $O
run() {
  env "$@" python bench.py --sites-steps 0 --config5-graphs 0 --e2e-steps 0 --exact-shortcut-steps 0 --no-cpu-baseline --stream-batches 0 --plain-steps 0 2> $O/err.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'env': '$*', 'value': round(d['value']), 'ms_per_step': round(d['ms_per_step'],2), 'kernel_ms': d.get('kernel_ms')}))" | tee -a $O/fused_probe.jsonl
}
run PG_LEAN_FUSED_PROBE=6
run PG_LEAN_FUSED_PROBE=7
run A=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 --e2e-steps 0 --config5-graphs 0 --exact-shortcut-steps 0 --plain-steps 0 > /dev/null 2> "$O/prof.err"
head -6 $O/prof/bench_kernel_stats.csv | cut -c1-200
