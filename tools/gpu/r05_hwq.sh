#!/bin/bash
# the runtime's hardware-queue budget (GPU_MAX_HW_QUEUES, default 4): e2e leg in both modes and a short headline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5q; mkdir -p $O
run() {
  env "$@" python bench.py --reads 200000 --steps 10 --warmup 2 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 --e2e-verify 100 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]);e=d['e2e']
print(json.dumps({'env': '$*', 'reads_per_s': round(d['value']), 'sites_genotyped_per_s': round(e['sites_genotyped_per_s']), 'cpu_us': round(e['cpu_us_per_site_sample'],1), 'mismatches': e['mismatches'], 'with_path_matching': round(e['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/hwq_ab.jsonl
}
run A=0
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=16
run GPU_MAX_HW_QUEUES=8 PG_SEED_STREAMS=4
run A=0
run GPU_MAX_HW_QUEUES=8
