#!/bin/bash
# round 6, call x: the default bench form twice, with the cgroup's throttling counters around the e2e passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6x; mkdir -p $O
cat /sys/fs/cgroup/cpu.stat | head -8
for i in 1 2; do
python bench.py --steps 20 --warmup 5 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e=d['e2e']
print(json.dumps({'value': round(d['value']), 'sites_genotyped_per_s': round(e['sites_genotyped_per_s']), 'cpu_us': round(e['cpu_us_per_site_sample'],1), 'throttle': e.get('cpu_throttling_rank0'), 'with_path_matching': round(e['with_path_matching']['sites_genotyped_per_s']), 'path_cpu_us': round(e['with_path_matching']['cpu_us_per_site_sample_this_rank'],1), 'path_throttle': e['with_path_matching'].get('cpu_throttling_this_rank')}))" | tee -a $O/default_form.jsonl
done
