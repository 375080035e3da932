#!/bin/bash
# e2e leg: (site, sample) pairs per device batch x lanes, now that the device (not the 16 CPUs) sets the rate
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5t; mkdir -p $O
for cfg in '{}' '{"sites_per_batch":256}' '{"sites_per_batch":384}' '{"sites_per_batch":384,"lanes":32}' '{"sites_per_batch":512,"lanes":32}' '{"lanes":32}' '{"sites_per_batch":128}' '{}'; do
  python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 6 --e2e-options "$cfg" 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'options': json.loads('''$cfg'''), 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'ms_per_step': round(d['ms_per_step'],1), 'cpu_us_per_site_sample': round(d['cpu_us_per_site_sample'],1), 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/batch_ab.jsonl
done
