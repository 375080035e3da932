#!/bin/bash
# lanes and batch size again, now that the device lock is no longer what the lanes queue for
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5l; mkdir -p $O
for cfg in '{}' '{"lanes":32}' '{"lanes":48}' '{"sites_per_batch":256}' '{"sites_per_batch":128}' '{"lanes":32,"sites_per_batch":128}' '{"lanes":16}' '{}'; do
  python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 --e2e-options "$cfg" 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'options': json.loads('''$cfg'''), 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'cpu_us_per_site_sample': round(d['cpu_us_per_site_sample'],1), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/lanes_ab.jsonl
done
