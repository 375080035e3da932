#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5u; mkdir -p $O
for cfg in '{}' '{"sites_per_batch":512}' '{"sites_per_batch":768}' '{"sites_per_batch":384,"lanes":16}' '{"sites_per_batch":512,"lanes":16}' '{"sites_per_batch":384,"lanes":20}' '{}'; do
  python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 --e2e-options "$cfg" 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'options': json.loads('''$cfg'''), 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/batch_ab2.jsonl
done
