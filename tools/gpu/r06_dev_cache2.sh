#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6devcache; mkdir -p $O
run() {
  python bench.py "$@" 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
e=d['e2e']
print(json.dumps({'args': '$*', 'value': round(d['value']), 'sites_genotyped_per_s': round(e['sites_genotyped_per_s']), 'with_path_matching': round(e['with_path_matching']['sites_genotyped_per_s']), 'all_four': round(e['with_all_four_stages']['sites_genotyped_per_s'])}))" | tee -a $O/dev_cache_ab2.jsonl
}
run --no-cpu-baseline
run --plain-steps 0
run --exact-shortcut-steps 0
