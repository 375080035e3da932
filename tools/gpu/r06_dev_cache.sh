#!/bin/bash
# round 6: the e2e leg's all-four-stages pass behind the default command line's other legs, with the device block cache letting its largest idle blocks go
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6devcache; mkdir -p $O
run() {
  python bench.py "$@" 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
e=d['e2e']
print(json.dumps({'args': '$*', 'value': round(d['value']), 'sites_genotyped_per_s': round(e['sites_genotyped_per_s']), 'with_path_matching': round(e['with_path_matching']['sites_genotyped_per_s']), 'all_four': round(e['with_all_four_stages']['sites_genotyped_per_s'])}))" | tee -a $O/dev_cache_ab.jsonl
}
run --no-cpu-baseline --plain-steps 0 --exact-shortcut-steps 0
run --no-cpu-baseline --plain-steps 0 --exact-shortcut-steps 0 --sites-steps 0 --config5-graphs 0 --stream-batches 0 --reads 20000 --steps 1 --warmup 0
