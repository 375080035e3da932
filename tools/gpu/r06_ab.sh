#!/bin/bash
# round 6: the all-four-stages pass of the e2e leg with and without the exact-shortcut passes in front of it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6ab; mkdir -p $O
run() {
  python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --config5-graphs 0 --no-cpu-baseline --stream-batches 0 --exact-shortcut-steps 0 "$@" 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'args': '$*', 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s']), 'shortcut': round(d['with_exact_shortcut']['sites_genotyped_per_s']) if d.get('with_exact_shortcut') else None, 'all_four': round(d['with_all_four_stages']['sites_genotyped_per_s'])}))" | tee -a $O/all_four_ab.jsonl
}
run --no-e2e-shortcut; run; run --no-e2e-shortcut; run
