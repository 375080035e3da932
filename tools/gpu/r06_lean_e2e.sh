#!/bin/bash
# round 6: the BAM -> genotypes leg with the lean stage from 0 / 4096 / 16384 pairs per chunk on and with the plain stage
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6leane2e; mkdir -p $O
run() {
  env "$@" python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --config5-graphs 0 --no-cpu-baseline --stream-batches 0 --exact-shortcut-steps 0 --plain-steps 0 --no-e2e-shortcut 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'env': '$*', 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'cpu_us': round(d['cpu_us_per_site_sample'],1), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s']), 'all_four': round(d['with_all_four_stages']['sites_genotyped_per_s'])}))" | tee -a $O/lean_e2e_ab.jsonl
}
run PG_LEAN=0
run PG_LEAN_MIN_PAIRS=0
run PG_LEAN_MIN_PAIRS=4096
run PG_LEAN_MIN_PAIRS=16384
run PG_LEAN=0
run PG_LEAN_MIN_PAIRS=0
