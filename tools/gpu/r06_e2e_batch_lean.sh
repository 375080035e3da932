#!/bin/bash
# round 6: the BAM -> genotypes leg with larger batches per lane, plain stage and lean stage for every chunk
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6e2ebatch; mkdir -p $O
e2e() {
  spb=$1; shift
  env "$@" python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --config5-graphs 0 --no-cpu-baseline --stream-batches 0 --exact-shortcut-steps 0 --plain-steps 0 --no-e2e-shortcut --e2e-options "{\"sites_per_batch\": $spb}" 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'sites_per_batch': $spb, 'env': '$*', 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'cpu_us': round(d['cpu_us_per_site_sample'],1), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/e2e_batch_lean.jsonl
}
e2e 192 A=1
e2e 384 A=1
e2e 384 PG_LEAN_MIN_CELLS=0
e2e 768 A=1
e2e 768 PG_LEAN_MIN_CELLS=0
e2e 1536 PG_LEAN_MIN_CELLS=0
