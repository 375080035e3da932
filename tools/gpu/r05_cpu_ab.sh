#!/bin/bash
# the e2e leg: default, and with glibc's mmap threshold raised (PG_MALLOPT=<MB>)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5c; mkdir -p $O
run() {
  env "$@" python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 --e2e-options '{}' 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'env': '$*', 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'cpu_us_per_site_sample': round(d['cpu_us_per_site_sample'],1), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/cpu_ab.jsonl
}
run A=0
run PG_MALLOPT=256
run A=0
run PG_MALLOPT=256
