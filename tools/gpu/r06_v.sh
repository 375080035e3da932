#!/bin/bash
# round 6, call v: the e2e leg's path-matching rate: 3 against 6 timed passes, light preceding legs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6v; mkdir -p $O
run() {
  python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --config5-graphs 0 --no-cpu-baseline --stream-batches 0 --e2e-steps $1 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'e2e_steps': $1, 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'cpu_us_per_site_sample': round(d['cpu_us_per_site_sample'],1), 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s']), 'path_cpu_us': round(d['with_path_matching']['cpu_us_per_site_sample_this_rank'],1), 'all_four': round(d['with_all_four_stages']['sites_genotyped_per_s'])}))" | tee -a $O/steps_ab.jsonl
}
run 3; run 6; run 3; run 6
