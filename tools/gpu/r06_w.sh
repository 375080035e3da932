#!/bin/bash
# round 6, call w: which of the bench's earlier legs depresses the e2e leg's path-matching pass (65 - 68 k in the default form, 85 k after light legs)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6w; mkdir -p $O
run() {
  name=$1; shift
  python bench.py "$@" 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'form': '$name', 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s']), 'path_cpu_us': round(d['with_path_matching']['cpu_us_per_site_sample_this_rank'],1), 'all_four': round(d['with_all_four_stages']['sites_genotyped_per_s'])}))" | tee -a $O/legs_ab.jsonl
}
run no_cpu_baseline --steps 5 --warmup 2 --no-cpu-baseline
run no_sites --steps 5 --warmup 2 --sites-steps 0
run small_headline --reads 20000 --steps 1 --warmup 0
run no_config5_no_stream --steps 5 --warmup 2 --config5-graphs 0 --stream-batches 0
