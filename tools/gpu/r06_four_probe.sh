#!/bin/bash
# round 6: why does the e2e leg's all-four-stages pass read 22 k sites/s in the default command line and 30 k with the other legs off?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6four; mkdir -p $O
run() {
  python bench.py "$@" 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'args': '$*', 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s']), 'all_four': round(d['with_all_four_stages']['sites_genotyped_per_s'])}))" | tee -a $O/four_probe.jsonl
}
run --no-cpu-baseline --plain-steps 0 --exact-shortcut-steps 0
run --no-cpu-baseline --plain-steps 0 --exact-shortcut-steps 0 --sites-steps 0 --config5-graphs 0
run --no-cpu-baseline --plain-steps 0 --exact-shortcut-steps 0 --sites-steps 0 --config5-graphs 0 --stream-batches 0
run --no-cpu-baseline --plain-steps 0 --exact-shortcut-steps 0 --sites-steps 0 --config5-graphs 0 --stream-batches 0 --reads 20000 --steps 1 --warmup 0
