#!/bin/bash
# seed streams made by the first path stage: 1 / 2 (default) / 3, the e2e leg in both modes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5s2; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_path or test_gpu_workflow") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
run() {
  env "$@" python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 --e2e-options '{}' 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'env': '$*', 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/seed_ab.jsonl
}
run PG_SEED_STREAMS=1
run PG_SEED_STREAMS=2
run PG_SEED_STREAMS=3
run PG_SEED_STREAMS=1
run PG_SEED_STREAMS=2
run PG_SEED_STREAMS=3
