#!/bin/bash
# last A/B of the round: seed streams at 192 pairs per batch; the workflow with all four stages
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5la; mkdir -p $O
run() {
  cfg="$1"; shift
  env "$@" python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 --e2e-verify 100 --e2e-options "$cfg" 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'options': json.loads('''$cfg'''), 'env': '$*', 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'mismatches': d['mismatches'], 'path': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/last_ab.jsonl
}
run '{}' PG_SEED_STREAMS=1
run '{}' PG_SEED_STREAMS=2
run '{}' PG_SEED_STREAMS=3
run '{"kmer_sequence_matching":true,"klib_sequence_matching":true}' A=0
