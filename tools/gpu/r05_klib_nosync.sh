#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5kn; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_klib or test_gpu_kmer or test_gpu_workflow or host_cpp") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"; grep -E "^E |Error" $O/tests.log | head -5
python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 3 --e2e-verify 100 --e2e-options '{"kmer_sequence_matching":true,"klib_sequence_matching":true}' 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'kmer_klib_gssw_sites_per_s': round(d['sites_genotyped_per_s']), 'mismatches': d['mismatches'], 'all_four_sites_per_s': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee $O/e2e.json
