#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5p8; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_path or test_gpu_workflow or test_gpu_kmer or test_gpu_klib or host_cpp") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
timeout 600 python tools/stage_probe.py 1000000 > $O/stage_probe.json 2> $O/stage_probe.err; echo "probe rc=$?"; python -c "
import json;d=json.load(open('$O/stage_probe.json'));print({k:(round(v['reads_per_s']/1e6,2) if isinstance(v,dict) and 'reads_per_s' in v else None) for k,v in d.items()})"
MODES=path bash tools/gpu/r05_mode_trace.sh | grep -A30 beside_the_fills | grep -B3 -A8 "pg_path_kernel"
for i in 1 2; do
python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 --e2e-options '{}' 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/e2e.jsonl
done
PG_SEED_STREAMS=2 python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 --e2e-options '{}' 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'seed_streams': 2, 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/e2e.jsonl
