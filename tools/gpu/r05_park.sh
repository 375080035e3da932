#!/bin/bash
# stage calls that park the blocks they outgrow instead of waiting for the batch under the device lock: tests, e2e leg, call timing
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5k3; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_workflow or host_cpp or test_gpu_counts or test_gpu_path or test_gpu_klib or test_gpu_kmer") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
for i in 1 2 3; do
python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 --e2e-options '{}' 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'ms_per_step': round(d['ms_per_step'],1), 'cpu_us_per_site_sample': round(d['cpu_us_per_site_sample'],1), 'mismatches': d['mismatches'], 'concordant': d['genotypes_equal_truth'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/e2e.jsonl
done
PG_API_TIMING=1 python tools/e2e/mode_trace.py run /dev/shm/pg_t 4 > $O/api_timing.json 2> $O/api_timing.err; cat $O/api_timing.json; grep PG_API_TIMING $O/api_timing.err | head -14
