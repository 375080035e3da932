#!/bin/bash
# lanes with two chunks in flight: the workflow tests, then the e2e leg for several lane counts (and with two fill streams)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5x; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_workflow or host_cpp or two_ranks_shard") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
for cfg in '{}' '{"lanes":16}' '{"lanes":12}' '{"lanes":20}' '{"lanes":16,"sites_per_batch":256}' '{"lanes":16,"sites_per_batch":128}' '{}'; do
  python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 6 --e2e-options "$cfg" 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'options': json.loads('''$cfg'''), 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'cpu_us': round(d['cpu_us_per_site_sample'],1), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/pipe_ab.jsonl
done
PG_FILL_STREAMS=2 PG_FILLS_LOW=1 python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 6 --e2e-options '{"lanes":16}' 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'two_fill_streams_low': True, 'lanes': 16, 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/pipe_ab.jsonl
