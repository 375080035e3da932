#!/bin/bash
# one copy stream per calling thread in turn (PG_COPY_STREAMS): workflow tests with 4, the e2e leg with 1 / 2 / 4 / 8
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5cs; mkdir -p $O
(PG_COPY_STREAMS=4 timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_workflow or host_cpp or test_gpu_counts or test_gpu_path") > $O/tests.log 2>&1; echo "tests (4 copy streams) rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
run() {
  env "$@" python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 --e2e-options '{}' 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'env': '$*', 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'cpu_us_per_site_sample': round(d['cpu_us_per_site_sample'],1), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/copy_ab.jsonl
}
run PG_COPY_STREAMS=1
run PG_COPY_STREAMS=2
run PG_COPY_STREAMS=4
run PG_COPY_STREAMS=8
run PG_COPY_STREAMS=1
run PG_COPY_STREAMS=4
