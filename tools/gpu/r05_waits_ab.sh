#!/bin/bash
# host waits on blocking events (default) against spinning ones (PG_SPIN_WAITS=1) in the bench process, where torch has the device
# in use before the workflow asks for hipDeviceScheduleBlockingSync: the e2e leg in both modes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5w; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_workflow or host_cpp or test_gpu_counts") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
run() {
  env "$@" python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 --e2e-options '{}' 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'env': '$*', 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'cpu_us_per_site_sample': round(d['cpu_us_per_site_sample'],1), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/waits_ab.jsonl
}
run PG_SPIN_WAITS=1
run A=0
run PG_SPIN_WAITS=1
run A=0
