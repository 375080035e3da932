#!/bin/bash
# round 5, call i: k-mer index tables kept per thread; phases; e2e with one / two fill streams now that the host is no longer the limit
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5i; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_path or test_gpu_counts or test_gpu_workflow or host_cpp") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
python tools/e2e/phase_probe.py 6000 path_sequence_matching=1 | tee $O/phase_path.json
for rep in 1 2; do for fs in 1 2; do
  PG_FILL_STREAMS=$fs python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 6 2> $O/e2e_fs${fs}_$rep.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'fill_streams': $fs, 'rep': $rep, 'sites_genotyped_per_s': d['sites_genotyped_per_s'], 'ms_per_step': d['ms_per_step'], 'cpu_us_per_site_sample': d['cpu_us_per_site_sample'], 'mismatches': d['mismatches'], 'with_path_matching_sites_per_s': d['with_path_matching']['sites_genotyped_per_s']}))" | tee -a $O/tail_ab2.jsonl
done; done
