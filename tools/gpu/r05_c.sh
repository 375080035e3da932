#!/bin/bash
# round 5, call c: the device-resident cascade hand-over and the one-wait download under test; stage probe with the composed
# cascades; the e2e leg (A/A/B/B: fill streams 1 / 2) on the new host code
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5c; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_path or test_gpu_klib or test_gpu_kmer or test_gpu_workflow or test_gpu_counts or host_cpp or two_fill_streams") > $O/tests.log 2>&1; echo "tests rc=$? $(tail -4 $O/tests.log | head -1)"
python tools/stage_probe.py 1000000 > $O/stage_probe.json 2> $O/stage_probe.err; echo "probe rc=$?"; cat $O/stage_probe.json; tail -3 $O/stage_probe.err
for rep in 1 2; do for fs in 1 2; do
  PG_FILL_STREAMS=$fs python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 6 2> $O/e2e_fs${fs}_$rep.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'fill_streams': $fs, 'rep': $rep, 'sites_genotyped_per_s': d['sites_genotyped_per_s'], 'ms_per_step': d['ms_per_step'], 'cpu_us_per_site_sample': d['cpu_us_per_site_sample'], 'mismatches': d['mismatches'], 'concordant': d['genotypes_equal_truth']}))" | tee -a $O/tail_ab.jsonl
done; done
