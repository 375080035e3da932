#!/bin/bash
# round 6, call ac: the device-built path index, its two build launches now on the seed stream of the first path stage: path / workflow / cascade tests, then the e2e leg A/B (PG_PATH_INDEX_DEVICE=1 = the device builder; the host builder is the default)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6ac; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_path or test_gpu_workflow or host_cpp or hand_over or cascade") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
run() {
  env "$@" python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --config5-graphs 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 6 --no-e2e-shortcut --exact-shortcut-steps 0 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'env': '$*', 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'cpu_us_per_site_sample': round(d['cpu_us_per_site_sample'],1), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s']), 'path_cpu_us': round(d['with_path_matching']['cpu_us_per_site_sample_this_rank'],1), 'path_equal': d['with_path_matching']['genotypes_equal_the_gssw_only_run_on_this_rank']}))" | tee -a $O/index_ab.jsonl
}
run PG_PATH_INDEX_DEVICE=1
run A=1
run PG_PATH_INDEX_DEVICE=1
run A=1
