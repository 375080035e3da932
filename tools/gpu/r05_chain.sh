#!/bin/bash
# fewer dispatches in the seed chain and the count pass (memsets folded into kernels, one publish kernel for the two sizes): the GPU
# suite's stage / cascade / workflow tests, stage probe, e2e leg
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5ch; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not test_gpu_scale") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"; grep -E "^E |Error" $O/tests.log | head -5
timeout 600 python tools/stage_probe.py 1000000 > $O/stage_probe.json 2> $O/stage_probe.err; python -c "
import json;d=json.load(open('$O/stage_probe.json'));print({k:(round(v['reads_per_s']/1e6,2) if isinstance(v,dict) and 'reads_per_s' in v else None) for k,v in d.items()})"
for i in 1 2 3; do
python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'cpu_us': round(d['cpu_us_per_site_sample'],1), 'mismatches': d['mismatches'], 'path': round(d['with_path_matching']['sites_genotyped_per_s']), 'path_cpu_us': round(d['with_path_matching']['cpu_us_per_site_sample_this_rank'],1)}))" | tee -a $O/e2e.jsonl
done
