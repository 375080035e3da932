#!/bin/bash
# round 5, call l: the hand-over without any host step (full plan + empty slots): whole -m gpu suite, stage probe, phases, e2e
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5l; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
python tools/stage_probe.py 1000000 > $O/stage_probe.json 2> $O/stage_probe.err; echo "probe rc=$?"; python -c "
import json; d=json.load(open('$O/stage_probe.json')); print({k: (round(v['reads_per_s']/1e6,2), round(v.get('vs_predicted',0),3)) for k,v in d.items() if isinstance(v,dict)})"
python tools/e2e/phase_probe.py 6000 path_sequence_matching=1 | tee $O/phase_path.json
python tools/e2e/phase_probe.py 6000 | tee $O/phase_gssw.json
python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 6 2> $O/e2e.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({k: d[k] for k in ('sites_genotyped_per_s','ms_per_step','cpu_us_per_site_sample','mismatches','genotypes_equal_truth','with_path_matching')}))" | tee $O/e2e.json
