#!/bin/bash
# round 6, call i: the whole GPU suite on the new path kernel, the stage probe, the e2e leg (gssw-only / path / all four)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6i; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
python tools/stage_probe.py 1000000 > $O/stage_probe.json 2> $O/stage_probe.err; echo "probe rc=$?"; python -c "
import json; d=json.load(open('$O/stage_probe.json')); print({k: (round(v['reads_per_s']/1e6,2), round(v.get('vs_predicted',0),3)) for k,v in d.items() if isinstance(v,dict)})"
python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --config5-graphs 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 6 2> $O/e2e.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({k: d.get(k) for k in ('sites_genotyped_per_s','ms_per_step','cpu_us_per_site_sample','mismatches','genotypes_equal_truth','with_path_matching','with_all_four_stages')}))" | tee $O/e2e.json
