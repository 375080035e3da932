#!/bin/bash
# round 5, call d: the whole -m gpu suite on the tree with the device-resident cascade, the stage probe, one e2e line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5d; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $O/tests.log 2>&1; echo "tests rc=$? $(tail -4 $O/tests.log | head -1)"
python tools/stage_probe.py 1000000 > $O/stage_probe.json 2> $O/stage_probe.err; echo "probe rc=$?"; cat $O/stage_probe.json; tail -3 $O/stage_probe.err
python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 6 2> $O/e2e.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({k: d[k] for k in ('sites_genotyped_per_s','ms_per_step','cpu_us_per_site_sample','mismatches','genotypes_equal_truth')}))" | tee $O/e2e.json
