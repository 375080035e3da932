#!/bin/bash
# round 6, call aa: exact shortcut through the workflow + the bench's two new legs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6aa; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_exact or synthetic_sites or test_bench") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"; grep -E "^E" $O/tests.log | cut -c1-400 | head -20
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sites-steps 0 --config5-graphs 0 --stream-batches 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print("value", round(d["value"]), json.dumps(d.get("exact_shortcut")))
e=d["e2e"]; print("e2e", round(e["sites_genotyped_per_s"]), round(e["cpu_us_per_site_sample"],1), "path", round(e["with_path_matching"]["sites_genotyped_per_s"]), json.dumps(e.get("with_exact_shortcut")))
PY
tail -5 $O/bench.err
