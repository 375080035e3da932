#!/bin/bash
# round 6, call a: the whole GPU suite on the tree with the ADVICE fixes + the bench in the driver's form (new keys: hbm_alg_h_only_frac,
# value_streaming, config5)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6a; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
(time python bench.py) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
r=d['roofline']
print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'value_streaming': d.get('value_streaming'), 'frac': r['frac'], 'h_only': r.get('hbm_alg_h_only_frac'),
  'hbm_measured': r.get('hbm_measured_frac'), 'avg_launch_ms': r['avg_launch_ms'], 'verified': d.get('verified',{}).get('mismatches'),
  'config5': {k: d['config5'].get(k) for k in ('reads_per_s','ms_per_step','cell_updates_per_s','hbm_alg_h_only_frac')} if 'config5' in d else None,
  'config5_verified': d.get('config5',{}).get('verified',{}).get('mismatches'),
  'e2e': {k: d['e2e'].get(k) for k in ('sites_genotyped_per_s','cpu_us_per_site_sample','mismatches')} if 'e2e' in d else None}))
PY
