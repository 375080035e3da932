#!/bin/bash
# A/B of fill-kernel variants + parity of the working tree
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
bash tools/gpu/call_ab.sh "$@" | python -c "
import sys, json, collections
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
by=collections.defaultdict(list)
for r in rows: by[r['variant']].append(r)
for v,rs in by.items():
    ks=[k for k in rs[0] if isinstance(rs[0][k],(int,float)) and ('ms' in k or 'per_s' in k)]
    print(v, {k: round(min(r[k] for r in rs) if 'ms' in k else max(r[k] for r in rs),3) for k in ks})
"
mkdir -p gpurun_out/s
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_scale.py -m gpu -q --timeout 800 -p no:cacheprovider -x ) > gpurun_out/s/pytest.log 2>&1; tail -4 gpurun_out/s/pytest.log
