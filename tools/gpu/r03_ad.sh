#!/bin/bash
# round 3, call AD: seed cache + predecessor summary for the wide variants -- parity (all GPU tests of the fill), read-length probe
# A/B against the commit before (tools/variants/lib_head.so), headline check
# (tools/variants/lib_*.so are other builds of the same sources made beforehand with tools/build_variant.sh <commit|WORK> <name> [-D...];
#  they are not tracked -- the script records what was compared, profiles/r03_trace_tax.md the outcome)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r03_ad; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_general.py tests/test_gpu_counts.py -m gpu -q -x 2>&1 | tail -3
PG_LIB=$R/tools/variants/lib_head.so python tools/readlen_probe.py 100000 100,150,200,250,251,300,400,480 > $O/readlen_head.json 2> $O/readlen_head.err
python tools/readlen_probe.py 100000 100,150,200,250,251,300,400,480 > $O/readlen_new.json 2> $O/readlen_new.err
python - $O <<'PY'
import json, sys
for v in ("head", "new"):
    try:
        d = json.load(open("%s/readlen_%s.json" % (sys.argv[1], v)))
        print(v, [(r["read_len"], r["tcups"], r["reads_per_s"]) for r in d["rows"]])
    except Exception as e:
        print(v, "failed", e)
PY
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --stream-batches 0 --sites-steps 3 --collective off > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.3f M" % (d["value"] / 1e6), "sites/s", round(d["sites"]["sites_per_s"]), "cells", d["sites"]["cell_updates_per_s"] / 1e12)
PY
