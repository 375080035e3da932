#!/bin/bash
# round 3, call AG: column meta words 2 / 4 / 8 steps ahead on the 10 000-site set (every wavefront another graph: its meta words
# come from HBM, not from a cache the other wavefronts keep warm)
# (tools/variants/lib_*.so are other builds of the same sources made beforehand with tools/build_variant.sh <commit|WORK> <name> [-D...];
#  they are not tracked -- the script records what was compared, profiles/r03_trace_tax.md the outcome)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r03_ag; mkdir -p $O
for V in tree fa4 fa8 tree fa4 fa8; do
  if [ $V = tree ]; then unset PG_LIB; else export PG_LIB=$R/tools/variants/lib_$V.so; fi
  python bench.py --workload config3 --steps 4 --warmup 1 --no-cpu-baseline --collective off > $O/$V.json 2> $O/$V.err
  python - $O/$V.json $V <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["metric"][:40], round(d["value"]), "ms/step %.2f" % d["ms_per_step"])
PY
done
