#!/bin/bash
# round 3, call AA: second seed cache entry -- parity, then fill time against the library before it
# (tools/variants/lib_*.so are other builds of the same sources made beforehand with tools/build_variant.sh <commit|WORK> <name> [-D...];
#  they are not tracked -- the script records what was compared, profiles/r03_trace_tax.md the outcome)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r03_aa; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --stream-batches 0 --sites-steps 0 --collective off"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o bench -- $BENCH > $O/$name.json 2> $O/$name.err
  python - $O/$name <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + "/**/bench_kernel_stats.csv", recursive=True)
out = {}
for r in csv.DictReader(open(f[0])):
    for key in ("pg_fill_kernel", "pg_trace_kernel"):
        if key in r["Name"]:
            out[key[3:7]] = "%.3f" % (float(r["AverageNs"]) / 1e6)
d = json.loads(open(sys.argv[1] + ".json").read().strip().splitlines()[-1])
out["value_M"] = round(d["value"] / 1e6, 3)
print(sys.argv[1].split("/")[-1], out)
PY
}
run before_walk PG_LIB=$R/tools/variants/lib_seed2.so
run new_walk



