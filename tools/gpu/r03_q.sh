#!/bin/bash
# round 3, call Q: traceback variants (register budget x rounds of diagonal loads in flight) x wavefronts in flight: parity, then
# kernel stats of the headline loop
# (tools/variants/lib_*.so are other builds of the same sources made beforehand with tools/build_variant.sh <commit|WORK> <name> [-D...];
#  they are not tracked -- the script records what was compared, profiles/r03_trace_tax.md the outcome)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r03_q; mkdir -p $O
for V in w8b4 w6b4 w5b4; do
  PG_LIB=$R/tools/variants/lib_$V.so PG_TRACE_BLOCKS=2048 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --stream-batches 0 --sites-steps 0 --collective off"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o bench -- $BENCH > $O/$name.json 2> $O/$name.err
  python - $O/$name <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + "/**/bench_kernel_stats.csv", recursive=True)
if not f:
    print(sys.argv[1], "no stats"); sys.exit(0)
out = {}
for r in csv.DictReader(open(f[0])):
    for key in ("pg_fill_kernel", "pg_trace_kernel"):
        if key in r["Name"]:
            out[key[3:7]] = "%.3f" % (float(r["AverageNs"]) / 1e6)
try:
    d = json.loads(open(sys.argv[1] + ".json").read().strip().splitlines()[-1])
    out["value_M"] = round(d["value"] / 1e6, 3)
except Exception as e:
    out["value_M"] = str(e)
print(sys.argv[1].split("/")[-1], out)
PY
}
for V in w8b1 w8b4 w6b2 w6b4; do
  for B in 2048 3072; do run ${V}_$B PG_LIB=$R/tools/variants/lib_$V.so PG_TRACE_BLOCKS=$B; done
done
for B in 1024 1536 2048; do run w5b4_$B PG_LIB=$R/tools/variants/lib_w5b4.so PG_TRACE_BLOCKS=$B; done
