#!/bin/bash
# e2e probe: glibc malloc tunables (the profile says a third of the CPU is malloc / free of JSON nodes and strings)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/o; mkdir -p $O
export PG_E2E_DIR=$R/tools/e2e/_data
W=$PG_E2E_DIR
export PG_E2E_REPS=5
bash tools/e2e/run.sh 10000 30 32 512 8 1 > $O/base.log 2>&1
ldd --version | head -1
run() { name=$1; shift; ( env "$@" $W/grmpy_batch $W/ref.fa $W/manifest.txt $W/graphs.txt 32 $W/g_$name.json 512 8 1 ) > $O/$name.json 2> $O/$name.err; python - <<PY
import json
d=json.load(open("$O/$name.json"))
print("$name", [("%.0f" % r["sites_per_s"], "u%.2f s%.2f" % (r["cpu_user_s"], r["cpu_sys_s"])) for r in d["runs"][1:]])
PY
}
run base A=1
run tcache GLIBC_TUNABLES=glibc.malloc.tcache_count=2000
run tcache_big GLIBC_TUNABLES=glibc.malloc.tcache_count=20000:glibc.malloc.tcache_max=4096
run base2 A=1
