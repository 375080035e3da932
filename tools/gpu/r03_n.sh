#!/bin/bash
# round 3, call N: (1) what a small read of a cold 128-byte line costs in HBM traffic (tools/gpu/sector_probe.hip), each counter
# in its own pass; (2) the BAM -> genotypes probe at other batch sizes / lane counts: host CPU seconds against wall clock
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r03_n; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o $O/sector_probe tools/gpu/sector_probe.hip 2> $O/build.err || { cat $O/build.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sector_$C -o sector -- $O/sector_probe > $O/sector_$C.out 2> $O/sector_$C.err
  echo "$C rc=$?"
done
cd "$R"
python - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_n")
for c in ("FETCH_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"):
    for f in glob.glob(O + "/sector_%s/**/*counter_collection.csv" % c, recursive=True):
        rows = list(csv.DictReader(open(f)))
        per = collections.defaultdict(list)
        for r in rows:
            if r["Kernel_Name"].startswith("k_"):
                per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        print(c, {k: [round(x) for x in v] for k, v in per.items()})
PY
for cfg in "0 0" "256 0" "512 0" "256 16" "512 16" "1024 16"; do
  set -- $cfg
  PG_E2E_DIR=tools/e2e/_data PG_E2E_REPS=5 timeout 300 tools/e2e/run.sh 10000 30 16 $1 $2 1 > $O/e2e_$1_$2.log 2>&1
  python - $O/e2e_$1_$2.log "$cfg" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"graphs"'):
        runs = json.loads(l)["runs"][1:]
        best = min(runs, key=lambda r: r["total_s"])
        cpu = sorted(r["cpu_user_s"] + r["cpu_sys_s"] for r in runs)[len(runs) // 2]
        print("per_batch/lanes", sys.argv[2], "batches", best["batches"], "lanes", best["lanes"], "best total_s %.3f" % best["total_s"], "median cpu_s %.2f" % cpu)
PY
done
