#!/bin/bash
# where the device's time goes in an e2e pass: kernel trace of the e2e leg alone (6 timed passes + warm-up + 3 passes with the path stage)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5y; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o e2e -- python $R/bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 6 > $O/bench.json 2> $O/bench.err)
python - <<PY
import csv, glob, json, collections
f = glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
tot = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    tot[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; cnt[n] += 1
d = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])["e2e"]
passes = 1 + 6 + 3   # warm-up + timed + (1 warm-up + 2 timed) with the path stage
out = {"e2e_ms_per_pass": d["ms_per_step"], "sites_genotyped_per_s": d["sites_genotyped_per_s"], "passes_in_trace": passes,
       "kernel_ms_per_pass": {k: round(v / passes, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]},
       "launches_per_pass": {k: round(cnt[k] / passes, 1) for k in sorted(tot, key=lambda k: -tot[k])[:8]}}
print(json.dumps(out))
open("$O/e2e_device_time.json", "w").write(json.dumps(out, indent=1))
PY
