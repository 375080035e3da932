#!/bin/bash
# round 6: everything profiles/r06_* is made from, on ONE box, for the kernel sources of the tree as it is:
#   the counters of every kernel besides fill / traceback (tools/stage_counters.sh -> r06_stage_counters.json); PMC traffic
#   (tools/pmc_collect.sh -> traffic_r06.json) and SQ counters (tools/sq_collect.sh -> r06_sq_counters.json) of the fill and the
#   traceback, each in rocprofv3 passes of its own with --kernel-trace only; kernel stats of the bench command
#   (r06_kernel_stats.csv, r06_bench_under_rocprof.json); the default bench line in the driver's form (r06_bench_default.json);
#   the stage probe (r06_stage_probe.json)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6u; mkdir -p $O
export TMPDIR=/tmp
bash tools/stage_counters.sh 200000 > $O/stage_counters.log 2>&1; echo "stage counters rc=$?"
python tools/stage_counters_summary.py gpurun_out/stage_counters $O/r06_stage_counters.json 200000 > $O/stage_counters_summary.txt 2> $O/stage_counters_summary.err; echo "summary rc=$?"; cat $O/stage_counters_summary.txt
PG_HEAD=r06 bash tools/pmc_collect.sh r06 > $O/pmc.log 2>&1; echo "pmc rc=$?"
PG_HEAD=r06 bash tools/sq_collect.sh > $O/sq.log 2>&1; echo "sq rc=$?"
python tools/pmc_traffic.py gpurun_out/pmc_r06 $O/traffic_r06.json > /dev/null 2> $O/traffic.err; echo "traffic rc=$?"
python tools/sq_summary.py gpurun_out/sq $O/r06_sq_counters.json > /dev/null 2> $O/sqsum.err; echo "sqsum rc=$?"
cp $O/traffic_r06.json profiles/traffic_r06.json; cp $O/r06_sq_counters.json profiles/r06_sq_counters.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 --e2e-steps 0 > "$O/bench_under_rocprof.json" 2> "$O/prof.err"; echo "rocprof rc=$?"
head -5 $O/prof/bench_kernel_stats.csv | cut -c1-160
cd "$R"
python tools/stage_probe.py 1000000 > $O/stage_probe.json 2> $O/stage_probe.err; echo "probe rc=$?"
(time python bench.py --steps 20 --warmup 5) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
r=d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "bound", r["bound"], "frac", r["frac"], "hbm_measured_frac", r.get("hbm_measured_frac"), "formula", r.get("hbm_formula_frac"), "launches", r["launches"], r["avg_launch_ms"])
print("verified", d.get("verified"))
print("sites", {k: d["sites"][k] for k in ("sites_per_s","reads_per_s","ms_per_step")}, d["sites"].get("verified",{}).get("mismatches"))
e=d["e2e"]; print("e2e", {k: e[k] for k in ("sites_genotyped_per_s","cpu_us_per_site_sample","mismatches","genotype_concordance")}, e.get("with_path_matching",{}).get("sites_genotyped_per_s"), e.get("verified",{}).get("site_mismatches"))
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"])
PY
