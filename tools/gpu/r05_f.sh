#!/bin/bash
# round 5, call f: counters of the kernels besides fill / traceback; the new tests; the pipelined stage probe; the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
bash tools/stage_counters.sh 200000 > $O/stage_counters.log 2>&1; echo "stage counters rc=$?"
python tools/stage_counters_summary.py gpurun_out/stage_counters $O/r05_stage_counters.json 200000 > $O/stage_counters_summary.txt 2> $O/stage_counters_summary.err; echo "summary rc=$?"; cat $O/stage_counters_summary.txt; tail -3 $O/stage_counters_summary.err
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_kmer or test_gpu_path or test_gpu_klib or test_gpu_workflow or host_cpp") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
python tools/stage_probe.py 1000000 > $O/stage_probe.json 2> $O/stage_probe.err; echo "probe rc=$?"; cat $O/stage_probe.json; tail -3 $O/stage_probe.err
(time python bench.py --steps 20 --warmup 5) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
r=d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "bound", r["bound"], "frac", r["frac"], "hbm_measured_frac", r.get("hbm_measured_frac"), "formula", r.get("hbm_formula_frac"), "launches", r["launches"], r["avg_launch_ms"])
print("verified", d.get("verified"))
print("sites", {k: d["sites"][k] for k in ("sites_per_s","reads_per_s","ms_per_step")}, d["sites"].get("verified",{}).get("mismatches"))
e=d["e2e"]; print("e2e", {k: e[k] for k in ("sites_genotyped_per_s","cpu_us_per_site_sample","mismatches","genotype_concordance")}, e.get("with_path_matching"), e.get("verified",{}).get("site_mismatches"))
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"])
PY
