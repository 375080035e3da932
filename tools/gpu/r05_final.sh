#!/bin/bash
# round 5, last call: the whole -m gpu suite and smoke() on the final tree, then the default bench line in the driver's form
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5z; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
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
tail -3 $O/bench_default.err
