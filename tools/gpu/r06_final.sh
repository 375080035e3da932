#!/bin/bash
# round 6: the whole GPU suite, smoke, then the default bench line in the driver's form (twice)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6final; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"; grep -E "^E" $O/tests.log | cut -c1-300 | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2; do
(time python bench.py) > $O/bench_default_$i.json 2> $O/bench_default_$i.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default_$i.json") if l.startswith("{")][-1])
r=d["roofline"]
print("value", round(d["value"]), "ms", round(d["ms_per_step"],2), "frac", round(r["frac"],3), "hbm_measured", round(r.get("hbm_measured_frac") or 0,3), "h_only", round(r["hbm_alg_h_only_frac"],3), "streaming", round(d["value_streaming"]))
pl=d.get("plain_stage") or {}
print("plain_stage", round(pl.get("reads_per_s",0)), pl.get("records_differing_from_the_lean_step"), "hbm", pl.get("hbm_measured_frac"), "valu", pl.get("valu_issue_frac_with_measured_pairing"), "| lean one-traced-fill frac", r.get("hbm_alg_h_only_one_traced_fill_frac"), "bound", r["bound"])
print("verified", d.get("verified",{}).get("reads"), d.get("verified",{}).get("mismatches"), "shortcut", round(d["exact_shortcut"]["reads_per_s"]), d["exact_shortcut"]["records_differing_from_the_plain_step"], d["exact_shortcut"]["count_table_equal"])
print("sites", round(d["sites"]["sites_per_s"]), d["sites"].get("verified",{}).get("mismatches"), "config5", round(d["config5"]["reads_per_s"]), d["config5"]["verified"]["mismatches"])
e=d["e2e"]; print("e2e", round(e["sites_genotyped_per_s"]), round(e["cpu_us_per_site_sample"],1), e["mismatches"], "path", round(e["with_path_matching"]["sites_genotyped_per_s"]), "shortcut", round(e["with_exact_shortcut"]["sites_genotyped_per_s"]), e["with_exact_shortcut"]["documents_equal_the_gssw_only_run_on_this_rank"], "four", round(e["with_all_four_stages"]["sites_genotyped_per_s"]), "throttle", e.get("cpu_throttling_rank0"), e["with_path_matching"].get("cpu_throttling_this_rank"))
print("cpu_baseline", round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"])
PY
done
