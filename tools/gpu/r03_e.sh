#!/bin/bash
# round 3, call E: whole GPU suite on the current tree + the BAM -> genotypes probe (host CPU per site) + the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_e
mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest.log" 2>&1
echo "pytest rc=$?"; tail -15 "$O/pytest.log"
PG_E2E_DIR=tools/e2e/_data PG_E2E_REPS=8 timeout 600 tools/e2e/run.sh 10000 30 16 0 0 1 > "$O/e2e.log" 2>&1
echo "e2e rc=$?"; tail -3 "$O/e2e.log" | cut -c1-400
cp gpurun_out/e2e_probe.json "$O/e2e_probe.json" 2>/dev/null
PG_BGZF_ZLIB=1 PG_E2E_DIR=tools/e2e/_data PG_E2E_REPS=5 timeout 600 tools/e2e/run.sh 10000 30 16 0 0 1 > "$O/e2e_zlib.log" 2>&1
cp gpurun_out/e2e_probe.json "$O/e2e_probe_zlib.json" 2>/dev/null
python - <<'PY'
import json
for f in ("e2e_probe", "e2e_probe_zlib"):
    try:
        d = json.load(open("gpurun_out/r03_e/%s.json" % f))
        for r in d["runs"][2:]:
            print(f, round(r["total_s"], 4), "s  cpu", round(r["cpu_user_s"] + r["cpu_sys_s"], 3), " extract", round(r["extract_reads_s"], 3), "docs", round(r["documents_s"], 3),
                  "geno", round(r["genotypes_s"], 3), "load", round(r["load_graphs_s"], 3), "dev", round(r["device_batch_s"], 3), "rel", round(r["release_s"], 3))
    except Exception as e:
        print(f, "failed", e)
PY
timeout 600 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_default.json').readline()); print(d['value'], d['ms_per_step'], d['roofline']['hbm_measured_frac'], d['roofline']['valu'].get('issue_frac'), d['sites']['sites_per_s'], d['sites']['reads_per_s'])"
