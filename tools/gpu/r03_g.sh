#!/bin/bash
# round 3, call G: read-length probe (wide variants: 32 lanes per read against the 16-lane kernels) + the new workflow tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_g
mkdir -p "$O"
cd "$R"
timeout 600 python tools/readlen_probe.py 200000 150,250,251,300,400,480 > "$O/readlen_wide32.json" 2> "$O/readlen_wide32.err"
PG_WIDE16=1 timeout 600 python tools/readlen_probe.py 200000 251,300,400,480 > "$O/readlen_wide16.json" 2> "$O/readlen_wide16.err"
python - <<'PY'
import json
for f in ("readlen_wide32", "readlen_wide16"):
    try:
        d = json.load(open("gpurun_out/r03_g/%s.json" % f))
        print(f, [(r["read_len"], r["tcups"], r["reads_per_s"], round(r["device"]["fill_ms"] / max(1, r["device"]["fill_launches"]), 3)) for r in d["rows"]])
    except Exception as e:
        print(f, "failed", e)
PY
timeout 900 python -m pytest "tests/test_gpu_workflow.py::test_paragraph_validate_alignments" "tests/test_gpu_workflow.py::test_swaps_statistics_equal_the_references_expected_genotypes" tests/test_gpu_scale.py::test_stress_parity -m gpu -q > "$O/pytest.log" 2>&1
echo "pytest rc=$?"; tail -40 "$O/pytest.log"
