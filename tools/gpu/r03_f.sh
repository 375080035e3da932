#!/bin/bash
# round 3, call F: wide variants with 32 lanes per read -- parity first, then the read-length probe against the 16-lane kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_f
mkdir -p "$O"
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_klib.py tests/test_gpu_general.py tests/test_gpu_path.py -m gpu -q > "$O/pytest.log" 2>&1
echo "pytest rc=$?"; tail -25 "$O/pytest.log"
timeout 600 python tools/readlen_probe.py 200000 150,250,251,300,400,512 > "$O/readlen_wide32.json" 2> "$O/readlen_wide32.err"
PG_WIDE16=1 timeout 600 python tools/readlen_probe.py 200000 251,300,400,512 > "$O/readlen_wide16.json" 2> "$O/readlen_wide16.err"
python - <<'PY'
import json
for f in ("readlen_wide32", "readlen_wide16"):
    try:
        d = json.load(open("gpurun_out/r03_f/%s.json" % f))
        print(f, [(r["read_len"], r["tcups"], r["reads_per_s"], r["device"]["fill_ms"] / max(1, r["device"]["fill_launches"])) for r in d["rows"]])
    except Exception as e:
        print(f, "failed", e)
PY
