#!/bin/bash
# round 3, call H: the two workflow tests again; SQ counters of the wide (32-lane) fill at 400 bp against the byte fill at 224 bp
# (both 14 rows per lane)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_h
mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest "tests/test_gpu_workflow.py::test_paragraph_validate_alignments" "tests/test_gpu_workflow.py::test_swaps_statistics_equal_the_references_expected_genotypes" -m gpu -q > "$O/pytest.log" 2>&1
echo "pytest rc=$?"; grep -E "^E|passed|failed" "$O/pytest.log" | head -30
cd /tmp && export TMPDIR=/tmp
for L in 400 224; do
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$O/sq_${L}_$i" -o sq -- python $R/tools/readlen_probe.py 100000 $L > /dev/null 2> "$O/sq_${L}_$i.err"
  done
done
python - <<'PY'
import csv, glob, collections
for L in (400, 224):
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob("/root/repo/gpurun_out/r03_h/sq_%d_*/**/*counter_collection.csv" % L, recursive=True):
        for r in csv.DictReader(open(f)):
            if "pg_fill_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(L, {k: (v, n[k]) for k, v in sorted(tot.items())})
PY
