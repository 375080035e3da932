#!/bin/bash
# round 3, call I: the tree as it will be judged -- whole GPU suite, default bench line, read-length probe, kernel stats of the
# bench under rocprofv3, PMC traffic + SQ counters of the fill (separate --pmc passes)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_i
mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest.log" 2>&1
echo "pytest rc=$?"; tail -6 "$O/pytest.log"
timeout 600 python tools/readlen_probe.py 200000 100,150,250,251,300,400,480 > "$O/readlen_wide32.json" 2> "$O/readlen_wide32.err"
PG_WIDE16=1 timeout 600 python tools/readlen_probe.py 200000 251,300,400,480 > "$O/readlen_wide16.json" 2> "$O/readlen_wide16.err"
python - <<'PY'
import json
for f in ("readlen_wide32", "readlen_wide16"):
    try:
        d = json.load(open("gpurun_out/r03_i/%s.json" % f))
        print(f, [(r["read_len"], r["tcups"], r["reads_per_s"]) for r in d["rows"]])
    except Exception as e:
        print(f, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 > "$O/bench_under_rocprof.json" 2> "$O/prof.err"
echo "rocprof rc=$?"
cd "$R"
PG_HEAD=${PG_HEAD:-unknown} bash tools/pmc_collect.sh r03 > "$O/pmc.log" 2>&1
PG_HEAD=${PG_HEAD:-unknown} bash tools/sq_collect.sh > "$O/sq.log" 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_r03 gpurun_out/r03_i/traffic_r03.json 200000 > /dev/null 2>&1
python tools/sq_summary.py gpurun_out/sq gpurun_out/r03_i/r03_sq_counters.json 200000 > /dev/null 2>&1
cp gpurun_out/r03_i/traffic_r03.json profiles/traffic_r03.json; cp gpurun_out/r03_i/r03_sq_counters.json profiles/r03_sq_counters.json
timeout 600 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "bench rc=$? lines=$(wc -l < $O/bench_default.json)"
python -c "
import json; d=json.loads(open('$O/bench_default.json').readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['hbm_measured_frac'], r['valu'].get('issue_frac'), r['traffic_source']['usable'], d['sites']['sites_per_s'], d['sites']['cell_updates_per_s'], d['cell_updates_per_s'], d['dist']['collective_ab']['with_vs_without'])"
