#!/bin/bash
# round 4, call B: (1) two-cycle vs four-cycle opcodes (valu_rate, extended list); (2) the fill with its three additions per cell
# pair as 32-bit integer additions on the bit patterns (v_add_u32: 2 cycles) instead of v_pk_add_f16 (4): A/B of the fill alone,
# the whole -m gpu suite, the bench line with its verification
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_b
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
for round in 1 2 3; do for v in base iadd; do
  PG_LIB=$R/tools/variants/lib_$v.so timeout 200 python tools/fill_probe.py 200000 2>/dev/null | tail -1 | sed "s/^{/{\"variant\": \"$v\", /" | tee -a "$O/ab.jsonl"
done; done
timeout 300 python bench.py --steps 10 --warmup 2 --sites-steps 2 --stream-batches 0 > "$O/bench_iadd.json" 2> "$O/bench_iadd.err"; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/bench_iadd.json").readline()); r = d["roofline"]
print(d["value"], d["ms_per_step"], r["avg_launch_ms"], d.get("verified"), d["sites"]["sites_per_s"])
PY
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$O/pytest.log" 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
timeout 600 tools/ubench/valu_rate 3000 > "$O/valu_rate.json" 2> "$O/valu_rate.err"; echo "valu_rate rc=$?"
python - <<PY
import json
d = json.load(open("$O/valu_rate.json"))
seen = set()
for r in d["rows"]:
    if r["waves_per_simd"] == 8 and r["chains"] == 8:
        print("%-32s %.3f ns  %.2f cyc@2.39" % (r["op"], r["ns_per_wave_inst_per_simd"], r["ns_per_wave_inst_per_simd"] * 2.39))
PY
