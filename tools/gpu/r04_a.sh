#!/bin/bash
# round 4, call A: (1) what a wave64 VALU instruction costs + is the compiler's s_nop after packed instructions needed
# (tools/ubench/valu_rate), (2) the clock the chip holds under the fill (SMI samples during a bench loop; GRBM_GUI_ACTIVE of the
# fill launches), (3) A/B of the whole library with those nops stripped from pg_fill (tools/build_nonop_variant.sh): timing of
# the fill alone, the bench line with its 1 M-read verification against the reference's gssw.c.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_a
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
timeout 600 tools/ubench/valu_rate 3000 > "$O/valu_rate.json" 2> "$O/valu_rate.err"; echo "valu_rate rc=$?"
python - <<PY
import json
d = json.load(open("$O/valu_rate.json"))
print(d["recurrence_without_nops_vs_compiler_emitted"], d["clock_rate_khz_reported"], d["wall_clock_rate_khz"])
for r in d["rows"]:
    if r["waves_per_simd"] in (1, 4, 8) and r["chains"] in (0, 1, 8):
        print("%-42s w=%d ch=%d  %.3f ns  %.2f ticks  ratio %.3f" % (r["op"], r["waves_per_simd"], r["chains"], r["ns_per_wave_inst_per_simd"], r["memtime_ticks_per_wave_inst_median"], r["memtime_per_memrealtime_median"]))
PY
# (2) clock under load
timeout 300 python tools/clock_probe.py "$O/clock_probe.json" -- python bench.py --steps 60 --warmup 3 --no-cpu-baseline --sites-steps 0 --stream-batches 0 > "$O/bench_60steps.json" 2> "$O/bench_60steps.err"; echo "clock probe rc=$?"
tail -2 "$O/bench_60steps.err" | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$O/grbm" -o g -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 --collective off > /dev/null 2> "$O/grbm.err"; echo "grbm rc=$?"
cd "$R"
python - <<PY
import csv, glob, collections
cnt = collections.defaultdict(list)
for p in glob.glob("$O/grbm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "pg_fill_kernel" in r["Kernel_Name"]:
            cnt[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r.get("End_Timestamp", 0)) - int(r.get("Start_Timestamp", 0))))
for k, v in cnt.items():
    print(k, [(c, ns, c / ns if ns else None) for c, ns in v][:4])
PY
# (3) A/B: nops stripped
for round in 1 2 3; do for v in base nonop; do
  PG_LIB=$R/tools/variants/lib_$v.so timeout 200 python tools/fill_probe.py 200000 2>/dev/null | tail -1 | sed "s/^{/{\"variant\": \"$v\", /" | tee -a "$O/ab.jsonl"
done; done
for v in nonop base; do
  PG_LIB=$R/tools/variants/lib_$v.so timeout 300 python bench.py --steps 10 --warmup 2 --sites-steps 1 --stream-batches 0 > "$O/bench_$v.json" 2> "$O/bench_$v.err"; echo "bench $v rc=$?"
  python - <<PY
import json
d = json.loads(open("$O/bench_$v.json").readline()); r = d["roofline"]
print("$v", d["value"], d["ms_per_step"], r["avg_launch_ms"], d.get("verified"), d["sites"]["sites_per_s"])
PY
done
