#!/bin/bash
# round 4: what a wave64 VALU instruction costs on this chip and the clock it holds under the fill
#   -> gpurun_out/r04_ubench/valu_rate.json (profiles/r04_valu_rate.json), clock_probe.json (profiles/r04_clock_probe.json is
#      its amd-smi series + summary), grbm.csv (GRBM_GUI_ACTIVE of the fill launches under --pmc: cycles of all 8 XCDs)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_ubench
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
timeout 600 tools/ubench/valu_rate 3000 > "$O/valu_rate.json" 2> "$O/valu_rate.err"; echo "valu_rate rc=$?"
timeout 300 python tools/clock_probe.py "$O/clock_probe.json" -- python bench.py --steps 60 --warmup 3 --no-cpu-baseline --sites-steps 0 --stream-batches 0 > "$O/bench_60steps.json" 2> "$O/bench_60steps.err"; echo "clock probe rc=$?"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$O/grbm" -o g -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 --collective off > /dev/null 2> "$O/grbm.err"; echo "grbm rc=$?"
