#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes for the HBM traffic of pg_fill_kernel, each counter in its own
# rocprofv3 run with --kernel-trace only (never combined with sys/hip/hsa tracing), plus a calibration run on
# kernels with a known byte count (torch zero_ / copy_ of 1 GiB), as MI355X_MICROARCH.md's HBM section asks.
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/pmc_${1:-r02}
mkdir -p "$OUT"
echo "${PG_HEAD:-unknown}" > "$OUT/head.txt"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOTDIR/bench.py --steps 1 --warmup 0 --reads 200000 --no-cpu-baseline --stream-batches 0 --sites-steps 0 --e2e-steps 0 --collective off --exact-shortcut-steps 0 --plain-steps 0 --config5-graphs 0"
CAL="python $ROOTDIR/tools/pmc_calibrate.py"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/bench_$C" -o bench -- $BENCH > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.err"
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/cal_$C" -o cal -- $CAL > "$OUT/cal_$C.out" 2> "$OUT/cal_$C.err"
done
find "$OUT" -name "*.csv" | head -20
