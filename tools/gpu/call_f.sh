#!/bin/bash
# round-2 profile call: end-to-end probe (BAM -> genotypes) plain / under rocprofv3, bench under rocprofv3, PMC traffic, SQ counters
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/f; mkdir -p $O
export PG_E2E_DIR=$R/tools/e2e/_data
W=$PG_E2E_DIR
# ---- 1. end to end, 32 host threads, 8 lanes, packed reads (three runs inside grmpy_batch? it prints its own timing) ----
( time PG_E2E_REPS=8 bash tools/e2e/run.sh 10000 30 16 0 0 1 ) > $O/e2e_run.log 2>&1; tail -4 $O/e2e_run.log | cut -c1-400
cp gpurun_out/e2e_probe.json $O/e2e_probe.json 2>/dev/null
( PG_DEVICES=0,0 $W/grmpy_batch $W/ref.fa $W/manifest.txt $W/graphs.txt 16 $W/genotypes2.json 0 0 1 ) > $O/e2e_two_slots.json 2> $O/e2e_two_slots.err; tail -c 600 $O/e2e_two_slots.json
export TMPDIR=/tmp  # (stay in the repo root: the manifest of the e2e data set names its BAM relative to it)
# ---- 2. the same under rocprofv3: kernel stats, then HIP API stats (separate runs, no counters) ----
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e_kernels -o e2e -- $W/grmpy_batch $W/ref.fa $W/manifest.txt $W/graphs.txt 16 $W/g3.json 0 0 1 > $O/e2e_kernels.out 2> $O/e2e_kernels.err
timeout 300 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $O/e2e_hip -o e2e -- $W/grmpy_batch $W/ref.fa $W/manifest.txt $W/graphs.txt 16 $W/g4.json 0 0 1 > $O/e2e_hip.out 2> $O/e2e_hip.err
# ---- 3. bench under rocprofv3 (kernel stats) and its HIP API stats with the streaming leg ----
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kernels -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 > $O/bench_kernels.json 2> $O/bench_kernels.err
timeout 400 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $O/bench_hip -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 8 > $O/bench_hip.json 2> $O/bench_hip.err
find $O -name "*stats*.csv" | head -20
for f in $(find $O -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
for f in $(find $O -name "*hip_api_stats.csv"); do echo "== $f"; head -12 $f; done
# ---- 4. PMC traffic + SQ counters (own runs) ----
bash tools/pmc_collect.sh r02 > $O/pmc_collect.log 2>&1; python tools/pmc_traffic.py gpurun_out/pmc_r02 gpurun_out/f/traffic_r02.json 200000 > $O/pmc_traffic.log 2>&1; tail -12 $O/pmc_traffic.log
bash tools/sq_collect.sh > $O/sq_collect.log 2>&1; tail -3 $O/sq_collect.log
# keep the merged output small
find $O -name "*.db" -delete; find $O -name "*trace.csv" -size +2M -delete; du -sh $O gpurun_out/pmc_r02 gpurun_out/sq
