#!/bin/bash
# round 4: HIP API call counts of the BAM -> genotypes probe (3 passes of 10 000 sites = 369 batches): tools/gpu/r04_e2e_api.sh
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r04_e2e_api; mkdir -p $O
W=/tmp/pg_e2e_api; mkdir -p $W
g++ -std=c++17 -O2 -pthread -rdynamic -Iparagraph_amd/host/include -Itools/e2e -o $W/grmpy_batch tools/e2e/grmpy_batch.cpp \
    -Lparagraph_amd -lparagraph_host -lparagraph_amd -Wl,-rpath,$PWD/paragraph_amd || exit 1
D=tools/e2e/_data
export TMPDIR=/tmp
PG_E2E_REPS=3 timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d $W/prof -o e2e -- $W/grmpy_batch $D/ref.fa $D/manifest.txt $D/graphs.txt 16 $W/genotypes.json 0 0 1 > $O/run.log 2>&1
echo rc=$?
f=$(find $W/prof -name '*hip_api_stats.csv' | head -1); cp "$f" $O/hip_api_stats.csv; head -25 $O/hip_api_stats.csv | cut -d, -f1-4
