#!/bin/bash
# round 3, call AH: HIP API statistics of the BAM -> genotypes probe (which runtime calls take host time)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r03_ah; mkdir -p $O
W=$R/tools/e2e/_data
g++ -std=c++17 -O2 -pthread -rdynamic -Iparagraph_amd/host/include -Itools/e2e -o $W/grmpy_batch tools/e2e/grmpy_batch.cpp -Lparagraph_amd -lparagraph_host -lparagraph_amd -Wl,-rpath,$PWD/paragraph_amd
export TMPDIR=/tmp
PG_E2E_REPS=4 timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d $O/prof -o e2e -- tools/e2e/_data/grmpy_batch tools/e2e/_data/ref.fa tools/e2e/_data/manifest.txt tools/e2e/_data/graphs.txt 16 $O/genotypes.json 0 0 1 > $O/run.json 2> $O/run.err
echo "rc=$?"
f=$(find $O/prof -name "*hip_api_stats.csv" | head -1)
cut -c1-140 $f | head -24
