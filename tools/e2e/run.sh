#!/bin/bash
# End-to-end probe on the GPU box: synthetic sites -> BAM -> genotypes.  ./tools/e2e/run.sh [n_sites] [depth] [threads] [sites_per_batch] [lanes] [packed]
set -e
N=${1:-2000}; DEPTH=${2:-30}; THREADS=${3:-16}; PER_BATCH=${4:-1024}; LANES=${5:-4}; PACKED=${6:-1}
WORK=${PG_E2E_DIR:-/tmp/pg_e2e}; mkdir -p $WORK gpurun_out
# PG_E2E_DIR may point at a data set made beforehand (python tools/e2e/make_sites.py tools/e2e/_data ...), shipped with the repo
[ -f $WORK/reads.bam ] || python tools/e2e/make_sites.py $WORK $N $DEPTH 1
g++ -std=c++17 -O2 -pthread -rdynamic -Iparagraph_amd/host/include -Itools/e2e -o $WORK/grmpy_batch tools/e2e/grmpy_batch.cpp \
    -Lparagraph_amd -lparagraph_host -lparagraph_amd -Wl,-rpath,$PWD/paragraph_amd
$WORK/grmpy_batch $WORK/ref.fa $WORK/manifest.txt $WORK/graphs.txt $THREADS $WORK/genotypes.json $PER_BATCH $LANES $PACKED | tee gpurun_out/e2e_probe.json
python - <<PY
import json
truth = {t["ID"]: t for t in json.load(open("$WORK/truth.json"))}
got = json.load(open("$WORK/genotypes.json"))
ok = sum(1 for g in got if g["SYN"] == truth[g["ID"]]["gt"])
bad = [(g["ID"], g["SYN"], truth[g["ID"]]["gt"], truth[g["ID"]]["kind"]) for g in got if g["SYN"] != truth[g["ID"]]["gt"]][:8]
print("genotype concordance with the simulated truth: %d / %d" % (ok, len(got)), bad)
PY
