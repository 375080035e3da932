#!/bin/bash
# where the host's CPU goes in the BAM -> genotypes job on the final tree: tools/e2e/grmpy_batch on the bench's synthetic data set, 8
# plain passes, then 30 passes under the SIGPROF sampler (tools/e2e/prof.hh)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6prof; mkdir -p $O
D=/dev/shm/pg_prof; mkdir -p $D
python -c "
import sys,os
sys.path.insert(0,'.')
from paragraph_amd import synth_e2e
synth_e2e.make_dataset('$D', n_sites=10000, procs=os.cpu_count() or 1)"
g++ -std=c++17 -O2 -g -pthread -rdynamic -Iparagraph_amd/host/include -Itools/e2e -o /tmp/grmpy_batch tools/e2e/grmpy_batch.cpp -Lparagraph_amd -lparagraph_host -lparagraph_amd -Wl,-rpath,$PWD/paragraph_amd || exit 1
PG_E2E_REPS=8 /tmp/grmpy_batch $D/ref.fa $D/manifest.txt $D/graphs.txt 16 $D/genotypes.json 0 0 1 > $O/e2e_probe.json 2> $O/e2e_probe.err
python - <<PY
import json
d = json.load(open("$O/e2e_probe.json"))
for r in d["runs"][1:]:
    print("total %.4f s  cpu %.3f+%.3f s  sites/s %.0f  faults %d" % (r["total_s"], r["cpu_user_s"], r["cpu_sys_s"], r["sites_per_s"], r.get("minor_faults", 0)))
PY
PG_E2E_PROF=$O/prof.txt PG_E2E_REPS=30 /tmp/grmpy_batch $D/ref.fa $D/manifest.txt $D/graphs.txt 16 $D/genotypes.json 0 0 1 > $O/e2e_prof.json 2> $O/e2e_prof.err
python tools/e2e/prof_report.py $O/prof.txt 70 > $O/prof_report.txt 2>&1
gzip -f $O/prof.txt
sed -n '1,12p' $O/prof_report.txt; sed -n '/^leaf/,$p' $O/prof_report.txt | head -60
