#!/bin/bash
# which KFD calls the runtime makes in the BAM -> genotypes job, and what they cost: 8 passes under the counting shim
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5io; mkdir -p $O
D=/dev/shm/pg_prof; mkdir -p $D
python -c "
import sys,os
sys.path.insert(0,'.')
from paragraph_amd import synth_e2e
synth_e2e.make_dataset('$D', n_sites=10000, procs=os.cpu_count() or 1)"
g++ -std=c++17 -O2 -g -pthread -rdynamic -Iparagraph_amd/host/include -Itools/e2e -o /tmp/grmpy_batch tools/e2e/grmpy_batch.cpp -Lparagraph_amd -lparagraph_host -lparagraph_amd -Wl,-rpath,$PWD/paragraph_amd || exit 1
gcc -O2 -shared -fPIC -o /tmp/ioctl_count.so tools/e2e/ioctl_count.c -ldl
LD_PRELOAD=/tmp/ioctl_count.so PG_E2E_REPS=8 /tmp/grmpy_batch $D/ref.fa $D/manifest.txt $D/graphs.txt 16 $D/genotypes.json 0 0 1 > $O/run.json 2> $O/ioctl.txt
cat $O/ioctl.txt | tail -30
