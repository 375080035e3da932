#!/bin/bash
# the device-free half of the host pipeline (graph loading + packed extraction) on ONE core of the GPU box's host, with a sampled profile
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/host_probe; mkdir -p $O
D=tools/e2e/_data3k   # python tools/e2e/make_sites.py tools/e2e/_data3k 3000 30 1 (graphs.txt holds absolute paths of where it was made: rewritten below)
sed "s#^.*/graphs/#$R/$D/graphs/#" $D/graphs.txt > /tmp/graphs3k.txt
g++ -std=c++17 -O2 -g -pthread -rdynamic -Iparagraph_amd/host/include -Itools/e2e -o /tmp/host_probe tools/e2e/host_probe.cpp -Lparagraph_amd -lparagraph_host -lparagraph_amd -Wl,-rpath,$PWD/paragraph_amd || exit 1
taskset -c 3 /tmp/host_probe $D/ref.fa $D/reads.bam /tmp/graphs3k.txt 6 | tail -3 | tee $O/probe.jsonl
PG_E2E_PROF=$O/prof.txt taskset -c 3 /tmp/host_probe $D/ref.fa $D/reads.bam /tmp/graphs3k.txt 12 > /dev/null
python tools/e2e/prof_report.py $O/prof.txt 40 > $O/prof_report.txt 2>&1
sed -n '/^leaf/,$p' $O/prof_report.txt | head -70
