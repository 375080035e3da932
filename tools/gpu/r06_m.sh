#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6m; mkdir -p $O
for spb in 192 384 768; do
python tools/e2e/phase_probe.py 10000 kmer_sequence_matching=1 klib_sequence_matching=1 path_sequence_matching=1 sites_per_batch=$spb | tee -a $O/phase_all_four_spb.jsonl
done
for spb in 384 768; do
python tools/e2e/phase_probe.py 10000 sites_per_batch=$spb | tee -a $O/phase_gssw_spb.jsonl
python tools/e2e/phase_probe.py 10000 path_sequence_matching=1 sites_per_batch=$spb | tee -a $O/phase_path_spb.jsonl
done
