#!/bin/bash
# round 6, call l: new 8-slot test; where the lanes' time goes with the k-mer + klib stages on; kernel stats of that mode
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6l; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "device_slots") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
python tools/e2e/phase_probe.py 6000 kmer_sequence_matching=1 klib_sequence_matching=1 | tee $O/phase_kmer_klib.json
python tools/e2e/phase_probe.py 6000 klib_sequence_matching=1 | tee $O/phase_klib.json
python tools/e2e/phase_probe.py 6000 kmer_sequence_matching=1 | tee $O/phase_kmer.json
python tools/e2e/phase_probe.py 6000 | tee $O/phase_gssw.json
