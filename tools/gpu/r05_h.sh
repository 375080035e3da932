#!/bin/bash
# round 5, call h: the k-mer index built without string maps (path stage, KmerFilter, workflow tests); phases and e2e lines with and without path matching
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5h; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_path or test_gpu_counts or test_gpu_workflow or host_cpp or test_gpu_kmer or test_gpu_klib") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
python tools/e2e/phase_probe.py 6000 | tee $O/phase_gssw.json
python tools/e2e/phase_probe.py 6000 path_sequence_matching=1 | tee $O/phase_path.json
python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 6 2> $O/e2e.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({k: d[k] for k in ('sites_genotyped_per_s','ms_per_step','cpu_us_per_site_sample','mismatches','genotypes_equal_truth','with_path_matching')}))" | tee $O/e2e.json
