#!/bin/bash
# round 6: the lean stage in one launch per chunk (pg_fill_lean_fused_kernel, the default) against its three-launch form and the plain
# stage on the headline; the BAM -> genotypes leg with it for every chunk
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6fused; mkdir -p $O
run() {
  env "$@" python bench.py --sites-steps 0 --config5-graphs 0 --e2e-steps 0 --exact-shortcut-steps 0 --no-cpu-baseline --stream-batches 0 --plain-steps 0 2> $O/err.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'env': '$*', 'value': round(d['value']), 'ms_per_step': round(d['ms_per_step'],2), 'kernel_ms': d.get('kernel_ms')}))" | tee -a $O/fused_ab.jsonl
}
run A=1
run PG_LEAN_FUSED=0
run PG_LEAN=0
run A=1
run PG_LEAN_FUSED=0
e2e() {
  env "$@" python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --config5-graphs 0 --no-cpu-baseline --stream-batches 0 --exact-shortcut-steps 0 --plain-steps 0 --no-e2e-shortcut 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'env': '$*', 'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'cpu_us': round(d['cpu_us_per_site_sample'],1), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s']), 'path_equal': d['with_path_matching']['genotypes_equal_the_gssw_only_run_on_this_rank'], 'all_four': round(d['with_all_four_stages']['sites_genotyped_per_s'])}))" | tee -a $O/fused_e2e_ab.jsonl
}
e2e A=1
e2e PG_LEAN_MIN_CELLS=0
e2e A=1
e2e PG_LEAN_MIN_CELLS=0
PG_LEAN_MIN_CELLS=0 python bench.py --sites-steps 0 --config5-graphs 0 --e2e-steps 0 --exact-shortcut-steps 0 --stream-batches 0 2> $O/err2.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'value': round(d['value']), 'verified': d.get('verified'), 'plain': {k: d['plain_stage'][k] for k in ('reads_per_s','records_differing_from_the_lean_step','cigar_strings_equal')}})[:900])"
